"""Config plumbing: the dotted-path instantiation that is LeftRefill's plugin mechanism (reference ldm/util.py:71-86)."""
import importlib


def exists(v):
    return v is not None


def default(v, d):
    if v is not None:
        return v
    return d() if callable(d) else d


def get_obj_from_str(path, reload=False):
    module, _, cls = path.rpartition(".")
    mod = importlib.import_module(module)
    if reload:
        mod = importlib.reload(mod)
    return getattr(mod, cls)


def instantiate_from_config(config):
    if "target" not in config:
        if config in ("__is_first_stage__", "__is_unconditional__"):
            return None
        raise KeyError("Expected key `target` to instantiate.")
    params = config.get("params", dict())
    return get_obj_from_str(config["target"])(**(params if params is not None else {}))


def count_params(model, verbose=False):
    n = sum(p.numel() for p in model.parameters())
    if verbose:
        print(f"{model.__class__.__name__} has {n * 1e-6:.2f} M params.")
    return n
