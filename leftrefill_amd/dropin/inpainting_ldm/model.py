"""`inpainting_ldm.model`: create_model / load_state_dict, as imported by test_inpainting.py:19
(reference inpainting_ldm/model.py:13-29).  YAML is read with PyYAML when omegaconf is absent."""
import os

import torch

from ldm.util import instantiate_from_config


def get_state_dict(d):
    return d.get('state_dict', d)


def load_state_dict(ckpt_path, location='cpu'):
    _, ext = os.path.splitext(ckpt_path)
    if ext.lower() == ".safetensors":
        import safetensors.torch
        sd = safetensors.torch.load_file(ckpt_path, device=location)
    else:
        sd = get_state_dict(torch.load(ckpt_path, map_location=torch.device(location)))
    sd = get_state_dict(sd)
    print(f'Loaded state_dict from [{ckpt_path}]')
    return sd


class _Cfg(dict):
    """Minimal attribute-access dict standing in for an OmegaConf node."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    @staticmethod
    def wrap(o):
        if isinstance(o, dict):
            return _Cfg({k: _Cfg.wrap(v) for k, v in o.items()})
        if isinstance(o, list):
            return [_Cfg.wrap(v) for v in o]
        return o


def load_config(config_path):
    try:
        from omegaconf import OmegaConf
        return OmegaConf.load(config_path)
    except ImportError:
        import yaml
        with open(config_path) as f:
            return _Cfg.wrap(yaml.safe_load(f))


def create_model(config_path):
    config = load_config(config_path)
    model = instantiate_from_config(config.model).cpu()
    print(f'Loaded model config from [{config_path}]')
    return model
