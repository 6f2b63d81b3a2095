"""RNG-free deterministic tensor fill, keyed by name (test infrastructure).

The SD2-inpainting checkpoint is not shipped with the reference
(README.md:37-38, test_inpainting.py:96), so every parity test fills both the
reference model and the build with the same closed-form pseudo-random values:

    seed  = crc32(name)
    u_i   = splitmix64(i + seed * 0x9E3779B97F4A7C15)  ->  [-1, 1)
    ndim>=2 weight : u * sqrt(3 / fan_in)      (unit-variance preserving)
    1-D "*.weight" : 1 + 0.1 u                 (norm gains)
    1-D otherwise  : 0.02 u                    (biases)

It also overwrites the reference's zero-initialised layers
(openaimodel.py:228,730; attention.py:381), which would otherwise make a
fresh UNet output exactly 0.
"""
import zlib

import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 arrays (wrap-around math)."""
    with np.errstate(over="ignore"):
        z = x + _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def uniform_pm1(name, numel, offset=0):
    """`numel` float64 values in [-1, 1), a pure function of (name, index)."""
    seed = np.uint64(zlib.crc32(name.encode("utf-8")))
    with np.errstate(over="ignore"):
        base = seed * _GOLD
        idx = np.arange(offset, offset + numel, dtype=np.uint64) + base
    z = _splitmix64(idx)
    # top 53 bits -> [0,1) -> [-1,1)
    u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return u * 2.0 - 1.0


def fill_like(name, shape, kind=None):
    """float32 numpy array of `shape` for parameter/input `name`.

    kind: None -> infer from name/shape as in the module docstring;
          "unit" -> plain u in [-1,1); "normalish" -> sum of 4 uniforms scaled
          to unit variance (cheap Gaussian-like inputs for latents/contexts).
    """
    shape = tuple(int(s) for s in shape)
    numel = int(np.prod(shape)) if len(shape) else 1
    if kind == "normalish":
        acc = np.zeros(numel, dtype=np.float64)
        for j in range(4):
            acc += uniform_pm1(name + "#%d" % j, numel)
        # var of U(-1,1) = 1/3 ; sum of 4 -> 4/3
        return (acc * np.sqrt(3.0 / 4.0)).astype(np.float32).reshape(shape)
    u = uniform_pm1(name, numel)
    if kind == "unit":
        return u.astype(np.float32).reshape(shape)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        return (u * np.sqrt(3.0 / fan_in)).astype(np.float32).reshape(shape)
    if name.endswith(".weight"):
        return (1.0 + 0.1 * u).astype(np.float32).reshape(shape)
    return (0.02 * u).astype(np.float32).reshape(shape)


def fill_state_dict(shapes, prefix=""):
    """shapes: {key: shape}. Returns {key: torch.float32 tensor}."""
    import torch

    out = {}
    for k, shp in shapes.items():
        out[k] = torch.from_numpy(fill_like(prefix + k, shp))
    return out
