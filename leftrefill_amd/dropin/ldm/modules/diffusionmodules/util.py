"""Schedules, embeddings and layer factories with the reference's names (ldm/modules/diffusionmodules/util.py).

The nn layers created here are parameter containers with the reference's state-dict keys; on the hot path their
weights are re-laid-out once and consumed by the HIP kernels (leftrefill_amd.engine), not by torch.
"""
import math

import numpy as np
import torch
import torch.nn as nn


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    """float64 beta table (reference util.py:21-43)."""
    if schedule == "linear":
        return np.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=np.float64) ** 2
    if schedule == "sqrt_linear":
        return np.linspace(linear_start, linear_end, n_timestep, dtype=np.float64)
    if schedule == "sqrt":
        return np.linspace(linear_start, linear_end, n_timestep, dtype=np.float64) ** 0.5
    if schedule == "cosine":
        ts = np.arange(n_timestep + 1, dtype=np.float64) / n_timestep + cosine_s
        a = np.cos(ts / (1 + cosine_s) * np.pi / 2) ** 2
        a = a / a[0]
        return np.clip(1 - a[1:] / a[:-1], 0, 0.999)
    raise ValueError(f"schedule '{schedule}' unknown.")


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps, verbose=True):
    """Sub-sampled timestep list, +1 shifted (reference util.py:46-60): S=50 -> [1, 21, ..., 981]."""
    if ddim_discr_method == "uniform":
        stride = num_ddpm_timesteps // num_ddim_timesteps
        base = np.arange(0, num_ddpm_timesteps, stride)
    elif ddim_discr_method == "quad":
        base = (np.linspace(0, np.sqrt(num_ddpm_timesteps * .8), num_ddim_timesteps) ** 2).astype(int)
    else:
        raise NotImplementedError(f'There is no ddim discretization method called "{ddim_discr_method}"')
    steps = base + 1
    if verbose:
        print(f"Selected timesteps for ddim sampler: {steps}")
    return steps


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta, verbose=True):
    """(sigmas, alphas, alphas_prev) as float64 numpy.  `alphacums` is the fp32-rounded alpha-bar table; like the
    reference (util.py:63-74) 1-a_t is formed in fp32 and the rest in float64."""
    ac32 = np.asarray(alphacums, dtype=np.float32)
    a32 = ac32[ddim_timesteps]
    alphas = a32.astype(np.float64)
    alphas_prev = np.concatenate([ac32[:1], ac32[ddim_timesteps[:-1]]]).astype(np.float64)
    # The reference evaluates (1 - a_prev) / (1 - a_t) with a_t an fp32 *tensor*: numpy defers to
    # Tensor.__rtruediv__, i.e. an fp32 reciprocal of (1 - a_t) times the float64 numerator.  Reproduced bit for bit.
    recip32 = (np.float32(1) / (np.float32(1) - a32)).astype(np.float32)
    ratio = a32.astype(np.float64) / alphas_prev
    sigmas = eta * np.sqrt((1 - alphas_prev) * recip32.astype(np.float64) * (1 - ratio))
    if verbose:
        print(f"Selected alphas for ddim sampler: a_t: {alphas}; a_(t-1): {alphas_prev}")
        print(f"For the chosen value of eta, which is {eta}, this results in the following sigma_t schedule "
              f"for ddim sampler {sigmas}")
    return sigmas, alphas, alphas_prev


def extract_into_tensor(a, t, x_shape):
    out = a.gather(-1, t)
    return out.reshape(t.shape[0], *((1,) * (len(x_shape) - 1)))


def noise_like(shape, device, repeat=False):
    if repeat:
        return torch.randn((1, *shape[1:]), device=device).repeat(shape[0], *((1,) * (len(shape) - 1)))
    return torch.randn(shape, device=device)


def checkpoint(func, inputs, params, flag):
    """Inference-only build: activation checkpointing (reference util.py:102-151) is a pass-through."""
    return func(*inputs)


def timestep_embedding(timesteps, dim, max_period=10000, repeat_only=False):
    """[N] -> [N, dim] sinusoidal embedding, cos first (reference util.py:154-174).  On a HIP device this is one
    kernel (lr_timestep_embedding, fp16 out); the host branch exists for CPU-side glue and tests."""
    if repeat_only:
        return timesteps[:, None].repeat(1, dim)
    if timesteps.is_cuda:
        from leftrefill_amd import ops
        return ops.timestep_embedding(timesteps, dim)
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


class GroupNorm32(nn.GroupNorm):
    """32-group GroupNorm, eps 1e-5 (reference util.py:217-219): parameter container for lr_groupnorm_*."""


def normalization(channels):
    return GroupNorm32(32, channels)


def conv_nd(dims, *args, **kwargs):
    if dims != 2:
        raise ValueError(f"unsupported dimensions: {dims} (the MI355X build covers the 2-D UNet only)")
    return nn.Conv2d(*args, **kwargs)


def linear(*args, **kwargs):
    return nn.Linear(*args, **kwargs)
