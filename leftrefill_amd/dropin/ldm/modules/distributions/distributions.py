"""Diagonal Gaussian posterior of the KL-VAE (reference ldm/modules/distributions/distributions.py:24-70).

Quirk kept on purpose: `sample()` re-seeds the global torch RNG (CPU and device) to 42 on EVERY call
(distributions.py:35-39), so after `get_input` the RNG state -- and therefore the `torch.randn` start code of the DDIM
sampler -- is a fixed function of the device."""
import numpy as np
import torch


class DiagonalGaussianDistribution(object):
    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self):
        torch.manual_seed(42)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(42)
        noise = torch.randn(self.mean.shape).to(device=self.parameters.device)   # host RNG, like the reference
        return self.mean + self.std * noise

    def mode(self):
        return self.mean

    def kl(self, other=None):
        if self.deterministic:
            return torch.Tensor([0.])
        if other is None:
            return 0.5 * torch.sum(self.mean ** 2 + self.var - 1.0 - self.logvar, dim=[1, 2, 3])
        return 0.5 * torch.sum((self.mean - other.mean) ** 2 / other.var + self.var / other.var - 1.0 - self.logvar
                               + other.logvar, dim=[1, 2, 3])

    def nll(self, sample, dims=(1, 2, 3)):
        if self.deterministic:
            return torch.Tensor([0.])
        return 0.5 * torch.sum(np.log(2.0 * np.pi) + self.logvar + (sample - self.mean) ** 2 / self.var, dim=list(dims))
