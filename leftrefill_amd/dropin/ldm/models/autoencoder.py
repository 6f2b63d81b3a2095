"""`ldm.models.autoencoder.AutoencoderKL` (reference ldm/models/autoencoder.py:13-91), inference part only:
encode -> DiagonalGaussianDistribution(quant_conv(encoder(x))), decode -> decoder(post_quant_conv(z)).
Host-side PyTorch-ROCm module (VAE stays off the HIP hot path by the north star); state-dict keys as the reference."""
import torch
import torch.nn as nn

from ldm.modules.diffusionmodules.model import Decoder, Encoder
from ldm.modules.distributions.distributions import DiagonalGaussianDistribution
from ldm.util import instantiate_from_config


class AutoencoderKL(nn.Module):
    def __init__(self, ddconfig, lossconfig=None, embed_dim=4, ckpt_path=None, ignore_keys=(), image_key="image",
                 colorize_nlabels=None, monitor=None, ema_decay=None, learn_logvar=False):
        super().__init__()
        self.learn_logvar = learn_logvar
        self.image_key = image_key
        ddconfig = dict(ddconfig)
        self.encoder = Encoder(**ddconfig)
        self.decoder = Decoder(**ddconfig)
        self.loss = instantiate_from_config(lossconfig) if lossconfig else None
        assert ddconfig["double_z"]
        self.quant_conv = nn.Conv2d(2 * ddconfig["z_channels"], 2 * embed_dim, 1)
        self.post_quant_conv = nn.Conv2d(embed_dim, ddconfig["z_channels"], 1)
        self.embed_dim = embed_dim
        if monitor is not None:
            self.monitor = monitor
        self.use_ema = False
        if ckpt_path is not None:
            sd = torch.load(ckpt_path, map_location="cpu")["state_dict"]
            sd = {k: v for k, v in sd.items() if not any(k.startswith(ik) for ik in ignore_keys)}
            self.load_state_dict(sd, strict=False)

    def encode(self, x):
        return DiagonalGaussianDistribution(self.quant_conv(self.encoder(x)))

    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))

    def forward(self, input, sample_posterior=True):
        posterior = self.encode(input)
        z = posterior.sample() if sample_posterior else posterior.mode()
        return self.decode(z), posterior
