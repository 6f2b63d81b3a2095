"""`ldm.models.autoencoder.AutoencoderKL` (reference ldm/models/autoencoder.py:13-91), inference part only:
encode -> DiagonalGaussianDistribution(quant_conv(encoder(x))), decode -> decoder(post_quant_conv(z)).
State-dict keys as the reference.  CUDA inputs run on the HIP kernels (leftrefill_amd.vae_engine: the conv / GroupNorm /
GEMM kernels of the UNet step; weights are packed on first use and re-packed when a parameter changes); the nn.Module
graph itself is the PyTorch host-side definition (used for CPU tensors, e.g. the CPU golden pin) -- set `use_hip = False`
to force it on a GPU."""
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

    use_hip = True

    def _sig(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def prepare(self, force=False):
        """Pack the weights for the HIP kernels; redone automatically when a parameter was (re)loaded or moved."""
        from leftrefill_amd import vae_engine
        sig = self._sig()
        packed = getattr(self, "_lr_packed", None)
        if packed is None or force or packed[0] != sig:
            bad = [n for n, m in self.named_modules() if isinstance(m, nn.Conv2d) and n not in
                   ("encoder.conv_in", "decoder.conv_out", "encoder.conv_out", "decoder.conv_in", "quant_conv",
                    "post_quant_conv") and (m.in_channels % 64 or m.out_channels % 64)]
            if bad:
                raise RuntimeError(f"VAE on the HIP kernels needs channel widths that are multiples of 64 ({bad[0]}); "
                                   "set `use_hip = False` on this AutoencoderKL to run the PyTorch definition")
            with torch.no_grad():
                packed = (sig, vae_engine.PackedEncoder(self.encoder, self.quant_conv),
                          vae_engine.PackedDecoder(self.decoder, self.post_quant_conv))
            self._lr_packed = packed
        return packed

    def encode(self, x):
        if x.is_cuda and self.use_hip:
            from leftrefill_amd import vae_engine
            with torch.no_grad():
                return DiagonalGaussianDistribution(vae_engine.encode_moments(x.float(), self.prepare()[1]))
        return DiagonalGaussianDistribution(self.quant_conv(self.encoder(x)))

    def decode(self, z):
        if z.is_cuda and self.use_hip:
            from leftrefill_amd import vae_engine
            with torch.no_grad():
                return vae_engine.decode(z.float(), self.prepare()[2])
        return self.decoder(self.post_quant_conv(z))

    def forward(self, input, sample_posterior=True):
        posterior = self.encode(input)
        z = posterior.sample() if sample_posterior else posterior.mode()
        return self.decode(z), posterior
