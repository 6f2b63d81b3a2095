"""KL-VAE encoder / decoder modules with the reference's class names and state-dict keys
(ldm/modules/diffusionmodules/model.py: Upsample 51-66, Downsample 69-88, ResnetBlock 91-150, AttnBlock 153-204,
Encoder 453-544, Decoder 547-653) so `first_stage_config` instantiates and the SD2 VAE weights (`first_stage_model.*`,
248 tensors) load.

These nn.Modules hold the parameters and define the PyTorch formulation; on a HIP device `AutoencoderKL.encode / decode`
run `leftrefill_amd/vae_engine.py` (the same implicit-GEMM conv / GroupNorm / attention kernels as the UNet step, SURVEY.md
section 8f-1) on the weights packed from them.  The module `forward`s below are the host-side formulation the north star
leaves on PyTorch-ROCm: they serve CPU tensors, autograd through the VAE, and `use_hip=False`.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ldm.modules.attention import MemoryEfficientCrossAttention  # noqa: F401  (name imported by the reference, line 10)


def nonlinearity(x):
    return x * torch.sigmoid(x)


def Normalize(in_channels, num_groups=32):
    return nn.GroupNorm(num_groups=num_groups, num_channels=in_channels, eps=1e-6, affine=True)


class Upsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)

    def forward(self, x):
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        return self.conv(x) if self.with_conv else x


class Downsample(nn.Module):
    """stride-2 conv on an input padded (0,1,0,1): asymmetric, unlike the UNet's Downsample (model.py:83)."""

    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=2, padding=0)

    def forward(self, x):
        if self.with_conv:
            return self.conv(F.pad(x, (0, 1, 0, 1), mode="constant", value=0))
        return F.avg_pool2d(x, kernel_size=2, stride=2)


class ResnetBlock(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout, temb_channels=512):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.use_conv_shortcut = conv_shortcut
        self.norm1 = Normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if temb_channels > 0:
            self.temb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = Normalize(out_channels)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if in_channels != out_channels:
            if conv_shortcut:
                self.conv_shortcut = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
            else:
                self.nin_shortcut = nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)

    def forward(self, x, temb=None):
        h = self.conv1(nonlinearity(self.norm1(x)))
        if temb is not None:
            h = h + self.temb_proj(nonlinearity(temb))[:, :, None, None]
        h = self.conv2(self.dropout(nonlinearity(self.norm2(h))))
        if self.in_channels != self.out_channels:
            x = self.conv_shortcut(x) if self.use_conv_shortcut else self.nin_shortcut(x)
        return x + h


class AttnBlock(nn.Module):
    """Single-head spatial self-attention with 1x1-conv projections, scale C^-1/2 (model.py:153-204)."""

    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.k = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.v = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.proj_out = nn.Conv2d(in_channels, in_channels, kernel_size=1)

    def forward(self, x):
        h = self.norm(x)
        b, c, hh, ww = h.shape
        q = self.q(h).reshape(b, c, hh * ww).transpose(1, 2)
        k = self.k(h).reshape(b, c, hh * ww).transpose(1, 2)
        v = self.v(h).reshape(b, c, hh * ww).transpose(1, 2)
        o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None], scale=float(c) ** -0.5)[:, 0]
        return x + self.proj_out(o.transpose(1, 2).reshape(b, c, hh, ww))


def make_attn(in_channels, attn_type="vanilla", attn_kwargs=None):
    if attn_type in ("vanilla", "vanilla-xformers"):
        return AttnBlock(in_channels)
    if attn_type == "none":
        return nn.Identity(in_channels)
    raise NotImplementedError(f"attn_type {attn_type!r} is not used by the LeftRefill VAE")


class Encoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True, use_linear_attn=False,
                 attn_type="vanilla", **ignore_kwargs):
        super().__init__()
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        self.conv_in = nn.Conv2d(in_channels, ch, kernel_size=3, stride=1, padding=1)
        cur = resolution
        widths = (1,) + tuple(ch_mult)
        self.in_ch_mult = widths
        self.down = nn.ModuleList()
        block_in = ch
        for lvl in range(self.num_resolutions):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_in, block_out = ch * widths[lvl], ch * ch_mult[lvl]
            for _ in range(num_res_blocks):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0, dropout=dropout))
                block_in = block_out
                if cur in attn_resolutions:
                    attn.append(make_attn(block_in, attn_type=attn_type))
            down = nn.Module()
            down.block, down.attn = block, attn
            if lvl != self.num_resolutions - 1:
                down.downsample = Downsample(block_in, resamp_with_conv)
                cur //= 2
            self.down.append(down)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type=attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, 2 * z_channels if double_z else z_channels, kernel_size=3, stride=1,
                                  padding=1)

    def forward(self, x):
        h = self.conv_in(x)
        for lvl in range(self.num_resolutions):
            for i in range(self.num_res_blocks):
                h = self.down[lvl].block[i](h, None)
                if len(self.down[lvl].attn) > 0:
                    h = self.down[lvl].attn[i](h)
            if lvl != self.num_resolutions - 1:
                h = self.down[lvl].downsample(h)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h, None)), None)
        return self.conv_out(nonlinearity(self.norm_out(h)))


class Decoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False,
                 use_linear_attn=False, attn_type="vanilla", **ignorekwargs):
        super().__init__()
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        self.give_pre_end, self.tanh_out = give_pre_end, tanh_out
        block_in = ch * ch_mult[-1]
        cur = resolution // 2 ** (self.num_resolutions - 1)
        self.z_shape = (1, z_channels, cur, cur)
        self.conv_in = nn.Conv2d(z_channels, block_in, kernel_size=3, stride=1, padding=1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type=attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        ups = []
        for lvl in reversed(range(self.num_resolutions)):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_out = ch * ch_mult[lvl]
            for _ in range(num_res_blocks + 1):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0, dropout=dropout))
                block_in = block_out
                if cur in attn_resolutions:
                    attn.append(make_attn(block_in, attn_type=attn_type))
            up = nn.Module()
            up.block, up.attn = block, attn
            if lvl != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
                cur *= 2
            ups.insert(0, up)      # index == resolution level, as in the reference (model.py:591-610)
        self.up = nn.ModuleList(ups)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, out_ch, kernel_size=3, stride=1, padding=1)

    def forward(self, z):
        self.last_z_shape = z.shape
        h = self.conv_in(z)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h, None)), None)
        for lvl in reversed(range(self.num_resolutions)):
            for i in range(self.num_res_blocks + 1):
                h = self.up[lvl].block[i](h, None)
                if len(self.up[lvl].attn) > 0:
                    h = self.up[lvl].attn[i](h)
            if lvl != 0:
                h = self.up[lvl].upsample(h)
        if self.give_pre_end:
            return h
        h = self.conv_out(nonlinearity(self.norm_out(h)))
        return torch.tanh(h) if self.tanh_out else h
