"""Time the KL-VAE decode / encode at 512x1024 (shipped width) on the HIP kernels vs the PyTorch definition of the module."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import leftrefill_amd.dropin as dropin  # noqa: E402

dropin.install()
from ldm.models.autoencoder import AutoencoderKL  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
dd = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4],
          num_res_blocks=2, attn_resolutions=[], dropout=0.0)
m = AutoencoderKL(dd, {"target": "torch.nn.Identity"}, 4).to(dev).eval()
gen = torch.Generator(device=dev).manual_seed(0)
with torch.no_grad():
    for name, p in m.named_parameters():
        if p.dim() >= 2:
            p.copy_(torch.randn(p.shape, device=dev, generator=gen) * (1.0 / p[0].numel()) ** 0.5)
z = torch.randn(B, 4, 64, 128, device=dev, generator=gen)
x = torch.randn(B, 3, 512, 1024, device=dev, generator=gen).clamp(-1, 1)


def timeit(f, n=3):
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    m.use_hip = True
    d_hip = timeit(lambda: m.decode(z))
    e_hip = timeit(lambda: m.encode(x))
    y_hip = m.decode(z)
    if len(sys.argv) > 2 and sys.argv[2] == "hip":
        print(f"B={B} decode {d_hip:.1f} ms  encode {e_hip:.1f} ms")
        sys.exit(0)
    m.use_hip = False
    d_t32 = timeit(lambda: m.decode(z), 1)
    e_t32 = timeit(lambda: m.encode(x), 1)
    y_ref = m.decode(z)
    with torch.autocast("cuda"):
        d_t16 = timeit(lambda: m.decode(z), 1)
        e_t16 = timeit(lambda: m.encode(x), 1)
rel = ((y_hip - y_ref).norm() / y_ref.norm()).item()
# 2*MAC of the conv / attention matmuls per image (decode 4.96 + 0.14 TFLOP, encode 2.16 + 0.14)
print(f"B={B} decode: hip {d_hip:.1f} ms ({B * 5.10 / d_hip:.0f} TFLOP/s)  torch fp32 {d_t32:.1f} ms  torch autocast {d_t16:.1f} ms")
print(f"B={B} encode: hip {e_hip:.1f} ms ({B * 2.30 / e_hip:.0f} TFLOP/s)  torch fp32 {e_t32:.1f} ms  torch autocast {e_t16:.1f} ms")
print(f"decode rel-L2 hip vs torch fp32 at 512x1024: {rel:.3e}")
from leftrefill_amd import ops  # noqa: E402
for k, v in sorted(ops.tile_cache().items(), key=lambda kv: -kv[0][0]):
    print(k[:6], v)
