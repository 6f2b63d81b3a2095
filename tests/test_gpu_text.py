"""Prompt-encoder text tower on the HIP kernels (causal attention, GELU-epilogue GEMM) vs a PyTorch module of the published
OpenCLIP architecture (nn.MultiheadAttention residual blocks with the causal attn_mask).  open_clip is not installed in
this image; the reference's own PromptCLIPEmbedder code is pinned through oracle/clip_stub.py (a stand-in for the package's
surface) and tests/golden/text.npz -- see test_prompt_embedder_on_hip_matches_reference_goldens and tests/test_text_cpu.py."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

from oracle import golden_spec as G, weights  # noqa: E402


class _Block(nn.Module):          # open_clip.transformer.ResidualAttentionBlock (pre-LN), attribute names as open_clip
    def __init__(self, d, heads):
        super().__init__()
        self.ln_1 = nn.LayerNorm(d)
        self.attn = nn.MultiheadAttention(d, heads)
        self.ln_2 = nn.LayerNorm(d)
        self.mlp = nn.Sequential()
        self.mlp.add_module("c_fc", nn.Linear(d, 4 * d))
        self.mlp.add_module("gelu", nn.GELU())
        self.mlp.add_module("c_proj", nn.Linear(4 * d, d))

    def forward(self, x, attn_mask):
        h = self.ln_1(x)
        x = x + self.attn(h, h, h, need_weights=False, attn_mask=attn_mask)[0]
        return x + self.mlp(self.ln_2(x))


class _Tower(nn.Module):
    def __init__(self, d=256, heads=4, layers=3, ctx=77):
        super().__init__()
        self.transformer = nn.Module()
        self.transformer.resblocks = nn.ModuleList([_Block(d, heads) for _ in range(layers)])
        self.positional_embedding = nn.Parameter(torch.zeros(ctx, d))
        self.ln_final = nn.LayerNorm(d)
        m = torch.full((ctx, ctx), float("-inf"))
        self.register_buffer("attn_mask", m.triu_(1), persistent=False)

    def encode(self, emb, layer_idx):      # Refill_modules.py:181-201
        x = emb + self.positional_embedding
        x = x.permute(1, 0, 2)
        for i, r in enumerate(self.transformer.resblocks):
            if i == len(self.transformer.resblocks) - layer_idx:
                break
            x = r(x, self.attn_mask)
        return self.ln_final(x.permute(1, 0, 2))


@pytest.mark.parametrize("layer_idx", [0, 1], ids=["last", "penultimate"])
def test_text_tower_matches_torch_restatement(layer_idx):
    from leftrefill_amd import text_engine
    d = torch.device("cuda:0")
    m = _Tower().eval()
    m.load_state_dict({k: torch.from_numpy(weights.fill_like("txt." + k, v.shape)).half().float()
                       for k, v in m.state_dict().items()})
    emb = G.T("txt.emb", (3, 77, 256)).half().float()
    with torch.no_grad():
        ref = m.encode(emb, layer_idx)
        md = m.to(d)
        out = text_engine.encode_with_transformer(emb.to(d), text_engine.PackedTextTower(md, layer_idx))
    err = (out.cpu() - ref).abs()
    rel = (err.norm() / ref.norm()).item()
    print(f"[text tower layer_idx={layer_idx}] rel_l2 {rel:.3e} max_abs {err.max().item():.3e} (|ref| max {ref.abs().max().item():.2f})")
    assert torch.isfinite(out).all() and rel < 3e-3 and err.max().item() < 3e-2


@pytest.mark.parametrize("name,kw,prompts", G.TEXT_CASES, ids=[c[0] for c in G.TEXT_CASES])
def test_prompt_embedder_on_hip_matches_reference_goldens(name, kw, prompts):
    """Whole prompt encoder on the GPU (tokenise -> splice -> HIP text tower) against tests/golden/text.npz: outputs of the
    REFERENCE's PromptCLIPEmbedder run on oracle/clip_stub.py (the stand-in for the un-vendored open_clip package)."""
    import os
    import sys
    import numpy as np
    import leftrefill_amd.dropin as dropin
    from oracle import clip_stub
    dropin.install()
    sys.modules["open_clip"] = clip_stub
    try:
        from ldm.modules.encoders.Refill_modules import PromptCLIPEmbedder
        emb = PromptCLIPEmbedder(device="cuda", **kw).to("cuda:0").eval()
        with torch.no_grad():
            z = emb(prompts)
    finally:
        sys.modules.pop("open_clip", None)
    assert getattr(emb, "_lr_tower", None) is not None, "the HIP tower must have run"
    ref = torch.from_numpy(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "text.npz"))[name + ".z"])
    err = (z.float().cpu() - ref).abs()
    rel = (err.norm() / ref.norm()).item()
    print(f"[prompt encoder {name}] rel_l2 {rel:.3e} max_abs {err.max().item():.3e} (|ref| max {ref.abs().max().item():.2f})")
    assert z.shape == ref.shape and torch.isfinite(z).all() and rel < 3e-3 and err.max().item() < 3e-2


def test_causal_attention_kernel():
    from leftrefill_amd import ops
    d = torch.device("cuda:0")
    for B, heads, N in ((2, 4, 77), (1, 2, 64), (1, 1, 200)):
        C = heads * 64
        qkv = G.T(f"caus.{N}", (B * N, 3 * C)).half()
        q, k, v = (qkv[:, i * C:(i + 1) * C].float().reshape(B, N, heads, 64).permute(0, 2, 1, 3) for i in range(3))
        ref = torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=True, scale=0.125)
        ref = ref.permute(0, 2, 1, 3).reshape(B * N, C)
        g = qkv.to(d)
        out = ops.attention_causal(g[:, :C], g[:, C:2 * C], g[:, 2 * C:], B, heads, N, 0.125)
        assert (out.float().cpu() - ref).abs().max().item() < 2e-3


def test_gemm_gelu_epilogue():
    from leftrefill_amd import ops
    d = torch.device("cuda:0")
    M, K, N = 231, 256, 1024
    x = G.T("gelu.x", (M, K)).half()
    w = torch.from_numpy(weights.fill_like("gelu.w", (N, K))).half()
    b = torch.from_numpy(weights.fill_like("gelu.b", (N,)))
    r = G.T("gelu.r", (M, N)).half()
    ref = torch.nn.functional.gelu(x.float() @ w.float().t() + b) + r.float()
    for splits in (1, 2):
        out = ops.gemm_conv(x.to(d), w.to(d), B=1, H=1, W=M, taps=1, bias=b.to(d), resid=r.to(d), gelu=True, splits=splits)
        assert torch.allclose(out.float().cpu(), ref, rtol=2e-3, atol=2e-3)
