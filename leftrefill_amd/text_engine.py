"""OpenCLIP text tower of the prompt encoder on the HIP kernels, inference only.

Replaces `PromptCLIPEmbedder.encode_with_transformer` (reference ldm/modules/encoders/Refill_modules.py:181-201): add the
positional embedding, run the first `n_layers - layer_idx` pre-LN residual attention blocks with the causal mask
(open_clip `ResidualAttentionBlock`: x += out_proj(MHA(ln_1(x))); x += c_proj(gelu(c_fc(ln_2(x))))), then `ln_final`.
Token / special-token embedding lookup and the splice stay torch indexing in the caller (a gather of 77 rows).

The tower is described by duck-typed modules with open_clip's attribute names (`transformer.resblocks[i].{ln_1, attn
(nn.MultiheadAttention: in_proj_weight / in_proj_bias / out_proj), ln_2, mlp.c_fc, mlp.c_proj}`, `positional_embedding`,
`ln_final`), so the packed weights come straight from a loaded open_clip model.  open_clip itself is not in this image:
parity is checked against a PyTorch module of the same published architecture (tests/test_gpu_text.py) -- unpinned by the
reference's own weights / tokenizer.
"""
import torch

from . import ops
from .engine import PackedLinear, PackedNorm


class PackedTextBlock:
    def __init__(self, blk):
        self.ln1, self.ln2 = PackedNorm(blk.ln_1), PackedNorm(blk.ln_2)
        self.qkv = PackedLinear(weight=blk.attn.in_proj_weight, bias=blk.attn.in_proj_bias)
        self.out = PackedLinear(blk.attn.out_proj)
        self.fc, self.proj = PackedLinear(blk.mlp.c_fc), PackedLinear(blk.mlp.c_proj)
        self.heads = blk.attn.num_heads


class PackedTextTower:
    def __init__(self, model, layer_idx=0):
        blocks = list(model.transformer.resblocks)
        self.blocks = [PackedTextBlock(b) for b in blocks[:len(blocks) - layer_idx]]     # "penultimate": drop the last block
        self.pos = model.positional_embedding.detach().float()
        self.ln_final = PackedNorm(model.ln_final)
        self.width = self.pos.shape[1]
        if self.width % 64 or any(self.width // b.heads != 64 for b in self.blocks):
            raise RuntimeError("the attention kernel is specialised for d_head = 64 (ViT-H text tower: width 1024, 16 heads)")


def encode_with_transformer(text_emb, tower: PackedTextTower):
    """text_emb [B, L, width] (token embeddings with the special tokens spliced in) -> [B, L, width] fp32."""
    B, L, D = text_emb.shape
    x = (text_emb.float() + tower.pos[:L]).reshape(B * L, D).to(torch.float16).contiguous()
    M = B * L
    for p in tower.blocks:
        h = ops.layer_norm(x, p.ln1.g, p.ln1.b, p.ln1.eps)
        qkv = ops.gemm_conv(h, p.qkv.w, B=1, H=1, W=M, taps=1, bias=p.qkv.b)
        a = ops.attention_causal(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, p.heads, L, 64 ** -0.5)
        x = ops.gemm_conv(a, p.out.w, B=1, H=1, W=M, taps=1, bias=p.out.b, resid=x)
        h = ops.layer_norm(x, p.ln2.g, p.ln2.b, p.ln2.eps)
        h = ops.gemm_conv(h, p.fc.w, B=1, H=1, W=M, taps=1, bias=p.fc.b, gelu=True)
        x = ops.gemm_conv(h, p.proj.w, B=1, H=1, W=M, taps=1, bias=p.proj.b, resid=x)
    x = ops.layer_norm(x, tower.ln_final.g, tower.ln_final.b, tower.ln_final.eps)
    return x.float().reshape(B, L, D)
