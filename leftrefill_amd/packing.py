"""Weight re-layout from the reference's state-dict shapes to the kernels' layouts (done once after loading).

conv  OIHW [Cout, Cin, kh, kw] fp32  ->  [Cout_pad, kh*kw*Cin_pad] fp16, K index = (kh*3 + kw) * Cin_pad + c
linear [out, in]                    ->  [out, in] fp16 (already K-contiguous)
GEGLU proj [8C, C] (u rows then g rows, attention.py:56-57) -> rows interleaved in 16-row groups [u16 | g16 | ...]
"""
import torch


def pad_to(n, m):
    return ((n + m - 1) // m) * m


def pack_conv(w, cin_pad=None, cout_pad=None, dtype=torch.float16):
    cout, cin, kh, kw = w.shape
    cin_pad = cin_pad or pad_to(cin, 64)
    cout_pad = cout_pad or pad_to(cout, 64)
    out = torch.zeros(cout_pad, kh * kw, cin_pad, dtype=dtype, device=w.device)
    out[:cout, :, :cin] = w.permute(0, 2, 3, 1).reshape(cout, kh * kw, cin).to(dtype)
    return out.reshape(cout_pad, kh * kw * cin_pad).contiguous()


def pack_bias(b, n_pad=None):
    n_pad = n_pad or pad_to(b.numel(), 64)
    out = torch.zeros(n_pad, dtype=torch.float32, device=b.device)
    out[: b.numel()] = b.float()
    return out


def pack_linear(w, dtype=torch.float16):
    return w.to(dtype).contiguous()


def geglu_perm(n_half, device=None):
    """Row permutation: packed row p -> original row of the [2*n_half, K] GEGLU projection."""
    assert n_half % 16 == 0
    c = torch.arange(n_half, device=device)
    grp, within = c // 16, c % 16
    perm = torch.empty(2 * n_half, dtype=torch.long, device=device)
    perm[grp * 32 + within] = c                 # u rows
    perm[grp * 32 + 16 + within] = n_half + c   # gate rows
    return perm


def pack_geglu(w, b, dtype=torch.float16):
    n_half = w.shape[0] // 2
    perm = geglu_perm(n_half, w.device)
    return w[perm].to(dtype).contiguous(), b[perm].float().contiguous()


def fold_layernorm(w, b, gamma, beta, dtype=torch.float16):
    """Linear(LayerNorm(x)) as ONE GEMM on the raw x (lr_gemm_args.ln_stats):
        y = rstd * (x @ Wf^T - mean * cs) + bf,   Wf = W * gamma (fp16 / bf16),  cs[n] = sum_k Wf[n][k],  bf = W @ beta + b.
    cs is taken from the ROUNDED Wf so that it cancels exactly what the matrix cores accumulate."""
    w32 = w.float()
    wf = (w32 * gamma.float()[None, :]).to(dtype).contiguous()
    cs = wf.float().sum(dim=1).contiguous()
    bf = (w32 * beta.float()[None, :]).sum(dim=1)     # not `@`: packing may run inside a caller's autocast region (fp16 matmul)
    if b is not None:
        bf = bf + b.float()
    return wf, bf.contiguous(), cs


def xattn_perm(n, device=None):
    """k-slot order of the fused cross-attention block (lr_xattn_block_f16): inside every head of 64 channels, position
    32 p + 8 f + i (f < 4, i < 8) holds channel 32 p + 16 (i >> 2) + 4 f + (i & 3) -- the order in which a 16x16x32 MFMA's
    accumulator lanes hand a tile pair on as the next product's B operand."""
    pos = torch.arange(n, device=device)
    h, r = pos // 64, pos % 64
    p, f, i = r // 32, (r % 32) // 8, r % 8
    return h * 64 + 32 * p + 16 * (i // 4) + 4 * f + (i % 4)


def pack_pieces(w, dtype=torch.float16):
    """[N, K] with K % 64 == 0 -> [K / 64, N, 64]: the 64-column pieces the fused blocks stream through LDS, each piece consecutive
    in memory, the columns inside every piece in k-slot order (xattn_perm)."""
    n, k = w.shape
    perm = xattn_perm(k, w.device)
    return w[:, perm].to(dtype).reshape(n, k // 64, 64).permute(1, 0, 2).contiguous()


def pack_xattn(wk, wo, dtype=torch.float16):
    """(to_k.weight with its ROWS in k-slot order, to_out[0].weight as per-head pieces [heads, C, 64] with k-slot column order)."""
    perm = xattn_perm(wk.shape[0], wk.device)
    return wk[perm].to(dtype).contiguous(), pack_pieces(wo, dtype)


def pack_pm(w):
    """[N, K] packed weights -> piece-major [K / 64, N, 64] (lr_gemm_args.wt_pm): the 128-byte pieces of a K-step are consecutive."""
    N, K = w.shape
    assert K % 64 == 0
    return w.reshape(N, K // 64, 64).permute(1, 0, 2).contiguous()
