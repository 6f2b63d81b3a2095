// Fused cross-attention block of BasicTransformerBlock at C = 640 (level 1 of the SD2 UNet: 10 heads of 64; reference
// ldm/modules/attention.py:165-196 + 280-281) -- the register chain of xattn_block.hip (read its header first: operand orders, the
// k-slot order pi, the swapped 16x16x32 MFMA form) re-cut for rows that are twice as wide:
//
//     [x1 = a Wo1^T + bo1 + x]      out = x1 + to_out( softmax( (LayerNorm(x1) Wq^T) K^T * scale ) V )
//
// Work split.  The chain is row-local, and the 16384 rows of level 1 fill 256 CUs only with 64-row blocks: block = 4 waves = 64 rows
// of ONE sample, one wave per SIMD with the whole 512-register file: a wave keeps its 16 rows as B operands (80 registers), the 40
// output tiles out^T [640 x 16] (160), and with PRE the rounded x1 as residual (80).
// Weights stream through the same 3-slot ring of 40 KB pieces (16-byte LDS-DMA, source-side XOR swizzle, loads two pieces ahead,
// counted vmcnt + one barrier per piece); a [64 x 640] projection slice is two pieces (k halves), Wo_h [640 x 64] two pieces (n halves),
// K_h and V_h^T one 16 KB piece each -- six pieces per head, so the slot of every piece is a compile-time constant:
//     Q0 -> slot 0 | Q1 -> 1 | K -> 2 | V -> 0 | O0 -> 1 | O1 -> 2          (PRE: 20 pieces of Wo1 in front, piece j -> slot (j + 1) % 3)
// Every block streams 3 x 640 x 640 weights + K / V = 2.6 MB from its XCD's L2; that stream (not the matrix pipe) bounds the kernel
// (tools/micro/chain_stream_proto.hip, profiles/r04_chain_proto.txt).
#include "chain_common.h"

#define XB_C 640
#define XB_HEADS 10
#define XB_ROWS 64
#define XB_THREADS 256
#define XB_SLOT 40960
#define XB_PITCH 1296         // bytes per staged output row (1280 + 16: the 16 rows of a wave start in distinct banks)

struct Xattn640Params {
  const void* x; const void* wq; const float* bq; const void* k; const void* vt; const void* wo; const float* bo;
  void* out; float* st_out;
  const void* pre_a; const void* pre_w; const float* pre_b;
  int M, HW, Lc, ldk, nblocks;
  float eps, c;               // c = scale * log2(e)
  unsigned k_bytes, vt_bytes;
};

// NKT = 16-key tiles of the context (5: Lc <= 80, 6: Lc <= 96); PRE: see xattn_block.hip
template <typename T, int NKT, bool PRE>
__global__ __launch_bounds__(XB_THREADS) void xattn640_kernel(const Xattn640Params P) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int C = XB_C, NT = C / 16, KL = C / 64;   // 40 output tiles, 10 lines of 128 B per row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* par = reinterpret_cast<float*>(smem + 3 * XB_SLOT);      // [3][C]: bq | bo | bo1 (PRE)

  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  int bid = blockIdx.x;
  {   // XCD-aware bijective remap: consecutive row blocks (one sample's K / V) stay on one XCD's L2
    const int q = P.nblocks >> 3, r = P.nblocks & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int m0 = bid * XB_ROWS;
  const int b = m0 / P.HW;
#ifndef XB_ROTATE      // measured (profiles/r06_rotation_ab.txt): 61.9 vs 61.5 us plain, 71.1 vs 73.8 us with PRE -- within the run-to-run spread; off
  const int h0 = 0;
#else
  // first head of this block's cyclic head order (see the head loop): a function of the block's position INSIDE its sample, so a row's
  // bits do not depend on the batch size or on which sample / rank carries it (consecutive blocks run on the same XCD after the remap)
  const int h0 = ((m0 - b * P.HW) / XB_ROWS) % XB_HEADS;
#endif
  const int m_w0 = m0 + w * 16;

  // ---- this wave's 16 rows in B-operand form: lane (fr, fq) holds x[m_w0 + fr][64 t5 + 32 u + 8 fq .. + 7]
  const T* xrow = reinterpret_cast<const T*>(PRE ? P.pre_a : P.x) + (size_t)(m_w0 + fr) * C + 8 * fq;
  vec8<T> xf[KL][2];
#pragma unroll
  for (int t5 = 0; t5 < KL; ++t5)
#pragma unroll
    for (int u = 0; u < 2; ++u) xf[t5][u] = *reinterpret_cast<const vec8<T>*>(xrow + 64 * t5 + 32 * u);
  // PRE: the residual x in ACCUMULATOR layout (lane (fr, fq), tile j: x[m_w0 + fr][16 j + 4 fq .. + 3]); later the rounded x1
  vec4<T> xres[PRE ? NT : 1];
  if constexpr (PRE) {
    const T* xd = reinterpret_cast<const T*>(P.x) + (size_t)(m_w0 + fr) * C + 4 * fq;
#pragma unroll
    for (int j = 0; j < NT; ++j) xres[j] = *reinterpret_cast<const vec4<T>*>(xd + 16 * j);
  }
#pragma unroll
  for (int i0 = 0; i0 < (PRE ? 3 : 2) * (C / 4); i0 += XB_THREADS) {      // biases -> LDS (the loops below issue no register loads)
    const int i = i0 + t;
    if (i < (PRE ? 3 : 2) * (C / 4)) {
      const float* src = i < C / 4 ? P.bq + 4 * i : i < C / 2 ? P.bo + 4 * (i - C / 4) : P.pre_b + 4 * (i - C / 2);
      *reinterpret_cast<f32x4*>(par + 4 * i) = *reinterpret_cast<const f32x4*>(src);
    }
  }

  // ---- weight ring: every piece is a set of 1 KiB LDS-DMA instructions, 32 rows of 128 B per block-wide round (wave w: rows 8 w .. + 7)
  const __amdgpu_buffer_rsrc_t rsQ = uniform_rsrc(P.wq, (size_t)C * C * 2);
  const __amdgpu_buffer_rsrc_t rsO = uniform_rsrc(P.wo, (size_t)C * C * 2);
  const __amdgpu_buffer_rsrc_t rsK = uniform_rsrc(P.k, P.k_bytes);
  const __amdgpu_buffer_rsrc_t rsV = uniform_rsrc(P.vt, P.vt_bytes);
  const __amdgpu_buffer_rsrc_t rsP = uniform_rsrc(PRE ? P.pre_w : P.wq, (size_t)C * C * 2);
  const unsigned OOB = 0x80000000u;
  const int lrow = w * 8 + (lane >> 3);                      // row of a 32-row round this lane fills
  const int lchunk = (lane & 7) ^ ((lrow >> 1) & 7);         // source-side swizzle (bits 1..3 of the row: same for row + 32 i)
  // [64 rows x 320 k] of a [C][C] matrix (rows r0 .., columns c0 ..) as 5 sub-tiles [64 x 64 k]; instruction i = 0 .. 9: sub-tile i >> 1, row half i & 1
  // (live = false: an out-of-range offset -- zeros land in the slot, nobody reads them; keeps the loop free of branches and the counted
  // vmcnt waits uniform over the heads)
  auto issue_mat = [&](const __amdgpu_buffer_rsrc_t rs, int slot, int r0, int c0, int i, bool live = true) __attribute__((always_inline)) {
    const unsigned v0 = live ? (unsigned)(((r0 + lrow) * C + lchunk * 8) * 2) : OOB;
    const int t5 = i >> 1, hh = i & 1;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(smem + slot * XB_SLOT + (t5 * 64 + hh * 32 + w * 8) * 128), 16, v0,
                                             (hh * 32 * C + c0 + t5 * 64) * 2, 0, 0);
  };
  auto issue_k = [&](int slot, int h, int i) __attribute__((always_inline)) {    // K_h [128 keys x 64 d]; i = 0 .. 3: keys 32 i + lrow
    const int key = i * 32 + lrow;
    const unsigned vk = key < P.Lc ? (unsigned)((((size_t)b * P.Lc + key) * P.ldk + h * 64 + lchunk * 8) * 2) : OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lptr_t)(smem + slot * XB_SLOT + (i * 32 + w * 8) * 128), 16, vk, 0, 0, 0);
  };
  auto issue_v = [&](int slot, int h, int i) __attribute__((always_inline)) {    // V_h^T 2 x [64 d x 64 key slots]; i = 0 .. 3: rows 32 i + lrow of the pair
    const unsigned vv = (unsigned)((((b * XB_HEADS + h) * 2) * 64 + lrow) * 128 + lchunk * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lptr_t)(smem + slot * XB_SLOT + (i * 32 + w * 8) * 128), 16, vv, i * 32 * 128, 0, 0);
  };
  auto issue_wo = [&](int slot, int h, int nh, int i) __attribute__((always_inline)) {   // rows 320 nh + 32 i + lrow of piece h of Wo ([heads][640 n][64 k])
    const unsigned v0 = (unsigned)(((h * C + lrow) * 64 + lchunk * 8) * 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsO, (lptr_t)(smem + slot * XB_SLOT + (i * 32 + w * 8) * 128), 16, v0, (nh * 320 + i * 32) * 128, 0, 0);
  };
  if constexpr (PRE) {      // pieces 0, 1 of Wo1 -> slots 1, 2
#pragma unroll
    for (int i = 0; i < 10; ++i) issue_mat(rsP, 1, 0, 0, i);
#pragma unroll
    for (int i = 0; i < 10; ++i) issue_mat(rsP, 2, 0, 320, i);
  } else {                  // the first head's query projection slice
#pragma unroll
    for (int i = 0; i < 10; ++i) issue_mat(rsQ, 0, h0 * 64, 0, i);
#pragma unroll
    for (int i = 0; i < 10; ++i) issue_mat(rsQ, 1, h0 * 64, 320, i);
  }

  // ---- LayerNorm of the rows in registers (two-pass), gamma / beta live in Wq / bq
  if constexpr (!PRE) {
    float s = 0.f;
#pragma unroll
    for (int t5 = 0; t5 < KL; ++t5)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) s += (float)xf[t5][u][i];
    const float mean = xa_row4_sum(s) * (1.0f / C);
    float q2 = 0.f;
#pragma unroll
    for (int t5 = 0; t5 < KL; ++t5)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = (float)xf[t5][u][i] - mean; q2 = fmaf(d, d, q2); }
    const float rstd = rsqrtf(xa_row4_sum(q2) * (1.0f / C) + P.eps);
    const float nmr = -mean * rstd;
#pragma unroll
    for (int t5 = 0; t5 < KL; ++t5)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) xf[t5][u][i] = (T)fmaf((float)xf[t5][u][i], rstd, nmr);
  }

  f32x4 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const int sw = (fr >> 1) & 7;      // swizzle of every fragment row this lane reads (rows are fr + a multiple of 16)
  auto frag = [&](const char* base, int row, int chunk) -> vec8<T> {
    return *reinterpret_cast<const vec8<T>*>(base + row * 128 + ((chunk ^ sw) << 4));
  };
#define XB_FENCE() __builtin_amdgcn_sched_barrier(0)
  // one [64 x 320 k] piece against the B operands xb[0 .. 4][0 .. 1]: 10 k-steps of 4 MFMAs into a0 .. a3, fragment reads one step ahead,
  // one LDS-DMA instruction of a later piece after each step (dma(ks))
#define XB_MAT_STEP(Ws, XB0, A0, A1, A2, A3, DMA)                                                                              \
  {                                                                                                                             \
    vec8<T> fa[2][4];                                                                                                           \
    auto rd = [&](int ks, vec8<T> (&f)[4]) __attribute__((always_inline)) {                                                     \
      _Pragma("unroll") for (int jd = 0; jd < 4; ++jd) f[jd] = frag((Ws) + (ks >> 1) * 64 * 128, jd * 16 + fr, 4 * (ks & 1) + fq); \
    };                                                                                                                          \
    rd(0, fa[0]);                                                                                                               \
    _Pragma("unroll") for (int ks = 0; ks < 10; ++ks) {                                                                         \
      if (ks + 1 < 10) rd(ks + 1, fa[(ks + 1) & 1]);                                                                            \
      XB_FENCE();                                                                                                               \
      A0 = lr_mfma16(fa[ks & 1][0], xf[(XB0) + (ks >> 1)][ks & 1], A0);                                                         \
      A1 = lr_mfma16(fa[ks & 1][1], xf[(XB0) + (ks >> 1)][ks & 1], A1);                                                         \
      A2 = lr_mfma16(fa[ks & 1][2], xf[(XB0) + (ks >> 1)][ks & 1], A2);                                                         \
      A3 = lr_mfma16(fa[ks & 1][3], xf[(XB0) + (ks >> 1)][ks & 1], A3);                                                         \
      DMA;                                                                                                                      \
      XB_FENCE();                                                                                                               \
    }                                                                                                                           \
  }

  if constexpr (PRE) {
    // ================= pre steps: x1^T = Wo1 a^T, 64 output channels (4 tiles) x one k half per piece =======================
#pragma unroll
    for (int j = 0; j < 2 * KL; ++j) {
      const int p = j >> 1, kh = j & 1, slot = (j + 1) % 3;
      xa_wait_vmcnt<10>();                    // piece j landed (the next one may be in flight)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const char* Ws = smem + slot * XB_SLOT;
      // two pieces ahead: Wo1 pieces 2 .. 19, then the first head's two Wq pieces
      XB_MAT_STEP(Ws, 5 * kh, acc[4 * p], acc[4 * p + 1], acc[4 * p + 2], acc[4 * p + 3],
                  { if (j + 2 < 2 * KL) issue_mat(rsP, (j + 3) % 3, 64 * ((j + 2) >> 1), 320 * ((j + 2) & 1), ks);
                    else issue_mat(rsQ, j + 2 - 2 * KL, h0 * 64, 320 * (j + 2 - 2 * KL), ks); });
    }
    // x1 = acc + bo1 + x, rounded to fp16 (what the unfused path stores); LayerNorm of the rounded rows; B operands in accumulator order
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const f32x4 bj = *reinterpret_cast<const f32x4*>(par + 2 * C + j * 16 + 4 * fq);
      vec4<T> r16;
#pragma unroll
      for (int r = 0; r < 4; ++r) { r16[r] = (T)(acc[j][r] + bj[r] + (float)xres[j][r]); s += (float)r16[r]; }
      xres[j] = r16;
    }
    const float mean = xa_row4_sum(s) * (1.0f / C);
    float q2 = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float d = (float)xres[j][r] - mean; q2 = fmaf(d, d, q2); }
    const float rstd = rsqrtf(xa_row4_sum(q2) * (1.0f / C) + P.eps);
    const float nmr = -mean * rstd;
#pragma unroll
    for (int pp = 0; pp < NT / 2; ++pp) {
      f32x4 n0, n1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        n0[r] = fmaf((float)xres[2 * pp][r], rstd, nmr);
        n1[r] = fmaf((float)xres[2 * pp + 1][r], rstd, nmr);
      }
      xf[pp >> 1][pp & 1] = xa_pack<T>(n0, n1);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  // The heads are independent terms of out^T: block b starts at head h0 = (b / 8) mod 10 and wraps, so the ~32 blocks an XCD runs at once
  // stream DIFFERENT weight pieces from its L2 at any moment instead of all asking for the same 40 KB.
#pragma unroll 1
  for (int hi = 0; hi < XB_HEADS; ++hi) {
    const int h = hi + h0 < XB_HEADS ? hi + h0 : hi + h0 - XB_HEADS;
    const int hn = h + 1 < XB_HEADS ? h + 1 : 0;      // the head after this one
    const bool more = hi + 1 < XB_HEADS;
    // ================= Q0 / Q1: q_h^T = Wq_h xn^T  (two k halves of 10 k-steps x 4 MFMAs) =================================
    f32x4 qa[4] = {z4, z4, z4, z4};
    xa_wait_vmcnt<10>();                      // Q0 landed (Q1 in flight)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (first head: the bias rows written to LDS above)
    __builtin_amdgcn_s_barrier();
    XB_MAT_STEP(smem, 0, qa[0], qa[1], qa[2], qa[3], { if (ks < 4) issue_k(2, h, ks); });           // K_h -> slot 2
    xa_wait_vmcnt<4>();                       // Q1 landed (K_h in flight)
    __builtin_amdgcn_s_barrier();
    XB_MAT_STEP(smem + XB_SLOT, 5, qa[0], qa[1], qa[2], qa[3], { if (ks < 4) issue_v(0, h, ks); }); // V_h^T -> slot 0
#pragma unroll
    for (int jd = 0; jd < 4; ++jd) qa[jd] += *reinterpret_cast<const f32x4*>(par + h * 64 + jd * 16 + 4 * fq);
    const vec8<T> qb0 = xa_pack<T>(qa[0], qa[1]), qb1 = xa_pack<T>(qa[2], qa[3]);

    // ================= K: S^T = K_h q_h^T, softmax ===================================================================
    xa_wait_vmcnt<4>();                       // K_h landed (V_h^T in flight)
    __builtin_amdgcn_s_barrier();
    f32x4 sa[NKT];
    {
      const char* Ks = smem + 2 * XB_SLOT;
      vec8<T> k0[NKT], k1[NKT];
#pragma unroll
      for (int jk = 0; jk < NKT; ++jk) k0[jk] = frag(Ks, jk * 16 + fr, fq);
#pragma unroll
      for (int jk = 0; jk < NKT; ++jk) k1[jk] = frag(Ks, jk * 16 + fr, 4 + fq);
      XB_FENCE();
#pragma unroll
      for (int jk = 0; jk < NKT; ++jk) sa[jk] = lr_mfma16(k0[jk], qb0, z4);
      issue_wo(1, h, 0, 0); issue_wo(1, h, 0, 1);                      // Wo_h rows 0 .. 319 -> slot 1
      XB_FENCE();
#pragma unroll
      for (int jk = 0; jk < NKT; ++jk) sa[jk] = lr_mfma16(k1[jk], qb1, sa[jk]);
      issue_wo(1, h, 0, 2); issue_wo(1, h, 0, 3);
      XB_FENCE();
    }
    float mx = -INFINITY;
#pragma unroll
    for (int jk = 0; jk < NKT; ++jk)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (jk * 16 + 4 * fq + r >= P.Lc) sa[jk][r] = -INFINITY;
        mx = fmaxf(mx, sa[jk][r]);
      }
    issue_wo(1, h, 0, 4); issue_wo(1, h, 0, 5); issue_wo(1, h, 0, 6);
    mx = xa_row4_max(mx);
    const float mc = mx * P.c;
    float l = 0.f;
#pragma unroll
    for (int jk = 0; jk < NKT; ++jk)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(sa[jk][r], P.c, -mc));
        sa[jk][r] = p;
        l += p;
      }
    issue_wo(1, h, 0, 7); issue_wo(1, h, 0, 8); issue_wo(1, h, 0, 9);
    l = xa_row4_sum(l);
    vec8<T> pb[3];
    pb[0] = xa_pack<T>(sa[0], sa[1]);
    pb[1] = xa_pack<T>(sa[2], sa[3]);
    pb[2] = xa_pack<T>(sa[4], NKT > 5 ? sa[NKT - 1] : z4);

    // ================= V: O_h^T = V_h^T P^T ============================================================================
    xa_wait_vmcnt<10>();                      // V_h^T landed (Wo_h rows 0 .. 319 in flight)
    __builtin_amdgcn_s_barrier();
    f32x4 oa[4];
    {
      const char* Vs = smem;
      vec8<T> v0[4], v1[4], v2[4];
#pragma unroll
      for (int jd = 0; jd < 4; ++jd) v0[jd] = frag(Vs, jd * 16 + fr, fq);
#pragma unroll
      for (int jd = 0; jd < 4; ++jd) v1[jd] = frag(Vs, jd * 16 + fr, 4 + fq);
#pragma unroll
      for (int jd = 0; jd < 4; ++jd) v2[jd] = frag(Vs + 64 * 128, jd * 16 + fr, fq);
      XB_FENCE();
#pragma unroll
      for (int jd = 0; jd < 4; ++jd) oa[jd] = lr_mfma16(v0[jd], pb[0], z4);
      issue_wo(2, h, 1, 0); issue_wo(2, h, 1, 1); issue_wo(2, h, 1, 2);      // Wo_h rows 320 .. 639 -> slot 2
      XB_FENCE();
#pragma unroll
      for (int jd = 0; jd < 4; ++jd) oa[jd] = lr_mfma16(v1[jd], pb[1], oa[jd]);
      issue_wo(2, h, 1, 3); issue_wo(2, h, 1, 4); issue_wo(2, h, 1, 5);
      XB_FENCE();
#pragma unroll
      for (int jd = 0; jd < 4; ++jd) oa[jd] = lr_mfma16(v2[jd], pb[2], oa[jd]);
      issue_wo(2, h, 1, 6); issue_wo(2, h, 1, 7); issue_wo(2, h, 1, 8); issue_wo(2, h, 1, 9);
      XB_FENCE();
    }
    const float inv = __builtin_amdgcn_rcpf(l);
#pragma unroll
    for (int jd = 0; jd < 4; ++jd) oa[jd] *= inv;
    const vec8<T> ob0 = xa_pack<T>(oa[0], oa[1]), ob1 = xa_pack<T>(oa[2], oa[3]);

    // ================= O0 / O1: out^T += Wo_h O_h^T  (two n halves of 10 groups x 4 MFMAs) =================================
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
      xa_wait_vmcnt<10>();                    // this half landed (the next piece in flight)
      __builtin_amdgcn_s_barrier();
      const char* Os = smem + (1 + nh) * XB_SLOT;
      vec8<T> fa[2][4];
      // group g: k half g / 5, output tiles 20 nh + 4 (g % 5) .. + 3
      auto rd = [&](int g, vec8<T> (&f)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) f[q] = frag(Os, (4 * (g % 5) + q) * 16 + fr, 4 * (g / 5) + fq);
      };
      rd(0, fa[0]);
#pragma unroll
      for (int g = 0; g < 10; ++g) {
        if (g + 1 < 10) rd(g + 1, fa[(g + 1) & 1]);
        XB_FENCE();
#pragma unroll
        for (int q = 0; q < 4; ++q)
          acc[20 * nh + 4 * (g % 5) + q] = lr_mfma16(fa[g & 1][q], g < 5 ? ob0 : ob1, acc[20 * nh + 4 * (g % 5) + q]);
        issue_mat(rsQ, nh, hn * 64, 320 * nh, g, more);      // Wq of the next head, k half nh -> slot nh
        XB_FENCE();
      }
    }
  }
#undef XB_FENCE
#undef XB_MAT_STEP

  // ---- epilogue: (acc + bias) -> fp16 -> this wave's 16 LDS rows -> 16-byte pieces: + x, store, row statistics
  xa_wait_vmcnt<0>();                         // (the last head's dead prefetches)
  __syncthreads();                            // every wave is done with the ring
  char* stg = smem + w * (16 * XB_PITCH);
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const f32x4 v = acc[j] + *reinterpret_cast<const f32x4*>(par + C + j * 16 + 4 * fq);
    vec4<T> hv;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      hv[r] = (T)v[r];
      if constexpr (PRE) hv[r] = (T)((float)hv[r] + (float)xres[j][r]);      // + x1 (same two roundings as the unfused kernels)
    }
    *reinterpret_cast<vec4<T>*>(stg + fr * XB_PITCH + (j * 16 + 4 * fq) * 2) = hv;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int row = lane >> 2, sub = lane & 3;
  const T* xr = reinterpret_cast<const T*>(P.x) + (size_t)(m_w0 + row) * C;
  T* orow = reinterpret_cast<T*>(P.out) + (size_t)(m_w0 + row) * C;
  float s1 = 0.f, s2 = 0.f;
  constexpr int NP = C / 32;                  // 16-byte pieces per lane (4 lanes per row)
  uint4 rx[PRE ? 1 : NP];
  if constexpr (!PRE) {
#pragma unroll
    for (int it = 0; it < NP; ++it) rx[it] = *reinterpret_cast<const uint4*>(xr + (sub + 4 * it) * 8);
  }
#pragma unroll
  for (int it = 0; it < NP; ++it) {
    const int piece = sub + 4 * it;
    float a[8], e[8];
    uint4 pk = *reinterpret_cast<const uint4*>(stg + row * XB_PITCH + piece * 16);
    lr_unpack8<T>(pk, a);
    if constexpr (!PRE) {
      lr_unpack8<T>(rx[it], e);
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] += e[i];
      pk = lr_pack8<T>(a);
    }
    *reinterpret_cast<uint4*>(orow + piece * 8) = pk;
    lr_unpack8<T>(pk, a);
#pragma unroll
    for (int i = 0; i < 8; ++i) { s1 += a[i]; s2 = fmaf(a[i], a[i], s2); }
  }
  if (P.st_out) {
    s1 += __shfl_xor(s1, 1, 64); s2 += __shfl_xor(s2, 1, 64);
    s1 += __shfl_xor(s1, 2, 64); s2 += __shfl_xor(s2, 2, 64);
    if (sub == 0) {
      float2 o; o.x = s1; o.y = s2;
      *reinterpret_cast<float2*>(P.st_out + (size_t)(m_w0 + row) * 2) = o;
    }
  }
#endif
}

template <typename T>
static int xattn640_t(const lr_xattn_args* a, lr_stream_t s) {
  if (a->M % XB_ROWS || a->HW % XB_ROWS || a->M % a->HW) return LR_E_UNSUPPORTED;     // a block stays inside one sample
  const int B = a->M / a->HW;
  const int64_t kb = (int64_t)B * a->Lc * a->ldk * 2, vb = (int64_t)B * XB_HEADS * 2 * 64 * 128;
  if (kb >= ((int64_t)1 << 31) || vb >= ((int64_t)1 << 31)) return LR_E_UNSUPPORTED;
  Xattn640Params P;
  P.x = a->x; P.wq = a->wq; P.bq = a->bq; P.k = a->k; P.vt = a->vt; P.wo = a->wo; P.bo = a->bo; P.out = a->out;
  P.st_out = a->stats_out;
  P.M = a->M; P.HW = a->HW; P.Lc = a->Lc; P.ldk = a->ldk; P.nblocks = a->M / XB_ROWS;
  P.eps = a->ln_eps; P.c = a->scale * 1.44269504088896340736f;
  P.k_bytes = (unsigned)kb; P.vt_bytes = (unsigned)vb;
  P.pre_a = a->pre_a; P.pre_w = a->pre_w; P.pre_b = a->pre_b;
  const size_t smem = 3 * XB_SLOT + 3 * XB_C * sizeof(float);
  const bool six = a->Lc > 80, pre = a->pre_a != nullptr;
  const void* fns[4] = {reinterpret_cast<const void*>(xattn640_kernel<T, 5, false>), reinterpret_cast<const void*>(xattn640_kernel<T, 6, false>),
                        reinterpret_cast<const void*>(xattn640_kernel<T, 5, true>), reinterpret_cast<const void*>(xattn640_kernel<T, 6, true>)};
  static unsigned long long attr_done[4] = {0, 0, 0, 0};
  const int v = (pre ? 2 : 0) + (six ? 1 : 0);
  if (lr_attr_needed(&attr_done[v])) hipFuncSetAttribute(fns[v], hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  const dim3 grid(P.nblocks), block(XB_THREADS);
  if (v == 0) hipLaunchKernelGGL((xattn640_kernel<T, 5, false>), grid, block, smem, (hipStream_t)s, P);
  else if (v == 1) hipLaunchKernelGGL((xattn640_kernel<T, 6, false>), grid, block, smem, (hipStream_t)s, P);
  else if (v == 2) hipLaunchKernelGGL((xattn640_kernel<T, 5, true>), grid, block, smem, (hipStream_t)s, P);
  else hipLaunchKernelGGL((xattn640_kernel<T, 6, true>), grid, block, smem, (hipStream_t)s, P);
  return lr_launch_status();
}

// called by lr_xattn_block_f16 / _bf16 (xattn_block.hip) after the shared argument checks, for C = 640 / heads = 10
int lr_xattn640_launch(const lr_xattn_args* a, int bf16_, lr_stream_t s) {
  return bf16_ ? xattn640_t<bf16>(a, s) : xattn640_t<f16>(a, s);
}
