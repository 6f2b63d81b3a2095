// Row-resident LayerNorm + Linear at C = 640 (level 1 of the SD2 UNet), for the two wide projections of a BasicTransformerBlock whose K loop
// is only 10 K-steps long (reference ldm/modules/attention.py:280 + 168-172: norm1 + attn1.to_q / to_k / to_v, fused [3C] wide; 282 + 51-58:
// norm3 + GEGLU.proj with `x * F.gelu(gate)`):
//
//     out = LayerNorm(x) W^T + b                       plain   (W = the LayerNorm-folded fused q|k|v projection, N = 1920)
//     out = u * gelu_erf(g),  [u | g] = LN(x) W^T + b  GEGLU   (W's rows interleaved [u16 | g16 | ...] like lr_gemm_args.geglu, N = 5120 -> 2560 columns)
//
// The tiled GEMM runs these as 256 x 320 tiles of a 10-step loop: each tile fills its ring, multiplies for 38 % of its time and drains through
// an epilogue (erf for the gate) with the matrix pipes idle, four tiles per CU one after the other (profiles/r05_conv_halo_proto.txt section 6).
// Here, like stin_block.hip at C = 320: a block keeps 128 rows in registers as B operands of swapped 16x16x32 MFMAs (8 waves x 16 rows, the
// LayerNorm two-pass in registers) and streams ITS SLICE of the weight rows -- grid = (M / 128) row blocks x ny column slices, 256 blocks at
// M = 16384 with ny = 2 -- through a 3-slot ring of 40 KB steps (64 weight rows x one k half; two steps per 64-column piece), one barrier per
// step, no fill / drain between pieces; a piece's four accumulator tiles are emitted (bias, gate, 16 bits, permlane swap, 16-byte stores)
// WHILE the next piece multiplies.  Weight bytes through the LDS are the same as the tiled GEMM's (every block streams its slice once per
// 128 rows); what goes away are the per-tile bubbles.
#include "chain_common.h"

#define RL_ROWS 128
#define RL_THREADS 512
#define RL_SLOT 40960
#ifndef RL_DMA_AUX
#define RL_DMA_AUX 0           // cache policy of the weight stream's LDS-DMA (nt = 2 measured: 59.0 vs 54.0 / 136.6 vs 136.3 us: no gain)
#endif
#ifndef RL_DEPTH
#define RL_DEPTH 1             // k-steps a fragment read runs ahead of its MFMAs (measured at M = 16384: depth 1 / 2 / 3 / 4 = 50.6 / 51.2 / 51.6 / 53.1 us
#endif                         // for q|k|v, 119.6 / 128.6 / 130.4 / 130.7 us for GEGLU -- the read latency is not what the loop waits for)
#ifndef RL_W4_DEFAULT
#define RL_W4_DEFAULT 0
#endif
#define RL_MAX_SLICE 2560      // weight rows per block whose bias fits the LDS region behind the ring

struct RowlinParams {
  const void* x; const void* w; const float* bias; void* out;
  int M, N, ld_out, ny, np;    // np = 64-row pieces per block = N / (64 ny)
  float eps;
  // GN (plain Linear only): x holds RAW tokens, the SpatialTransformer's GroupNorm(32) is applied to the rows as they are loaded (stin_block.hip)
  const float* gn_part; const float* gn_gamma; const float* gn_beta; int gn_chunks, gn_hw; float gn_eps;
};

// C = 320 | 640 | 1280 (levels 0 / 1 / 2: one / two / four 320-k steps per 64-column piece); LN = false: a plain Linear (proj_in).
// C = 320: every wave owns 32 rows as TWO B-operand sets (each weight fragment feeds two MFMAs; 256-row blocks, like stin_block.hip).
template <typename T, int C, bool GEGLU, bool LN, bool GN = false>
__global__ __launch_bounds__(RL_THREADS) void rowlin_kernel(const RowlinParams P) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int KL = C / 64;                // sub-tiles of 64 k, 5 per step
  constexpr int NKS = C / 320;              // steps per piece
  constexpr int RS = C == 320 ? 2 : 1;      // 16-row sets per wave
  constexpr int UPR = GEGLU ? 1 : 2;        // emission units (one 16-byte store per lane each) per row set and piece
  constexpr int NU = RS * UPR;              // ... per piece
  constexpr int S = NU;                     // 16-byte stores per lane and piece
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* par = reinterpret_cast<float*>(smem + 3 * RL_SLOT);      // bias of this block's weight rows

  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int odd = fq & 1, ch8 = (fq >> 1) * 8;
  const int rb = blockIdx.x / P.ny, y = blockIdx.x - rb * P.ny;
  const int m_w0 = rb * (RL_ROWS * RS) + w * 16 * RS;
  const int np = P.np;
  const int g_base = y * np;                // first piece (64 weight rows) of this block's slice
  const int j0 = rb % np;                   // the order of the pieces is free: neighbouring row blocks start at different pieces

  // ---- the wave's rows in B-operand form: lane (fr, fq), set rs holds x[m_w0 + 16 rs + fr][64 t5 + 32 u + 8 fq .. + 7]
  vec8<T> xf[RS][KL][2];
#pragma unroll
  for (int rs = 0; rs < RS; ++rs) {
    const T* xrow = reinterpret_cast<const T*>(P.x) + (size_t)(m_w0 + 16 * rs + fr) * C + 8 * fq;
#pragma unroll
    for (int t5 = 0; t5 < KL; ++t5)
#pragma unroll
      for (int u = 0; u < 2; ++u) xf[rs][t5][u] = *reinterpret_cast<const vec8<T>*>(xrow + 64 * t5 + 32 * u);
  }
  for (int i = t; i < np * 16; i += RL_THREADS)       // bias rows of the slice -> LDS (no register loads inside the loop below)
    *reinterpret_cast<f32x4*>(par + 4 * i) = *reinterpret_cast<const f32x4*>(P.bias + (size_t)g_base * 64 + 4 * i);

  // ---- weight ring: step (piece g, k half kh) = rows 64 g .. + 63, columns 320 kh .. + 319 as five [64 x 64 k] sub-tiles; 5 LDS-DMA per wave
  const __amdgpu_buffer_rsrc_t rsW = uniform_rsrc(P.w, (size_t)P.N * C * 2);
  const unsigned OOB = 0x80000000u;
  const int lrow = w * 8 + (lane >> 3);
  const int lchunk = (lane & 7) ^ ((lrow >> 1) & 7);
  const unsigned vrow = (unsigned)((lrow * C + lchunk * 8) * 2);
  auto issue = [&](int slot, int g, int kh, int i, bool live) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lptr_t)(smem + slot * RL_SLOT + (i * 64 + w * 8) * 128), 16, live ? vrow : OOB,
                                             (g * 64 * C + 320 * kh + i * 64) * 2, 0, RL_DMA_AUX);
  };
  auto piece_of = [&](int j) -> int { const int q = j0 + j; return g_base + (q < np ? q : q - np); };
  {      // the first two steps: slices 0, 1 of the first piece, or (one step per piece) the first two pieces
    const int g0 = piece_of(0);
#pragma unroll
    for (int i = 0; i < 5; ++i) issue(0, g0, 0, i, true);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      if constexpr (NKS == 1) issue(1, np > 1 ? piece_of(1) : g0, 0, i, np > 1); else issue(1, g0, 1, i, true);
    }
  }

  if constexpr (GN) {      // GroupNorm of the raw rows on the way in: the bits lr_groupnorm_apply_n would have written (chain_common.h)
    float* tabA = par + RL_MAX_SLICE;
    float* tabB = tabA + C;
    xa_gn_tables(P.gn_part, P.gn_chunks, (rb * (RL_ROWS * RS)) / P.gn_hw, P.gn_hw, C, P.gn_gamma, P.gn_beta, P.gn_eps, tabA, tabB, tabB + C, t, RL_THREADS);
#pragma unroll
    for (int rs = 0; rs < RS; ++rs)
#pragma unroll
      for (int t5 = 0; t5 < KL; ++t5)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int c0 = 64 * t5 + 32 * u + 8 * fq;
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(tabA + c0), a1 = *reinterpret_cast<const f32x4*>(tabA + c0 + 4);
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(tabB + c0), b1 = *reinterpret_cast<const f32x4*>(tabB + c0 + 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            xf[rs][t5][u][i] = (T)fmaf((float)xf[rs][t5][u][i], a0[i], b0[i]);
            xf[rs][t5][u][4 + i] = (T)fmaf((float)xf[rs][t5][u][4 + i], a1[i], b1[i]);
          }
        }
  }
  // ---- LayerNorm of the rows in registers (two-pass; gamma / beta live in W / bias)
  if constexpr (LN) {
#pragma unroll
    for (int rs = 0; rs < RS; ++rs) {
      __builtin_amdgcn_sched_barrier(0);      // one row set at a time (register pressure)
      float sm = 0.f;
#pragma unroll
      for (int t5 = 0; t5 < KL; ++t5)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int i = 0; i < 8; ++i) sm += (float)xf[rs][t5][u][i];
      const float mean = xa_row4_sum(sm) * (1.0f / C);
#pragma unroll
      for (int t5 = 0; t5 < KL; ++t5)
#pragma unroll
        for (int u = 0; u < 2; ++u) asm volatile("" : "+v"(xf[rs][t5][u]));      // (convert again per pass: keeps the converted floats out of the register file)
      float q2 = 0.f;
#pragma unroll
      for (int t5 = 0; t5 < KL; ++t5)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int i = 0; i < 8; ++i) { const float d = (float)xf[rs][t5][u][i] - mean; q2 = fmaf(d, d, q2); }
      const float rstd = rsqrtf(xa_row4_sum(q2) * (1.0f / C) + P.eps);
      const float nmr = -mean * rstd;
#pragma unroll
      for (int t5 = 0; t5 < KL; ++t5)
#pragma unroll
        for (int u = 0; u < 2; ++u) asm volatile("" : "+v"(xf[rs][t5][u]));
#pragma unroll
      for (int t5 = 0; t5 < KL; ++t5)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int i = 0; i < 8; ++i) xf[rs][t5][u][i] = (T)fmaf((float)xf[rs][t5][u][i], rstd, nmr);
    }
  }

  const int sw = (fr >> 1) & 7;
  auto frag = [&](const char* base, int row, int chunk) -> vec8<T> {
    return *reinterpret_cast<const vec8<T>*>(base + row * 128 + ((chunk ^ sw) << 4));
  };
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  T* outp = reinterpret_cast<T*>(P.out);
  // emit unit uu of a finished piece (local piece index lp, global g): row set uu / UPR; plain -- tiles (2 u, 2 u + 1) = 32 columns with
  // u = uu % 2; GEGLU -- the piece's 32 gated columns: + bias, [gate,] 16 bits, permlane swap, one 16-byte store per lane
  auto emit_unit = [&](const f32x4 (&accs)[RS][4], int lp, int g, int uu) __attribute__((always_inline)) {
    const int rs = uu / UPR, u = uu % UPR;
    const f32x4 (&acc)[4] = accs[rs];
    const float* b = par + lp * 64 + 4 * fq;
    f32x4 a, c;
    int col;
    if constexpr (GEGLU) {
      const f32x4 u0 = acc[0] + *reinterpret_cast<const f32x4*>(b), g0 = acc[1] + *reinterpret_cast<const f32x4*>(b + 16);
      const f32x4 u1 = acc[2] + *reinterpret_cast<const f32x4*>(b + 32), g1 = acc[3] + *reinterpret_cast<const f32x4*>(b + 48);
      const f32x2_t e0 = lr_gelu_erf2((f32x2_t){g0[0], g0[1]}), e1 = lr_gelu_erf2((f32x2_t){g0[2], g0[3]});
      const f32x2_t e2 = lr_gelu_erf2((f32x2_t){g1[0], g1[1]}), e3 = lr_gelu_erf2((f32x2_t){g1[2], g1[3]});
      a = (f32x4){u0[0] * e0[0], u0[1] * e0[1], u0[2] * e1[0], u0[3] * e1[1]};
      c = (f32x4){u1[0] * e2[0], u1[1] * e2[1], u1[2] * e3[0], u1[3] * e3[1]};
      col = g * 32 + odd * 16 + ch8;
    } else {
      a = acc[2 * u] + *reinterpret_cast<const f32x4*>(b + (2 * u) * 16);
      c = acc[2 * u + 1] + *reinterpret_cast<const f32x4*>(b + (2 * u + 1) * 16);
      col = g * 64 + (2 * u + odd) * 16 + ch8;
    }
    xa_swap_rows16(a, c);
    const float v[8] = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
    const uint4 pk = lr_pack8<T>(v);
#ifdef RL_DBG_NOSTORE
    if (pk.x == 0x12345678u && pk.y == 0x9abcdef0u)
#endif
    *reinterpret_cast<uint4*>(outp + (size_t)(m_w0 + 16 * rs + fr) * P.ld_out + col) = pk;
  };

#define RL_FENCE() __builtin_amdgcn_sched_barrier(0)
// timing experiments of developer variants (tools/build_variant.sh x -DRL_DBG_NODMA ...): results are garbage, only the clock matters
#ifdef RL_DBG_NODMA
#define RL_DBG_DMA(X)
#else
#define RL_DBG_DMA(X) X
#endif
#ifdef RL_DBG_NOMFMA
#define RL_DBG_MFMA(X) acc[rs][jd][0] += (float)fa[ks % (RL_DEPTH + 1)][jd][0];
#else
#define RL_DBG_MFMA(X) X
#endif

  // one step in ring slot `slot` (runtime): k slice KH (320 k) of the current piece into acc; fragment reads one k-step ahead; the five
  // LDS-DMA of the step after next after k-steps 0 .. 4 (slice (KH + 2) % NKS of the piece (KH + 2) / NKS further on -> slot2); EMIT(u) at
  // k-steps 5 .. (the previous piece's NU units).  NB = how many of the two previous steps were a piece's FIRST step with stores to issue
  // (those stores are vector memory operations younger than this step's loads, like the next step's five loads)
#define RL_STEP(KH, NB, EMIT)                                                                                                      \
  {                                                                                                                                \
    { const int nb_ = (NB); if (nb_ == 2) xa_wait_vmcnt<5 + 2 * S>(); else if (nb_ == 1) xa_wait_vmcnt<5 + S>(); else xa_wait_vmcnt<5>(); } \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                             \
    __builtin_amdgcn_s_barrier();                                                                                                  \
    const char* Ws = smem + slot * RL_SLOT;                                                                                        \
    const int slot2 = slot == 0 ? 2 : slot - 1;      /* (slot + 2) % 3 */                                                          \
    constexpr int po_ = ((KH) + 2) / NKS, kh2_ = ((KH) + 2) % NKS;                                                                 \
    const int g2_ = po_ == 0 ? g : po_ == 1 ? g_next : g_next2;                                                                    \
    const bool live2_ = po_ == 0 ? true : po_ == 1 ? more : more2;                                                                 \
    vec8<T> fa[RL_DEPTH + 1][4];                                                                                                   \
    auto rd = [&](int ks, vec8<T> (&f)[4]) __attribute__((always_inline)) {                                                        \
      _Pragma("unroll") for (int jd = 0; jd < 4; ++jd) f[jd] = frag(Ws + (ks >> 1) * 64 * 128, jd * 16 + fr, 4 * (ks & 1) + fq);   \
    };                                                                                                                             \
    _Pragma("unroll") for (int d_ = 0; d_ < RL_DEPTH; ++d_) rd(d_, fa[d_]);                                                        \
    _Pragma("unroll") for (int ks = 0; ks < 10; ++ks) {                                                                            \
      if (ks + RL_DEPTH < 10) rd(ks + RL_DEPTH, fa[(ks + RL_DEPTH) % (RL_DEPTH + 1)]);                                             \
      RL_FENCE();                                                                                                                  \
      _Pragma("unroll") for (int jd = 0; jd < 4; ++jd)                                                                             \
        _Pragma("unroll") for (int rs = 0; rs < RS; ++rs)                                                                          \
          RL_DBG_MFMA(acc[rs][jd] = lr_mfma16(fa[ks % (RL_DEPTH + 1)][jd], xf[rs][5 * (KH) + (ks >> 1)][ks & 1], ((KH) == 0 && ks == 0) ? z4 : acc[rs][jd]);) \
      RL_DBG_DMA(if (ks < 5) issue(slot2, g2_, kh2_, ks, live2_);)                                                                 \
      { constexpr int st_ = NU == 4 ? 1 : 2;                                                                                       \
        if (ks >= 5 && (ks - 5) % st_ == 0 && (ks - 5) / st_ < NU) { const int u = (ks - 5) / st_; EMIT; } }                       \
      RL_FENCE();                                                                                                                  \
    }                                                                                                                              \
    slot = slot == 2 ? 0 : slot + 1;                                                                                               \
  }

  f32x4 acc[RS][4], prev[RS][4];
  int slot = 0, g_prev = 0;
#pragma unroll 1
  for (int j = 0; j < np; ++j) {
    const int g = piece_of(j);
    const bool more = j + 1 < np, more2 = j + 2 < np;
    const int g_next = more ? piece_of(j + 1) : g, g_next2 = more2 ? piece_of(j + 2) : g;
    // the previous piece's stores are issued in a piece's first step; see RL_STEP for NB
    if constexpr (NKS == 1) {
      RL_STEP(0, (j >= 3) + (j >= 2), { if (j > 0) emit_unit(prev, g_prev - g_base, g_prev, u); });
    } else if constexpr (NKS == 2) {
      RL_STEP(0, j >= 2, { if (j > 0) emit_unit(prev, g_prev - g_base, g_prev, u); });
      RL_STEP(1, j >= 1, { });
    } else {
      RL_STEP(0, 0, { if (j > 0) emit_unit(prev, g_prev - g_base, g_prev, u); });
      RL_STEP(1, j >= 1, { });
      RL_STEP(2, j >= 1, { });
      RL_STEP(3, 0, { });
    }
#pragma unroll
    for (int rs = 0; rs < RS; ++rs)
#pragma unroll
      for (int jd = 0; jd < 4; ++jd) prev[rs][jd] = acc[rs][jd];
    g_prev = g;
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) emit_unit(prev, g_prev - g_base, g_prev, u);
  xa_wait_vmcnt<0>();                      // (the dead prefetches of the last pieces)
#undef RL_STEP
#undef RL_FENCE
#endif
}

// ---- the same at C = 640 as TWO independent 4-wave blocks per CU -------------------------------------------------------------------------
// The 8-wave block above runs its waves in lockstep: every wave reads the whole 40 KB step from the LDS for 16 rows (one LDS read per MFMA:
// the LDS pipe is as busy as the matrix pipe) and all of them emit at the same k-steps, so neither the reads nor the gate's VALU hide
// behind another wave's MFMAs (profiles/r06_stin_trace.txt).  Here a block is 4 waves x 32 rows (two B-operand sets per wave: every weight
// fragment feeds two MFMAs, half the LDS reads per MFMA), its step 32 weight rows x one k half = 20 KB, its ring 60 KB, 256 registers per
// lane -- two blocks fit a CU and nothing synchronises them: one block's barrier, emission and DMA issue run under the other's MFMAs.
// MEASURED AND LOST (developer builds only, knob LR_ROWLIN_W4; profiles/r06_rowlin_w4.txt, same box, M = 16384, cold): q|k|v 58.2 vs 52.8 us,
// GEGLU 127.5 vs 122.0 us, proj_in 36.1 vs 25.8 us -- halving the LDS reads per MFMA and decoupling the waves buys nothing, so the 8-wave
// block's ~2x distance from its MFMA time is not lockstep LDS reading; the extra row loads (ny = 4) and the 8-byte GEGLU stores cost more.
#ifdef LR_DEV_VARIANTS
#define R4_THREADS 256
#define R4_SLOT 20480
#define R4_ROWS 128
template <typename T, bool GEGLU, bool LN, bool GN>
__global__ __launch_bounds__(R4_THREADS, 2) void rowlin4_kernel(const RowlinParams P) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int C = 640, KL = 10, RS = 2;
  constexpr int S = RS;                     // stores per lane and piece (plain: 16 bytes = 32 columns of a row set; GEGLU: 8 bytes = 16 gated columns)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* par = reinterpret_cast<float*>(smem + 3 * R4_SLOT);

  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int odd = fq & 1, ch8 = (fq >> 1) * 8;
  const int rb = blockIdx.x / P.ny, y = blockIdx.x - rb * P.ny;
  const int m_w0 = rb * R4_ROWS + w * 32;
  const int np = P.np;                      // 32-row pieces of this block's slice
  const int g_base = y * np;
  const int j0 = rb % np;

  vec8<T> xf[RS][KL][2];
#pragma unroll
  for (int rs = 0; rs < RS; ++rs) {
    const T* xrow = reinterpret_cast<const T*>(P.x) + (size_t)(m_w0 + 16 * rs + fr) * C + 8 * fq;
#pragma unroll
    for (int t5 = 0; t5 < KL; ++t5)
#pragma unroll
      for (int u = 0; u < 2; ++u) xf[rs][t5][u] = *reinterpret_cast<const vec8<T>*>(xrow + 64 * t5 + 32 * u);
  }
  for (int i = t; i < np * 8; i += R4_THREADS)
    *reinterpret_cast<f32x4*>(par + 4 * i) = *reinterpret_cast<const f32x4*>(P.bias + (size_t)g_base * 32 + 4 * i);

  // step (piece g, k half kh) = weight rows 32 g .. + 31, columns 320 kh .. + 319 as five [32 x 64 k] sub-tiles: one LDS-DMA per wave each
  const __amdgpu_buffer_rsrc_t rsW = uniform_rsrc(P.w, (size_t)P.N * C * 2);
  const unsigned OOB = 0x80000000u;
  const int lrow = w * 8 + (lane >> 3);
  const int lchunk = (lane & 7) ^ ((lrow >> 1) & 7);
  const unsigned vrow = (unsigned)((lrow * C + lchunk * 8) * 2);
  auto issue = [&](int slot, int g, int kh, int i, bool live) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lptr_t)(smem + slot * R4_SLOT + (i * 32 + w * 8) * 128), 16, live ? vrow : OOB,
                                             (g * 32 * C + 320 * kh + i * 64) * 2, 0, RL_DMA_AUX);
  };
  auto piece_of = [&](int j) -> int { const int q = j0 + j; return g_base + (q < np ? q : q - np); };
  {
    const int g0 = piece_of(0);
#pragma unroll
    for (int i = 0; i < 5; ++i) issue(0, g0, 0, i, true);
#pragma unroll
    for (int i = 0; i < 5; ++i) issue(1, g0, 1, i, true);
  }

  if constexpr (GN) {
    float* tabA = par + RL_MAX_SLICE;
    float* tabB = tabA + C;
    xa_gn_tables(P.gn_part, P.gn_chunks, (rb * R4_ROWS) / P.gn_hw, P.gn_hw, C, P.gn_gamma, P.gn_beta, P.gn_eps, tabA, tabB, tabB + C, t, R4_THREADS);
#pragma unroll
    for (int rs = 0; rs < RS; ++rs)
#pragma unroll
      for (int t5 = 0; t5 < KL; ++t5)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int c0 = 64 * t5 + 32 * u + 8 * fq;
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(tabA + c0), a1 = *reinterpret_cast<const f32x4*>(tabA + c0 + 4);
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(tabB + c0), b1 = *reinterpret_cast<const f32x4*>(tabB + c0 + 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            xf[rs][t5][u][i] = (T)fmaf((float)xf[rs][t5][u][i], a0[i], b0[i]);
            xf[rs][t5][u][4 + i] = (T)fmaf((float)xf[rs][t5][u][4 + i], a1[i], b1[i]);
          }
        }
  }
  if constexpr (LN) {
#pragma unroll
    for (int rs = 0; rs < RS; ++rs) {
      __builtin_amdgcn_sched_barrier(0);
      float sm = 0.f;
#pragma unroll
      for (int t5 = 0; t5 < KL; ++t5)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int i = 0; i < 8; ++i) sm += (float)xf[rs][t5][u][i];
      const float mean = xa_row4_sum(sm) * (1.0f / C);
#pragma unroll
      for (int t5 = 0; t5 < KL; ++t5)
#pragma unroll
        for (int u = 0; u < 2; ++u) asm volatile("" : "+v"(xf[rs][t5][u]));
      float q2 = 0.f;
#pragma unroll
      for (int t5 = 0; t5 < KL; ++t5)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int i = 0; i < 8; ++i) { const float d = (float)xf[rs][t5][u][i] - mean; q2 = fmaf(d, d, q2); }
      const float rstd = rsqrtf(xa_row4_sum(q2) * (1.0f / C) + P.eps);
      const float nmr = -mean * rstd;
#pragma unroll
      for (int t5 = 0; t5 < KL; ++t5)
#pragma unroll
        for (int u = 0; u < 2; ++u) asm volatile("" : "+v"(xf[rs][t5][u]));
#pragma unroll
      for (int t5 = 0; t5 < KL; ++t5)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int i = 0; i < 8; ++i) xf[rs][t5][u][i] = (T)fmaf((float)xf[rs][t5][u][i], rstd, nmr);
    }
  }

  const int sw = (fr >> 1) & 7;
  auto frag = [&](const char* base, int row, int chunk) -> vec8<T> {
    return *reinterpret_cast<const vec8<T>*>(base + row * 128 + ((chunk ^ sw) << 4));
  };
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  T* outp = reinterpret_cast<T*>(P.out);
  // emit row set rs of a finished piece (local index lp, global g): plain -- its two tiles = 32 columns, one 16-byte store per lane;
  // GEGLU -- tiles (u, g) -> 16 gated columns, lane (fr, fq) holds columns 4 fq .. + 3 of row fr: one 8-byte store
  auto emit_unit = [&](const f32x4 (&accs)[RS][2], int lp, int g, int rs) __attribute__((always_inline)) {
    const f32x4 (&acc)[2] = accs[rs];
    const float* b = par + lp * 32 + 4 * fq;
    T* orow = outp + (size_t)(m_w0 + 16 * rs + fr) * P.ld_out;
    if constexpr (GEGLU) {
      const f32x4 u0 = acc[0] + *reinterpret_cast<const f32x4*>(b), g0 = acc[1] + *reinterpret_cast<const f32x4*>(b + 16);
      const f32x2_t e0 = lr_gelu_erf2((f32x2_t){g0[0], g0[1]}), e1 = lr_gelu_erf2((f32x2_t){g0[2], g0[3]});
      const vec4<T> pk = {(T)(u0[0] * e0[0]), (T)(u0[1] * e0[1]), (T)(u0[2] * e1[0]), (T)(u0[3] * e1[1])};
      *reinterpret_cast<vec4<T>*>(orow + g * 16 + 4 * fq) = pk;
    } else {
      f32x4 a = acc[0] + *reinterpret_cast<const f32x4*>(b);
      f32x4 c = acc[1] + *reinterpret_cast<const f32x4*>(b + 16);
      xa_swap_rows16(a, c);
      const float v[8] = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
      *reinterpret_cast<uint4*>(orow + g * 32 + odd * 16 + ch8) = lr_pack8<T>(v);
    }
  };

#define R4_FENCE() __builtin_amdgcn_sched_barrier(0)
  // one step (k half KH of the current piece) in ring slot `slot`; the five LDS-DMA of the step after next after k-steps 0 .. 4; the previous
  // piece's two row sets are emitted at k-steps 5 and 7 of a piece's first step.  NB as in RL_STEP.
#define R4_STEP(KH, NB, EMIT)                                                                                                      \
  {                                                                                                                                \
    { const int nb_ = (NB); if (nb_ == 1) xa_wait_vmcnt<5 + S>(); else xa_wait_vmcnt<5>(); }                                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                             \
    __builtin_amdgcn_s_barrier();                                                                                                  \
    const char* Ws = smem + slot * R4_SLOT;                                                                                        \
    const int slot2 = slot == 0 ? 2 : slot - 1;                                                                                    \
    const int g2_ = (KH) == 0 ? g_next : g_next;                                                                                   \
    const bool live2_ = more;                                                                                                      \
    vec8<T> fa[2][2];                                                                                                              \
    auto rd = [&](int ks, vec8<T> (&f)[2]) __attribute__((always_inline)) {                                                        \
      _Pragma("unroll") for (int jd = 0; jd < 2; ++jd) f[jd] = frag(Ws + (ks >> 1) * 32 * 128, jd * 16 + fr, 4 * (ks & 1) + fq);   \
    };                                                                                                                             \
    rd(0, fa[0]);                                                                                                                  \
    _Pragma("unroll") for (int ks = 0; ks < 10; ++ks) {                                                                            \
      if (ks + 1 < 10) rd(ks + 1, fa[(ks + 1) & 1]);                                                                               \
      R4_FENCE();                                                                                                                  \
      _Pragma("unroll") for (int jd = 0; jd < 2; ++jd)                                                                             \
        _Pragma("unroll") for (int rs = 0; rs < RS; ++rs)                                                                          \
          acc[rs][jd] = lr_mfma16(fa[ks & 1][jd], xf[rs][5 * (KH) + (ks >> 1)][ks & 1], ((KH) == 0 && ks == 0) ? z4 : acc[rs][jd]); \
      if (ks < 5) issue(slot2, g2_, (KH), ks, live2_);                                                                             \
      if (ks == 5 || ks == 7) { const int u = (ks - 5) >> 1; EMIT; }                                                               \
      R4_FENCE();                                                                                                                  \
    }                                                                                                                              \
    slot = slot == 2 ? 0 : slot + 1;                                                                                               \
  }

  f32x4 acc[RS][2], prev[RS][2];
  int slot = 0, g_prev = 0;
#pragma unroll 1
  for (int j = 0; j < np; ++j) {
    const int g = piece_of(j);
    const bool more = j + 1 < np;
    const int g_next = more ? piece_of(j + 1) : g;
    // step (j, 0) prefetches (j + 1, 0), step (j, 1) prefetches (j + 1, 1): two steps ahead, the same k half of the next piece
    R4_STEP(0, j >= 2, { if (j > 0) emit_unit(prev, g_prev - g_base, g_prev, u); });
    R4_STEP(1, j >= 1, { });
#pragma unroll
    for (int rs = 0; rs < RS; ++rs)
#pragma unroll
      for (int jd = 0; jd < 2; ++jd) prev[rs][jd] = acc[rs][jd];
    g_prev = g;
  }
#pragma unroll
  for (int u = 0; u < RS; ++u) emit_unit(prev, g_prev - g_base, g_prev, u);
  xa_wait_vmcnt<0>();
#undef R4_STEP
#undef R4_FENCE
#endif
}

template <typename T, bool GEGLU, bool LN, bool GN>
static int rowlin4_launch(const RowlinParams& P, hipStream_t st) {
  const size_t smem = 3 * R4_SLOT + (size_t)(RL_MAX_SLICE + (GN ? 2 * 640 + 64 : 0)) * sizeof(float);      // <= 77 KB: two blocks per CU
  static unsigned long long attr_done = 0;
  if (lr_attr_needed(&attr_done))
    hipFuncSetAttribute(reinterpret_cast<const void*>(rowlin4_kernel<T, GEGLU, LN, GN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL((rowlin4_kernel<T, GEGLU, LN, GN>), dim3((P.M / R4_ROWS) * P.ny), dim3(R4_THREADS), smem, st, P);
  return lr_launch_status();
}

// column slices of the 4-wave form: ~two blocks per CU, dividing the number of 32-row pieces, at least 3 pieces per block
static int rowlin4_ny(int M, int N) {
  const int rbs = M / R4_ROWS, pieces = N / 32;
  int want = 512 / (rbs > 0 ? rbs : 1);
  if (want < 1) want = 1;
  int ny = 0;
  for (int d = 1; d <= pieces && d <= want; ++d)
    if (pieces % d == 0 && pieces / d >= 3 && pieces / d * 32 <= RL_MAX_SLICE) ny = d;
  return ny;      // 0: no slicing fits
}
#endif      // LR_DEV_VARIANTS

// column slices per row block: as many as keep ~one block per CU, dividing the number of 64-row pieces
static int rowlin_ny(int M, int N, int rows) {
  const int rbs = M / rows, pieces = N / 64;
  int want = 256 / (rbs > 0 ? rbs : 1);
  if (want < 1) want = 1;
  int ny = 1;
  for (int d = 1; d <= pieces && d <= want; ++d)
    if (pieces % d == 0 && pieces / d * 64 <= RL_MAX_SLICE) ny = d;
  while (pieces / ny * 64 > RL_MAX_SLICE && ny < pieces) {      // the bias slice must fit behind the ring
    ++ny;
    while (ny < pieces && pieces % ny) ++ny;
  }
  return ny;
}

template <typename T, int C, bool GEGLU, bool LN, bool GN = false>
static int rowlin_launch(const RowlinParams& P, hipStream_t st) {
  const size_t smem = 3 * RL_SLOT + (size_t)(RL_MAX_SLICE + (GN ? 2 * C + 64 : 0)) * sizeof(float);
  static unsigned long long attr_done = 0;
  if (lr_attr_needed(&attr_done))
    hipFuncSetAttribute(reinterpret_cast<const void*>(rowlin_kernel<T, C, GEGLU, LN, GN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL((rowlin_kernel<T, C, GEGLU, LN, GN>), dim3((P.M / (C == 320 ? 2 * RL_ROWS : RL_ROWS)) * P.ny), dim3(RL_THREADS), smem, st, P);
  return lr_launch_status();
}

template <typename T>
static int rowlin_t(const lr_rowlin_args* a, lr_stream_t s) {
  if (!a || !a->x || !a->w || !a->bias || !a->out) return LR_E_ARG;
  if (a->M <= 0 || a->N <= 0 || (a->geglu != 0 && a->geglu != 1) || (a->ln != 0 && a->ln != 1)) return LR_E_ARG;
  // C = 1280 (level 2: 4096 rows) is compiled in developer builds only -- measured and lost to the tiled GEMM there: 31.9 vs 24.5 us
  // (N = 1280), 58.2 vs 54.8 (q|k|v), 128 vs 113 (GEGLU): 32 row blocks leave 5 .. 8 column slices of a few pieces each, and the slices
  // re-load the rows (profiles/r06_rowlin_microbench.txt)
  const int rows = a->C == 320 ? 2 * RL_ROWS : RL_ROWS;
  // C = 320 (level 0: the gated projection on 256-row blocks, for a feed-forward split into projection + one composed GEMM) is a developer
  // build instance too: measured 129.5 + 99.1 us against 220-225 us of the fused lr_ffn_block_f16, UNet step 17.92 vs 17.87 ms same box --
  // the gate's ~300 VALU instructions per piece and wave cost as much as the piece's MFMAs and the waves run them in lockstep
#ifdef LR_DEV_VARIANTS
  if ((a->C != 320 && a->C != 640 && a->C != 1280) || a->M % rows || a->N % 64) return LR_E_UNSUPPORTED;
#else
  if (a->C != 640 || a->M % rows || a->N % 64) return LR_E_UNSUPPORTED;
#endif
  if (a->geglu && !a->ln) return LR_E_UNSUPPORTED;      // (the gated projection always follows norm3)
  const int n_out = a->geglu ? a->N / 2 : a->N;
  if (a->ld_out < n_out || a->ld_out % 8) return LR_E_ALIGN;
  if (((uintptr_t)a->x | (uintptr_t)a->w | (uintptr_t)a->bias | (uintptr_t)a->out) & 15) return LR_E_ALIGN;
  if ((int64_t)a->N * a->C * 2 >= ((int64_t)1 << 31)) return LR_E_UNSUPPORTED;
  RowlinParams P;
  P.x = a->x; P.w = a->w; P.bias = a->bias; P.out = a->out;
  P.M = a->M; P.N = a->N; P.ld_out = a->ld_out; P.eps = a->ln_eps;
  P.ny = rowlin_ny(a->M, a->N, rows);
  P.np = a->N / 64 / P.ny;
  if (P.np * 64 > RL_MAX_SLICE) return LR_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)s;
  const bool gn = a->gn_part != nullptr;
  if (gn) {      // GroupNorm of the input rows: the plain Linear at C = 640 only (SpatialTransformer.norm + proj_in)
    if (a->C != 640 || a->ln || a->geglu) return LR_E_UNSUPPORTED;
    if (!a->gn_gamma || !a->gn_beta || a->gn_chunks <= 0 || a->gn_hw <= 0) return LR_E_ARG;
    if (a->gn_hw % rows || a->M % a->gn_hw) return LR_E_UNSUPPORTED;      // a block stays inside one sample
    if (((uintptr_t)a->gn_part & 7) || (((uintptr_t)a->gn_gamma | (uintptr_t)a->gn_beta) & 3)) return LR_E_ALIGN;
  }
  P.gn_part = a->gn_part; P.gn_gamma = a->gn_gamma; P.gn_beta = a->gn_beta; P.gn_chunks = a->gn_chunks; P.gn_hw = a->gn_hw; P.gn_eps = a->gn_eps;
#ifdef LR_DEV_VARIANTS
  // C = 640: the 4-wave form (two independent blocks per CU) where its slicing fits -- measured, lost
  if (a->C == 640 && LR_DEV("LR_ROWLIN_W4", RL_W4_DEFAULT)) {
    const int ny4 = rowlin4_ny(a->M, a->N);
    if (ny4 > 0) {
      P.ny = ny4;
      P.np = a->N / 32 / ny4;
      if (gn) return rowlin4_launch<T, false, false, true>(P, st);
      if (a->geglu) return rowlin4_launch<T, true, true, false>(P, st);
      return a->ln ? rowlin4_launch<T, false, true, false>(P, st) : rowlin4_launch<T, false, false, false>(P, st);
    }
  }
#endif
  if (gn) return rowlin_launch<T, 640, false, false, true>(P, st);
#ifdef LR_DEV_VARIANTS
  if (a->C == 320) {      // level 0: the gated projection only (q|k|v and proj_in are stin_block's)
    if (a->geglu) return rowlin_launch<T, 320, true, true>(P, st);
    return LR_E_UNSUPPORTED;
  }
#endif
  if (a->C == 640) {
    if (a->geglu) return rowlin_launch<T, 640, true, true>(P, st);
    return a->ln ? rowlin_launch<T, 640, false, true>(P, st) : rowlin_launch<T, 640, false, false>(P, st);
  }
#ifdef LR_DEV_VARIANTS
  if (a->geglu) return rowlin_launch<T, 1280, true, true>(P, st);
  return a->ln ? rowlin_launch<T, 1280, false, true>(P, st) : rowlin_launch<T, 1280, false, false>(P, st);
#else
  return LR_E_UNSUPPORTED;
#endif
}

extern "C" int lr_rowlin_f16(const lr_rowlin_args* a, lr_stream_t s) { return rowlin_t<f16>(a, s); }
extern "C" int lr_rowlin_bf16(const lr_rowlin_args* a, lr_stream_t s) { return rowlin_t<bf16>(a, s); }
