// Entry of a SpatialTransformer's (only) BasicTransformerBlock at C = 320 (level 0 of the SD2 UNet) as ONE kernel
// (reference ldm/modules/attention.py:405-408 `x = self.proj_in(x)` with use_linear, then 279-280 / 165-172: norm1 + attn1.to_q / to_k / to_v):
//
//     x1  = h Wp^T + bp                                  (proj_in; h = the GroupNorm-ed tokens)
//     qkv = LayerNorm(x1) [Wq; Wk; Wv]^T                 (gamma / beta folded into the weights / a bias row: packing.fold_layernorm)
//
// instead of the K = 320 GEMM (45 us at M = 65536: 5 K-steps under a 164 KB-per-block epilogue) and the LayerNorm-folded 960-column
// GEMM (84 us: 768 tiles in three rounds, each a 5-step loop between a fill and a 40-tile epilogue) -- 130 us for 54 GFLOP and
// 210 MB, against 27 us of HBM time for the bytes and 22 us of matrix time for the FLOPs.
//
// Register chain like xattn_block.hip / ffn_block.hip (swapped 16x16x32 MFMAs: weights = A operand from LDS, rows = B operand from
// registers), cut for a row-resident GEMM with a WIDE output: block = 8 waves = 256 rows, every wave owns 32 rows as TWO B-operand
// sets, so each weight fragment read from LDS feeds two MFMAs and 65536 rows are exactly 256 blocks: the 0.82 MB of weights stream
// once per CU from its XCD's L2.  One ring piece = 64 output columns x all 320 k (40 KB, five [64 x 64] sub-tiles, 3 slots, loads
// two pieces ahead, counted vmcnt + one barrier per piece); per piece a wave runs 80 MFMAs into 8 accumulator tiles and emits them at
// once -- + bias, 16 bits, permlane swap to 16 bytes per lane, four stores -- so outputs leave WHILE the next pieces multiply:
//   pieces 0 .. 4   x1 columns 64 p ..; then the wave reads its own 32 rows of x1 back (L2-hot, natural k order -- holding them in
//                   registers next to the first stage's operands overflows the 256 registers a wave has at two waves per SIMD: 209 spills)
//                   and normalises them in registers (two-pass LayerNorm);
//   pieces 5 .. 19  qkv columns 64 n ..
// The stores share the vmcnt queue with the LDS-DMA loads (vector memory operations of a wave complete in issue order): the counted
// waits below count both (4 stores per piece and wave).
#include "chain_common.h"

#define SI_C 320
#define SI_ROWS 256
#define SI_THREADS 512
#define SI_SLOT 40960
#define SI_OUT_AUX 0          // cache policy of the output stores (0 = default)

struct StinParams {
  const void* x; const void* wp; const float* bp; const void* wqkv; const float* bqkv;
  void* x1; void* qkv;
  int M, NQ, ld_qkv;          // NQ = width of the second stage (960), ld_qkv = row stride of qkv in elements
  float eps;
};

template <typename T>
__global__ __launch_bounds__(SI_THREADS) void stin_block_kernel(const StinParams P) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int C = SI_C, KL = C / 64;      // 5 sub-tiles of 64 k
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* par = reinterpret_cast<float*>(smem + 3 * SI_SLOT);      // [C] bp | [NQ] bqkv

  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int odd = fq & 1, ch8 = (fq >> 1) * 8;
  const int m_w0 = blockIdx.x * SI_ROWS + w * 32;

  // ---- the wave's 32 rows as two B-operand sets: lane (fr, fq), set rs holds x[m_w0 + 16 rs + fr][64 t5 + 32 u + 8 fq .. + 7]
  vec8<T> xf[2][KL][2];
#pragma unroll
  for (int rs = 0; rs < 2; ++rs) {
    const T* xrow = reinterpret_cast<const T*>(P.x) + (size_t)(m_w0 + 16 * rs + fr) * C + 8 * fq;
#pragma unroll
    for (int t5 = 0; t5 < KL; ++t5)
#pragma unroll
      for (int u = 0; u < 2; ++u) xf[rs][t5][u] = *reinterpret_cast<const vec8<T>*>(xrow + 64 * t5 + 32 * u);
  }
  for (int i = t; i < (C + P.NQ) / 4; i += SI_THREADS) {      // bias rows -> LDS (no register loads inside the piece loop)
    const float* src = i < C / 4 ? P.bp + 4 * i : P.bqkv + 4 * (i - C / 4);
    *reinterpret_cast<f32x4*>(par + 4 * i) = *reinterpret_cast<const f32x4*>(src);
  }

  // ---- weight ring: piece s < 5 = rows 64 s .. of Wp, piece s >= 5 = rows 64 (s - 5) .. of Wqkv; 5 LDS-DMA instructions of 1 KiB per wave
  const __amdgpu_buffer_rsrc_t rsP = uniform_rsrc(P.wp, (size_t)C * C * 2);
  const __amdgpu_buffer_rsrc_t rsQ = uniform_rsrc(P.wqkv, (size_t)P.NQ * C * 2);
  const int lrow = w * 8 + (lane >> 3);                      // row of a 64-row sub-tile this lane fills
  const int lchunk = (lane & 7) ^ ((lrow >> 1) & 7);         // source-side swizzle
  const unsigned vrow = (unsigned)((lrow * C + lchunk * 8) * 2);
  const unsigned OOB = 0x80000000u;
  // sub-tile i of rows 64 n .. of Wp -> slot
  auto issue_p = [&](int slot, int n, int i) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP, (lptr_t)(smem + slot * SI_SLOT + (i * 64 + w * 8) * 128), 16, vrow, n * 64 * C * 2 + i * 128, 0, 0);
  };
  // sub-tile i of rows 64 n .. of Wqkv -> slot (live = false: out-of-range offset, zeros nobody reads -- keeps the counted waits uniform)
  auto issue_q = [&](int slot, int n, int i, bool live = true) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lptr_t)(smem + slot * SI_SLOT + (i * 64 + w * 8) * 128), 16, live ? vrow : OOB,
                                             n * 64 * C * 2 + i * 128, 0, 0);
  };
#pragma unroll
  for (int i = 0; i < KL; ++i) issue_p(0, 0, i);
#pragma unroll
  for (int i = 0; i < KL; ++i) issue_p(1, 1, i);

  const int sw = (fr >> 1) & 7;
  auto frag = [&](const char* base, int row, int chunk) -> vec8<T> {
    return *reinterpret_cast<const vec8<T>*>(base + row * 128 + ((chunk ^ sw) << 4));
  };
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#define SI_FENCE() __builtin_amdgcn_sched_barrier(0)

  // one piece in slot SLOT: 10 k-steps x (4 fragment reads, 8 MFMAs), fragment reads one k-step ahead, one LDS-DMA instruction of the
  // piece after next (ISSUE, uses ks) after each of the first five k-steps; XB = the B operands (xf | xn); NWAIT = vector memory
  // operations younger than this piece's loads (stores of the two previous pieces + the next piece's loads); then the epilogue EPI(acc)
#define SI_PIECE(SLOT, XB, NWAIT, ISSUE, EPI)                                                                                   \
  {                                                                                                                             \
    xa_wait_vmcnt<NWAIT>();                                                                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                          \
    __builtin_amdgcn_s_barrier();                                                                                               \
    const char* Ws = smem + (SLOT) * SI_SLOT;                                                                                   \
    f32x4 acc[2][4] = {{z4, z4, z4, z4}, {z4, z4, z4, z4}};                                                                     \
    vec8<T> fa[2][4];                                                                                                           \
    auto rd = [&](int ks, vec8<T> (&f)[4]) __attribute__((always_inline)) {                                                     \
      _Pragma("unroll") for (int jd = 0; jd < 4; ++jd) f[jd] = frag(Ws + (ks >> 1) * 64 * 128, jd * 16 + fr, 4 * (ks & 1) + fq); \
    };                                                                                                                          \
    rd(0, fa[0]);                                                                                                               \
    _Pragma("unroll") for (int ks = 0; ks < 10; ++ks) {                                                                         \
      if (ks + 1 < 10) rd(ks + 1, fa[(ks + 1) & 1]);                                                                            \
      SI_FENCE();                                                                                                               \
      _Pragma("unroll") for (int jd = 0; jd < 4; ++jd) {                                                                        \
        acc[0][jd] = lr_mfma16(fa[ks & 1][jd], XB[0][ks >> 1][ks & 1], acc[0][jd]);                                             \
        acc[1][jd] = lr_mfma16(fa[ks & 1][jd], XB[1][ks >> 1][ks & 1], acc[1][jd]);                                             \
      }                                                                                                                         \
      if (ks < KL) { ISSUE; }                                                                                                   \
      SI_FENCE();                                                                                                               \
    }                                                                                                                           \
    EPI;                                                                                                                        \
  }

  // emit 8 accumulator tiles (2 row sets x 64 columns c0 ..) of one piece: + bias (LDS row `bias`), 16 bits, 16-byte stores to dst (row
  // stride ld elements)
  auto emit = [&](f32x4 (&acc)[2][4], const float* bias, T* dst, int ld, int c0) __attribute__((always_inline)) {
#pragma unroll
    for (int rs = 0; rs < 2; ++rs)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        f32x4 a = acc[rs][2 * q] + *reinterpret_cast<const f32x4*>(bias + c0 + (2 * q) * 16 + 4 * fq);
        f32x4 b = acc[rs][2 * q + 1] + *reinterpret_cast<const f32x4*>(bias + c0 + (2 * q + 1) * 16 + 4 * fq);
        xa_swap_rows16(a, b);
        const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        const uint4 pk = lr_pack8<T>(v);
        *reinterpret_cast<uint4*>(dst + (size_t)(m_w0 + 16 * rs + fr) * ld + c0 + (2 * q + odd) * 16 + ch8) = pk;
      }
  };

  // ================= stage 1: x1 = h Wp^T + bp ===========================================================================
  T* x1 = reinterpret_cast<T*>(P.x1);
  SI_PIECE(0, xf, 5, issue_p(2, 2, ks), emit(acc, par, x1, C, 0));
  SI_PIECE(1, xf, 9, issue_p(0, 3, ks), emit(acc, par, x1, C, 64));
  SI_PIECE(2, xf, 13, issue_p(1, 4, ks), emit(acc, par, x1, C, 128));
  SI_PIECE(0, xf, 13, issue_q(2, 0, ks), emit(acc, par, x1, C, 192));
  SI_PIECE(1, xf, 13, issue_q(0, 1, ks), emit(acc, par, x1, C, 256));

  // ---- the wave's x1 rows come back as the second stage's B operands (natural k order): they left rounded to 16 bits, which is what
  //      the unfused path normalises; the lines were written by this wave (vmcnt(0): its stores have reached the L2, and every older
  //      operation with them) and never read before, so the loads cannot find a stale copy in this CU's L1.  Two-pass LayerNorm in registers.
  xa_wait_vmcnt<0>();
#pragma unroll
  for (int rs = 0; rs < 2; ++rs) {
    const T* xrow = x1 + (size_t)(m_w0 + 16 * rs + fr) * C + 8 * fq;
#pragma unroll
    for (int t5 = 0; t5 < KL; ++t5)
#pragma unroll
      for (int u = 0; u < 2; ++u) xf[rs][t5][u] = *reinterpret_cast<const vec8<T>*>(xrow + 64 * t5 + 32 * u);
  }
#pragma unroll
  for (int rs = 0; rs < 2; ++rs) {
    __builtin_amdgcn_sched_barrier(0);      // one row set at a time (register pressure)
    float sm = 0.f;
#pragma unroll
    for (int t5 = 0; t5 < KL; ++t5)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) sm += (float)xf[rs][t5][u][i];
    const float mean = xa_row4_sum(sm) * (1.0f / C);
    // (each pass converts the 16-bit values again: the opaque touch keeps hipcc from holding 160 converted floats across the passes,
    // which spilled 15 registers to scratch)
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

  // ================= stage 2: qkv = xn Wqkv^T + bqkv, 64 columns per piece ================================================
  T* qkv = reinterpret_cast<T*>(P.qkv);
  const int nq = P.NQ / 64;                // pieces of this stage: a multiple of 3, so every piece's slot is a compile-time constant
#pragma unroll 1
  for (int n = 0; n < nq; n += 3) {        // pieces n, n + 1, n + 2 in slots 2, 0, 1 (five pieces of stage 1: 5 % 3 == 2)
    SI_PIECE(2, xf, 13, issue_q(1, n + 2, ks, n + 2 < nq), emit(acc, par + C, qkv, P.ld_qkv, n * 64));
    SI_PIECE(0, xf, 13, issue_q(2, n + 3, ks, n + 3 < nq), emit(acc, par + C, qkv, P.ld_qkv, (n + 1) * 64));
    SI_PIECE(1, xf, 13, issue_q(0, n + 4, ks, n + 4 < nq), emit(acc, par + C, qkv, P.ld_qkv, (n + 2) * 64));
  }
  xa_wait_vmcnt<0>();                      // (the dead prefetches of the last two pieces)
#undef SI_PIECE
#undef SI_FENCE
#endif
}

template <typename T>
static int stin_block_t(const lr_stin_args* a, lr_stream_t s) {
  if (!a || !a->x || !a->wp || !a->bp || !a->wqkv || !a->bqkv || !a->x1 || !a->qkv) return LR_E_ARG;
  if (a->M <= 0 || a->ld_qkv < a->NQ) return LR_E_ARG;
  if (a->C != SI_C || a->NQ <= 0 || a->NQ % 192 || a->NQ > 4096) return LR_E_UNSUPPORTED;      // whole triples of 64-column pieces
  if (a->M % SI_ROWS) return LR_E_UNSUPPORTED;
  if (a->ld_qkv % 8) return LR_E_ALIGN;
  if (((uintptr_t)a->x | (uintptr_t)a->wp | (uintptr_t)a->bp | (uintptr_t)a->wqkv | (uintptr_t)a->bqkv | (uintptr_t)a->x1 | (uintptr_t)a->qkv) & 15)
    return LR_E_ALIGN;
  if ((int64_t)a->NQ * SI_C * 2 >= ((int64_t)1 << 31)) return LR_E_UNSUPPORTED;
  StinParams P;
  P.x = a->x; P.wp = a->wp; P.bp = a->bp; P.wqkv = a->wqkv; P.bqkv = a->bqkv; P.x1 = a->x1; P.qkv = a->qkv;
  P.M = a->M; P.NQ = a->NQ; P.ld_qkv = a->ld_qkv; P.eps = a->ln_eps;
  const size_t smem = 3 * SI_SLOT + (size_t)(SI_C + a->NQ) * sizeof(float);
  static unsigned long long attr_done = 0;
  if (lr_attr_needed(&attr_done))
    hipFuncSetAttribute(reinterpret_cast<const void*>(stin_block_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL((stin_block_kernel<T>), dim3(a->M / SI_ROWS), dim3(SI_THREADS), smem, (hipStream_t)s, P);
  return lr_launch_status();
}

extern "C" int lr_stin_block_f16(const lr_stin_args* a, lr_stream_t s) { return stin_block_t<f16>(a, s); }
extern "C" int lr_stin_block_bf16(const lr_stin_args* a, lr_stream_t s) { return stin_block_t<bf16>(a, s); }
