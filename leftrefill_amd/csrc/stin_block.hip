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
  // GN: x holds the RAW tokens and the SpatialTransformer's GroupNorm(32) (attention.py:399-404) is applied to the rows as they are loaded
  const float* gn_part; const float* gn_gamma; const float* gn_beta; int gn_chunks, gn_hw; float gn_eps;
#ifdef SI_TRACE
  unsigned long long* trace;   // developer build only: shader-clock stamps [block][8 waves][64] (tools/trace_stin.py)
#endif
};

#ifdef SI_TRACE
#define SI_STAMP(k) do { if (P.trace && lane == 0) P.trace[((size_t)blockIdx.x * 8 + w) * 64 + (k)] = __builtin_readcyclecounter(); } while (0)
static unsigned long long* g_si_trace = nullptr;
extern "C" void lr_stin_set_trace(void* p) { g_si_trace = (unsigned long long*)p; }
#else
#define SI_STAMP(k) do { } while (0)
#endif

template <typename T, bool GN>
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
  int si_piece = 0;           // running piece number (trace stamps only)
  SI_STAMP(0);

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
  if constexpr (GN) {
    // the block's 256 rows lie in ONE sample (gn_hw % 256 == 0): its scale / shift tables, then the rows are normalised in place and
    // rounded to 16 bits -- the tensor lr_groupnorm_apply_n would have written, bit for bit, without the round trip through memory
    float* tabA = par + C + P.NQ;
    float* tabB = tabA + C;
    xa_gn_tables(P.gn_part, P.gn_chunks, (blockIdx.x * SI_ROWS) / P.gn_hw, P.gn_hw, C, P.gn_gamma, P.gn_beta, P.gn_eps, tabA, tabB, tabB + C, t, SI_THREADS);
#pragma unroll
    for (int rs = 0; rs < 2; ++rs)
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
  const int nq = P.NQ / 64;                // pieces of the second stage: a multiple of 3, so every piece's slot is a compile-time constant
#ifdef SI_NO_ROTATE
  const int p0 = 0, n0 = 0;
#else
  const int p0 = (blockIdx.x >> 3) % KL, n0 = (blockIdx.x >> 3) % nq;
#endif
#define PP(i) ((p0 + (i)) % KL)
#define QN(i) ((n0 + (i)) % nq)
#pragma unroll
  for (int i = 0; i < KL; ++i) issue_p(0, PP(0), i);
#pragma unroll
  for (int i = 0; i < KL; ++i) issue_p(1, PP(1), i);

  const int sw = (fr >> 1) & 7;
  auto frag = [&](const char* base, int row, int chunk) -> vec8<T> {
    return *reinterpret_cast<const vec8<T>*>(base + row * 128 + ((chunk ^ sw) << 4));
  };
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#define SI_FENCE() __builtin_amdgcn_sched_barrier(0)
// timing experiments of developer variants (tools/build_variant.sh x -DSI_DBG_NODMA ...): results are garbage, only the clock matters
#ifdef SI_DBG_NODMA
#define SI_DBG_DMA(X)
#else
#define SI_DBG_DMA(X) X
#endif
#ifdef SI_DBG_NOMFMA
#define SI_DBG_MFMA(X) acc[0][jd][0] += (float)fa[ks & 1][jd][0];
#else
#define SI_DBG_MFMA(X) X
#endif

  // one piece in slot SLOT: 10 k-steps x (4 fragment reads, 8 MFMAs) into accumulator set CUR, fragment reads one k-step ahead, one LDS-DMA
  // instruction of the piece after next (ISSUE, uses ks) after each of the first five k-steps; XB = the B operands; NWAIT = vector memory
  // operations younger than this piece's loads.  The PREVIOUS piece's accumulators (set CUR ^ 1) are emitted WHILE this one multiplies
  // (PREV(u), u = 0 .. 3, one 16-row x 32-column unit after each of k-steps 5 .. 8): all eight waves of the block run the same piece between
  // two barriers, so an epilogue that followed its own k-loop left the matrix pipes idle for its whole length (measured: 77 us, of which 20
  // matrix, 21 stores, 36 everything else, none of it overlapped).
#define SI_PIECE(SLOT, XB, NWAIT, ISSUE, CUR, PREV)                                                                             \
  {                                                                                                                             \
    SI_STAMP(2 + 3 * si_piece);                                                                                                 \
    xa_wait_vmcnt<NWAIT>();                                                                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                          \
    SI_STAMP(3 + 3 * si_piece);                                                                                                 \
    __builtin_amdgcn_s_barrier();                                                                                               \
    SI_STAMP(4 + 3 * si_piece);                                                                                                 \
    ++si_piece;                                                                                                                 \
    const char* Ws = smem + (SLOT) * SI_SLOT;                                                                                   \
    f32x4 (&acc)[2][4] = accs[CUR];                                                                                             \
    vec8<T> fa[2][4];                                                                                                           \
    auto rd = [&](int ks, vec8<T> (&f)[4]) __attribute__((always_inline)) {                                                     \
      _Pragma("unroll") for (int jd = 0; jd < 4; ++jd) f[jd] = frag(Ws + (ks >> 1) * 64 * 128, jd * 16 + fr, 4 * (ks & 1) + fq); \
    };                                                                                                                          \
    rd(0, fa[0]);                                                                                                               \
    _Pragma("unroll") for (int ks = 0; ks < 10; ++ks) {                                                                         \
      if (ks + 1 < 10) rd(ks + 1, fa[(ks + 1) & 1]);                                                                            \
      SI_FENCE();                                                                                                               \
      _Pragma("unroll") for (int jd = 0; jd < 4; ++jd) {                                                                        \
        SI_DBG_MFMA(acc[0][jd] = lr_mfma16(fa[ks & 1][jd], XB[0][ks >> 1][ks & 1], ks ? acc[0][jd] : z4);)                      \
        SI_DBG_MFMA(acc[1][jd] = lr_mfma16(fa[ks & 1][jd], XB[1][ks >> 1][ks & 1], ks ? acc[1][jd] : z4);)                      \
      }                                                                                                                         \
      SI_DBG_DMA(if (ks < KL) { ISSUE; })                                                                                       \
      if (ks >= 5 && ks < 9) { const int u = ks - 5; PREV; }                                                                    \
      SI_FENCE();                                                                                                               \
    }                                                                                                                           \
  }

  f32x4 accs[2][2][4];
  // emit unit u (row set u >> 1, 32 columns c0 + 32 (u & 1) ..) of a piece's accumulators: + bias (LDS row `bias`), 16 bits, one 16-byte
  // store per lane to dst (row stride ld elements)
  auto emit_unit = [&](f32x4 (&acc)[2][4], const float* bias, T* dst, int ld, int c0, int u) __attribute__((always_inline)) {
    const int rs = u >> 1, q = u & 1;
    f32x4 a = acc[rs][2 * q] + *reinterpret_cast<const f32x4*>(bias + c0 + (2 * q) * 16 + 4 * fq);
    f32x4 b = acc[rs][2 * q + 1] + *reinterpret_cast<const f32x4*>(bias + c0 + (2 * q + 1) * 16 + 4 * fq);
    xa_swap_rows16(a, b);
    const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    const uint4 pk = lr_pack8<T>(v);
#ifdef SI_DBG_NOSTORE      // (timing experiment, tools/build_variant.sh: keep the arithmetic, drop the store)
    if (pk.x == 0x12345678u && pk.y == 0x9abcdef0u)
#endif
    *reinterpret_cast<uint4*>(dst + (size_t)(m_w0 + 16 * rs + fr) * ld + c0 + (2 * q + odd) * 16 + ch8) = pk;
  };
#define SI_NONE do { } while (0)

  // ================= stage 1: x1 = h Wp^T + bp ===========================================================================
  // The order of the column pieces is free (each is emitted on its own): block b starts at piece (b / 8) mod 5 resp. mod nq and wraps,
  // so the ~32 blocks an XCD runs at once (block ids b = xcd mod 8) ask its L2 for DIFFERENT weight lines at any moment instead of all
  // for the same 40 KB.
  // NWAIT: in piece i the loads of piece i + 2 are issued (5, k-steps 0 .. 4), then the stores of piece i - 1 (4, k-steps 5 .. 8)
  T* x1 = reinterpret_cast<T*>(P.x1);
  SI_PIECE(0, xf, 5, issue_p(2, PP(2), ks), 0, SI_NONE);
  SI_PIECE(1, xf, 5, issue_p(0, PP(3), ks), 1, emit_unit(accs[0], par, x1, C, PP(0) * 64, u));
  SI_PIECE(2, xf, 9, issue_p(1, PP(4), ks), 0, emit_unit(accs[1], par, x1, C, PP(1) * 64, u));
  SI_PIECE(0, xf, 13, issue_q(2, QN(0), ks), 1, emit_unit(accs[0], par, x1, C, PP(2) * 64, u));
  SI_PIECE(1, xf, 13, issue_q(0, QN(1), ks), 0, emit_unit(accs[1], par, x1, C, PP(3) * 64, u));
#pragma unroll
  for (int u = 0; u < 4; ++u) emit_unit(accs[0], par, x1, C, PP(4) * 64, u);

  // ---- the wave's x1 rows come back as the second stage's B operands (natural k order): they left rounded to 16 bits, which is what
  //      the unfused path normalises; the lines were written by this wave (vmcnt(0): its stores have reached the L2, and every older
  //      operation with them) and never read before, so the loads cannot find a stale copy in this CU's L1.  Two-pass LayerNorm in registers.
  SI_STAMP(62);
  xa_wait_vmcnt<0>();
  SI_STAMP(63);
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
  // this block's n-th .. (n + 2)-th pieces in slots 2, 0, 1 (five pieces of stage 1: 5 % 3 == 2), accumulator sets 1, 0, 1 / 0, 1, 0 by the
  // parity of n; the previous piece's columns are emitted inside each
#define SI_TRIPLE(n, W0, W1, W2, A, FIRST)                                                                                                  \
  SI_PIECE(2, xf, W0, issue_q(1, QN((n) + 2), ks, (n) + 2 < nq), A, if (!(FIRST)) emit_unit(accs[(A) ^ 1], par + C, qkv, P.ld_qkv, QN((n) - 1) * 64, u)); \
  SI_PIECE(0, xf, W1, issue_q(2, QN((n) + 3), ks, (n) + 3 < nq), (A) ^ 1, emit_unit(accs[A], par + C, qkv, P.ld_qkv, QN(n) * 64, u));        \
  SI_PIECE(1, xf, W2, issue_q(0, QN((n) + 4), ks, (n) + 4 < nq), A, emit_unit(accs[(A) ^ 1], par + C, qkv, P.ld_qkv, QN((n) + 1) * 64, u));
  // (after the reload nothing is in flight: the first two waits are formal, the third sees only one piece's stores behind its loads)
  SI_TRIPLE(0, 13, 13, 9, 0, true)
#pragma unroll 1
  for (int n = 3; n < nq; n += 6) {        // two triples per trip: the accumulator parity repeats every two
    SI_TRIPLE(n, 13, 13, 13, 1, false)
    if (n + 3 < nq) { SI_TRIPLE(n + 3, 13, 13, 13, 0, false) }
  }
  if ((nq / 3) & 1) {                      // the last piece's columns
#pragma unroll
    for (int u = 0; u < 4; ++u) emit_unit(accs[0], par + C, qkv, P.ld_qkv, QN(nq - 1) * 64, u);
  } else {
#pragma unroll
    for (int u = 0; u < 4; ++u) emit_unit(accs[1], par + C, qkv, P.ld_qkv, QN(nq - 1) * 64, u);
  }
#undef SI_TRIPLE
  SI_STAMP(1);                             // (all pieces multiplied and emitted)
  xa_wait_vmcnt<0>();                      // (the dead prefetches of the last two pieces)
#undef SI_PIECE
#undef PP
#undef QN
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
  const bool gn = a->gn_part != nullptr;
  if (gn) {
    if (!a->gn_gamma || !a->gn_beta || a->gn_chunks <= 0 || a->gn_hw <= 0) return LR_E_ARG;
    if (a->gn_hw % SI_ROWS || a->M % a->gn_hw) return LR_E_UNSUPPORTED;      // a block stays inside one sample
    if (((uintptr_t)a->gn_part & 7) || (((uintptr_t)a->gn_gamma | (uintptr_t)a->gn_beta) & 3)) return LR_E_ALIGN;
  }
  P.gn_part = a->gn_part; P.gn_gamma = a->gn_gamma; P.gn_beta = a->gn_beta; P.gn_chunks = a->gn_chunks; P.gn_hw = a->gn_hw; P.gn_eps = a->gn_eps;
#ifdef SI_TRACE
  P.trace = g_si_trace;
#endif
  const size_t smem = 3 * SI_SLOT + (size_t)(SI_C + a->NQ + 2 * SI_C + 64) * sizeof(float);
  static unsigned long long attr_done[2] = {0, 0};
  if (gn) {
    if (lr_attr_needed(&attr_done[1]))
      hipFuncSetAttribute(reinterpret_cast<const void*>(stin_block_kernel<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((stin_block_kernel<T, true>), dim3(a->M / SI_ROWS), dim3(SI_THREADS), smem, (hipStream_t)s, P);
  } else {
    if (lr_attr_needed(&attr_done[0]))
      hipFuncSetAttribute(reinterpret_cast<const void*>(stin_block_kernel<T, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((stin_block_kernel<T, false>), dim3(a->M / SI_ROWS), dim3(SI_THREADS), smem, (hipStream_t)s, P);
  }
  return lr_launch_status();
}

extern "C" int lr_stin_block_f16(const lr_stin_args* a, lr_stream_t s) { return stin_block_t<f16>(a, s); }
extern "C" int lr_stin_block_bf16(const lr_stin_args* a, lr_stream_t s) { return stin_block_t<bf16>(a, s); }
