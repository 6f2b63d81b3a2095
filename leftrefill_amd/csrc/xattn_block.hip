// Fused cross-attention block of BasicTransformerBlock (reference ldm/modules/attention.py:165-196 + 281):
//
//     out = x + to_out( softmax( (LayerNorm(x) Wq^T) K^T * scale ) V )            K, V = projections of the <= 96 context tokens
//
// as ONE kernel for C = 320 (level 0 of the SD2 UNet: 5 heads of 64), instead of  LayerNorm-folded to_q GEMM -> 77-key
// attention -> to_out GEMM (+ residual): x is read once and `out` written once, q / the attention output never leave the
// registers.  The three kernels it replaces are memory / latency bound (K = C = 320 gives 5 K-steps per GEMM tile).
//
// Work split.  Block = 8 waves = 128 rows of ONE sample; each wave owns 16 rows and chains four matrix products through
// its registers with 16x16x32 MFMAs in the "swapped" form (weights are the A operand, the wave's rows are the B operand:
// lane (fr = lane & 15, fq = lane >> 4) holds D[n = 4 fq + r][row = fr]):
//     q_h^T [64 x 16]   = Wq_h  [64 x 320] . xn^T   B = the LayerNorm-ed rows, loaded ONCE from global into B-operand form
//     S^T   [keys x 16] = K_h   [keys x 64] . q_h^T B = the q accumulators of tiles (2p, 2p+1) packed to fp16: k-slot (fq, i)
//                                                       holds d = 32 p + 16 (i >> 2) + 4 fq + (i & 3)  =: pi
//     O_h^T [64 x 16]   = V_h^T [64 x keys] . P^T   B = the S^T accumulators after the softmax, key order pi likewise
//     out^T [320 x 16] += Wo_h  [320 x 64] . O_h^T  B = the O^T accumulators, d order pi
// The MFMA sums over k, so an operand pair only has to AGREE on the k order: K's d columns, V^T's key slots and Wo's
// in-head columns are stored in the order pi (host / pack kernel), which makes every A fragment one ds_read_b128.
// A row's softmax statistics live in the four lanes fr + 16 fq: in-lane reductions + v_permlane16_swap / permlane32_swap.
//
// Weights stream through a 3-slot LDS ring (40 KB slots, 16-byte LDS-DMA with the GEMM's source-side XOR swizzle:
// 128-byte rows, slot of chunk c in row r is c ^ ((r >> 1) & 7)), one piece per step, loads two steps ahead, counted
// vmcnt + one barrier per step; per head:  Wq_h (5 sub-tiles [64 x 64])  ->  K_h [128 x 64] | V_h^T (2 x [64 x 64])  ->  Wo_h [320 x 64].
// LayerNorm: gamma / beta are folded into Wq / bq at pack time (packing.fold_layernorm), the kernel normalises its rows
// in registers with a two-pass mean / variance (no producer statistics needed).
// Epilogue: out^T + bias -> fp16 -> wave-private LDS rows -> whole 16-byte pieces: + residual (x, L2-hot), store, and the
// per-row (sum, sumsq) of the rounded output for the LayerNorm folded into the GEGLU projection that follows.
#include "chain_common.h"
#include <type_traits>

#define XA_C 320
#define XA_HEADS 5
#define XA_ROWS 128
#define XA_THREADS 512
#define XA_SLOT 40960
#define XA_PITCH 656          // bytes per staged output row (640 + 16: the 16 rows of a wave start in distinct banks)
#define XA_VT_OFF 16384       // V^T sub-tiles inside the K|V slot

struct XattnParams {
  const void* x; const void* wq; const float* bq; const void* k; const void* vt; const void* wo; const float* bo;
  void* out; float* st_out;
  const void* pre_a; const void* pre_w; const float* pre_b;      // PRE: x1 = pre_a pre_w^T + pre_b + x runs in front (see below)
  int M, HW, Lc, ldk, nblocks;
  float eps, c;               // c = scale * log2(e)
  unsigned k_bytes, vt_bytes;
#ifdef LR_XATTN_TRACE
  unsigned long long* trace;   // developer build only: shader-clock stamps [block][8 waves][24] (tools/trace_xattn.py)
#endif
};

#ifdef LR_XATTN_TRACE
#define XA_STAMP(k) do { if (P.trace && lane == 0) P.trace[((size_t)blockIdx.x * 8 + w) * 24 + (k)] = __builtin_readcyclecounter(); } while (0)
static unsigned long long* g_xa_trace = nullptr;
extern "C" void lr_xattn_set_trace(void* p) { g_xa_trace = (unsigned long long*)p; }
#else
#define XA_STAMP(k) do { } while (0)
#endif

// NKT = 16-key tiles of the context (5: Lc <= 80, 6: Lc <= 96)
// PRE: the out-projection of the preceding self-attention runs in front, in the same registers (reference attention.py:280-281):
//     x1 = a Wo1^T + bo1 + x          (a = pre_a: the self-attention output, Wo1 = pre_w in its natural [320][320] layout)
//     out = x1 + to_out(attention(LayerNorm(x1) Wq, K, V))
// five more ring steps (Wo1 as five 64-row pieces) in front of the head loop; x1 never goes to memory: its fp32 accumulators are
// rounded to fp16 (what the unfused path stores), normalised, and handed to the q projection as B operands in accumulator order
// (Wq's columns are stored in the order pi for this variant), and the rounded x1 stays in 40 registers as the final residual.
template <typename T, int NKT, bool PRE>
__global__ __launch_bounds__(XA_THREADS) void xattn_block_kernel(const XattnParams P) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int C = XA_C, NT = C / 16, KL = C / 64;   // 20 output tiles, 5 lines of 128 B per row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* par = reinterpret_cast<float*>(smem + 3 * XA_SLOT);      // [3][C]: bq | bo | bo1 (PRE)

  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  int bid = blockIdx.x;
  {   // XCD-aware bijective remap: consecutive row blocks (one sample's K / V) stay on one XCD's L2
    const int q = P.nblocks >> 3, r = P.nblocks & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int m0 = bid * XA_ROWS;
  const int b = m0 / P.HW;
  const int m_w0 = m0 + w * 16;
  XA_STAMP(0);

  // ---- this wave's 16 rows in B-operand form: lane (fr, fq) holds x[m_w0 + fr][64 t5 + 32 u + 8 fq .. + 7] (the four fq lanes of a
  //      row read 64 consecutive bytes; the matching weight fragment of lane group fq is chunk 4 u + fq of sub-tile t5 -- consecutive
  //      chunks across fq like the GEMM's fragments, which is what keeps the ds_read_b128 conflict-free: with 16 fq + 8 u, i.e. chunk
  //      2 fq + u, every fragment read of the q projection took two LDS passes)
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
  if (t < (PRE ? 3 : 2) * (C / 4)) {      // biases -> LDS (the loop below issues no register loads: they would drain the LDS-DMA queue)
    const float* src = t < C / 4 ? P.bq + 4 * t : t < C / 2 ? P.bo + 4 * (t - C / 4) : P.pre_b + 4 * (t - C / 2);
    *reinterpret_cast<f32x4*>(par + 4 * t) = *reinterpret_cast<const f32x4*>(src);
  }

  // ---- weight ring
  const __amdgpu_buffer_rsrc_t rsQ = uniform_rsrc(P.wq, (size_t)C * C * 2);
  const __amdgpu_buffer_rsrc_t rsO = uniform_rsrc(P.wo, (size_t)C * C * 2);
  const __amdgpu_buffer_rsrc_t rsK = uniform_rsrc(P.k, P.k_bytes);
  const __amdgpu_buffer_rsrc_t rsV = uniform_rsrc(P.vt, P.vt_bytes);
  const unsigned OOB = 0x80000000u;
  const int lrow = w * 8 + (lane >> 3);                      // row of an 8-row LDS-DMA group this lane fills
  const int lchunk = (lane & 7) ^ ((lrow >> 1) & 7);         // source-side swizzle (bits 1..3 of the row: same for row + 64 i)
  // one 1 KiB LDS-DMA instruction of a piece (i = 0 .. 4 for Wq / Wo, 0 .. 3 for K|V): the main loop places them one at a
  // time between its MFMA groups (an LDS-DMA issue costs its wave ~60 cycles there, 100-185 in a burst -- MI355X_MICROARCH.md)
  auto issue_wq = [&](int slot, int h, int i) __attribute__((always_inline)) {   // 5 sub-tiles [64 d x 64 k]; wave w: rows 8 w .. + 7 of each
    const unsigned v0 = (unsigned)(((h * 64 + lrow) * C + lchunk * 8) * 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lptr_t)(smem + slot * XA_SLOT + (i * 64 + w * 8) * 128), 16, v0, i * 128, 0, 0);
  };
  auto issue_wo = [&](int slot, int h, int i) __attribute__((always_inline)) {   // piece h of Wo ([heads][320 n][64 k], 40 KB contiguous); rows 64 i + 8 w ..
    const unsigned v0 = (unsigned)(((h * C + lrow) * 64 + lchunk * 8) * 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsO, (lptr_t)(smem + slot * XA_SLOT + (i * 64 + w * 8) * 128), 16, v0, i * 64 * 128, 0, 0);
  };
  auto issue_kv = [&](int slot, int h, int i) __attribute__((always_inline)) {   // K [128 keys x 64 d] | V^T 2 x [64 d x 64 key slots]
    if (i < 2) {
      const int key = i * 64 + lrow;
      const unsigned vk = key < P.Lc ? (unsigned)((((size_t)b * P.Lc + key) * P.ldk + h * 64 + lchunk * 8) * 2) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lptr_t)(smem + slot * XA_SLOT + (i * 64 + w * 8) * 128), 16, vk, 0, 0, 0);
    } else {
      const unsigned vv = (unsigned)((((b * XA_HEADS + h) * 2) * 64 + lrow) * 128 + lchunk * 16);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lptr_t)(smem + slot * XA_SLOT + XA_VT_OFF + ((i - 2) * 64 + w * 8) * 128), 16, vv,
                                               (i - 2) * 64 * 128, 0, 0);
    }
  };
  const __amdgpu_buffer_rsrc_t rsP = uniform_rsrc(PRE ? P.pre_w : P.wq, (size_t)C * C * 2);
  auto issue_pre = [&](int slot, int p, int i) __attribute__((always_inline)) {   // rows 64 p .. + 63 of Wo1 as 5 sub-tiles [64 x 64 k]
    const unsigned v0 = (unsigned)(((p * 64 + lrow) * C + lchunk * 8) * 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP, (lptr_t)(smem + slot * XA_SLOT + (i * 64 + w * 8) * 128), 16, v0, i * 128, 0, 0);
  };
  if constexpr (PRE) {      // pieces 0, 1 of Wo1 -> slots 1, 2 (the five pre steps use slots 1, 2, 0, 1, 2: the head loop then finds its own)
#pragma unroll
    for (int i = 0; i < KL; ++i) issue_pre(1, 0, i);
#pragma unroll
    for (int i = 0; i < KL; ++i) issue_pre(2, 1, i);
  } else {
#pragma unroll
    for (int i = 0; i < KL; ++i) issue_wq(0, 0, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_kv(1, 0, i);
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

  XA_STAMP(1);                                // rows loaded and normalised
  f32x4 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const int sw = (fr >> 1) & 7;      // swizzle of every fragment row this lane reads (rows are fr + a multiple of 16)
  auto frag = [&](const char* base, int row, int chunk) -> vec8<T> {
    return *reinterpret_cast<const vec8<T>*>(base + row * 128 + ((chunk ^ sw) << 4));
  };
#define XA_FENCE() __builtin_amdgcn_sched_barrier(0)
  if constexpr (PRE) {
    // ================= pre steps: x1^T = Wo1 a^T, 64 output channels (4 tiles) per step ==============================
#pragma unroll
    for (int p = 0; p < KL; ++p) {
      const int slot = (p + 1) % 3;
      xa_wait_vmcnt<5>();                     // piece p landed (the next one may be in flight)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const char* Ws = smem + slot * XA_SLOT;
      vec8<T> fa[2][4];
      auto rd = [&](int ks, vec8<T> (&f)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int jd = 0; jd < 4; ++jd) f[jd] = frag(Ws + (ks >> 1) * 64 * 128, jd * 16 + fr, 4 * (ks & 1) + fq);
      };
      rd(0, fa[0]);
#pragma unroll
      for (int ks = 0; ks < 2 * KL; ++ks) {
        if (ks + 1 < 2 * KL) rd(ks + 1, fa[(ks + 1) & 1]);
        XA_FENCE();
#pragma unroll
        for (int jd = 0; jd < 4; ++jd) acc[4 * p + jd] = lr_mfma16(fa[ks & 1][jd], xf[ks >> 1][ks & 1], acc[4 * p + jd]);
        if (ks < KL) {                        // two steps ahead: Wo1 pieces 2 .. 4, then the first head's Wq / K|V pieces
          if (p + 2 < KL) issue_pre((p + 3) % 3, p + 2, ks);
          else if (p + 2 == KL) issue_wq(0, 0, ks);
          else if (ks < 4) issue_kv(1, 0, ks);
        }
        XA_FENCE();
      }
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

  // Fragment reads run one MFMA group ahead of their use (two register sets), and `sched_barrier(0)` pins
  //   [ds_reads of group g + 1] -> [MFMAs of group g (independent accumulators)] -> [one LDS-DMA issue]:
  // left to itself hipcc orders each product as one dependent accumulator chain with the read of every fragment issued one MFMA
  // before its use (~100 cycles per MFMA instead of ~17).
#pragma unroll 1
  for (int h = 0; h < XA_HEADS; ++h) {
    const bool more = h + 1 < XA_HEADS;
    // ================= step A: q_h^T = Wq_h xn^T  (10 groups = k-steps of 4 MFMAs) ================================
    xa_wait_vmcnt<4>();                       // Wq_h landed (K|V_h may still be in flight)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (first head: the bias rows written to LDS above)
    __builtin_amdgcn_s_barrier();
    XA_STAMP(2 + 4 * h);
    f32x4 qa[4] = {z4, z4, z4, z4};
    {
      const char* Ws = smem;                  // slot 0
      vec8<T> fa[2][4];
      auto rd = [&](int ks, vec8<T> (&f)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int jd = 0; jd < 4; ++jd) f[jd] = frag(Ws + (ks >> 1) * 64 * 128, jd * 16 + fr, 4 * (ks & 1) + fq);
      };
      rd(0, fa[0]);
#pragma unroll
      for (int ks = 0; ks < 2 * KL; ++ks) {
        if (ks + 1 < 2 * KL) rd(ks + 1, fa[(ks + 1) & 1]);
        XA_FENCE();
#pragma unroll
        for (int jd = 0; jd < 4; ++jd) qa[jd] = lr_mfma16(fa[ks & 1][jd], xf[ks >> 1][ks & 1], qa[jd]);
        if (ks < KL) issue_wo(2, h, ks);      // Wo_h -> slot 2 (its last readers passed the barrier above)
        XA_FENCE();
      }
    }
#pragma unroll
    for (int jd = 0; jd < 4; ++jd) qa[jd] += *reinterpret_cast<const f32x4*>(par + h * 64 + jd * 16 + 4 * fq);
    const vec8<T> qb0 = xa_pack<T>(qa[0], qa[1]), qb1 = xa_pack<T>(qa[2], qa[3]);

    // ================= step B: S^T = K_h q_h^T, softmax, O_h^T = V_h^T P^T =========================================
    XA_STAMP(3 + 4 * h);
    xa_wait_vmcnt<5>();                       // K|V_h landed (Wo_h in flight)
    __builtin_amdgcn_s_barrier();
    XA_STAMP(4 + 4 * h);
    const char* Ks = smem + XA_SLOT;
    f32x4 sa[NKT];
    {
      vec8<T> k0[NKT], k1[NKT];
#pragma unroll
      for (int jk = 0; jk < NKT; ++jk) k0[jk] = frag(Ks, jk * 16 + fr, fq);
#pragma unroll
      for (int jk = 0; jk < NKT; ++jk) k1[jk] = frag(Ks, jk * 16 + fr, 4 + fq);
      XA_FENCE();
#pragma unroll
      for (int jk = 0; jk < NKT; ++jk) sa[jk] = lr_mfma16(k0[jk], qb0, z4);
      if (more) { issue_wq(0, h + 1, 0); issue_wq(0, h + 1, 1); }      // Wq_{h+1} -> slot 0
      XA_FENCE();
#pragma unroll
      for (int jk = 0; jk < NKT; ++jk) sa[jk] = lr_mfma16(k1[jk], qb1, sa[jk]);
      if (more) { issue_wq(0, h + 1, 2); }
      XA_FENCE();
    }
    // V^T fragments of the first two key blocks are requested before the softmax (their latency hides under it)
    const char* Vs = Ks + XA_VT_OFF;
    vec8<T> v0[4], v1[4];
#pragma unroll
    for (int jd = 0; jd < 4; ++jd) v0[jd] = frag(Vs, jd * 16 + fr, fq);
#pragma unroll
    for (int jd = 0; jd < 4; ++jd) v1[jd] = frag(Vs, jd * 16 + fr, 4 + fq);
    float mx = -INFINITY;
#pragma unroll
    for (int jk = 0; jk < NKT; ++jk)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (jk * 16 + 4 * fq + r >= P.Lc) sa[jk][r] = -INFINITY;
        mx = fmaxf(mx, sa[jk][r]);
      }
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
    l = xa_row4_sum(l);
    vec8<T> pb[3];
    pb[0] = xa_pack<T>(sa[0], sa[1]);
    pb[1] = xa_pack<T>(sa[2], sa[3]);
    pb[2] = xa_pack<T>(sa[4], NKT > 5 ? sa[NKT - 1] : z4);
    f32x4 oa[4];
    {
      vec8<T> v2[4];
      XA_FENCE();
#pragma unroll
      for (int jd = 0; jd < 4; ++jd) v2[jd] = frag(Vs + 64 * 128, jd * 16 + fr, fq);
#pragma unroll
      for (int jd = 0; jd < 4; ++jd) oa[jd] = lr_mfma16(v0[jd], pb[0], z4);
      if (more) issue_wq(0, h + 1, 3);
      XA_FENCE();
#pragma unroll
      for (int jd = 0; jd < 4; ++jd) oa[jd] = lr_mfma16(v1[jd], pb[1], oa[jd]);
      if (more) issue_wq(0, h + 1, 4);
      XA_FENCE();
#pragma unroll
      for (int jd = 0; jd < 4; ++jd) oa[jd] = lr_mfma16(v2[jd], pb[2], oa[jd]);
      XA_FENCE();
    }
    const float inv = __builtin_amdgcn_rcpf(l);
#pragma unroll
    for (int jd = 0; jd < 4; ++jd) oa[jd] *= inv;
    const vec8<T> ob0 = xa_pack<T>(oa[0], oa[1]), ob1 = xa_pack<T>(oa[2], oa[3]);

    // ================= step C: out^T += Wo_h O_h^T  (10 groups of 4 MFMAs) ========================================
    if (more) xa_wait_vmcnt<5>(); else xa_wait_vmcnt<0>();     // Wo_h landed (Wq_{h+1} in flight)
    __builtin_amdgcn_s_barrier();
    XA_STAMP(5 + 4 * h);
    {
      const char* Os = smem + 2 * XA_SLOT;
      vec8<T> fa[2][4];
      // group g: p = g / 5, output tiles 4 (g % 5) .. + 3
      auto rd = [&](int g, vec8<T> (&f)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) f[q] = frag(Os, (4 * (g % 5) + q) * 16 + fr, 4 * (g / 5) + fq);
      };
      rd(0, fa[0]);
#pragma unroll
      for (int g = 0; g < 10; ++g) {
        if (g + 1 < 10) rd(g + 1, fa[(g + 1) & 1]);
        XA_FENCE();
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[4 * (g % 5) + q] = lr_mfma16(fa[g & 1][q], g < 5 ? ob0 : ob1, acc[4 * (g % 5) + q]);
        if (more && g < 4) issue_kv(1, h + 1, g);     // K|V_{h+1} -> slot 1
        XA_FENCE();
      }
    }
  }
#undef XA_FENCE

  // ---- epilogue: (acc + bias) -> fp16 -> this wave's 16 LDS rows -> 16-byte pieces: + x, store, row statistics
  XA_STAMP(22);
  __syncthreads();                            // every wave is done with the ring
  char* stg = smem + w * (16 * XA_PITCH);
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const f32x4 v = acc[j] + *reinterpret_cast<const f32x4*>(par + C + j * 16 + 4 * fq);
    vec4<T> hv;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      hv[r] = (T)v[r];
      if constexpr (PRE) hv[r] = (T)((float)hv[r] + (float)xres[j][r]);      // + x1 (same two roundings as the unfused kernels)
    }
    *reinterpret_cast<vec4<T>*>(stg + fr * XA_PITCH + (j * 16 + 4 * fq) * 2) = hv;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int row = lane >> 2, sub = lane & 3;
  const T* xr = reinterpret_cast<const T*>(P.x) + (size_t)(m_w0 + row) * C;
  T* orow = reinterpret_cast<T*>(P.out) + (size_t)(m_w0 + row) * C;
  float s1 = 0.f, s2 = 0.f;
  constexpr int NP = C / 32;                  // 16-byte pieces per lane (4 lanes per row)
  uint4 rx[NP];
  if constexpr (!PRE) {
#pragma unroll
    for (int it = 0; it < NP; ++it) rx[it] = *reinterpret_cast<const uint4*>(xr + (sub + 4 * it) * 8);
  }
#pragma unroll
  for (int it = 0; it < NP; ++it) {
    const int piece = sub + 4 * it;
    float a[8], e[8];
    uint4 pk = *reinterpret_cast<const uint4*>(stg + row * XA_PITCH + piece * 16);
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
#ifdef LR_XATTN_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  XA_STAMP(23);
#endif
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// V [B * Lc][ldv] (head h = columns h*64 .. + 64) -> V^T pack [B][heads][2][64 d][64 key slots], key slot 32 g + 8 fq + i
// holds key 32 g + 16 (i >> 2) + 4 fq + (i & 3) (the S^T accumulator order, see above), keys >= Lc are zero.
// grid = (heads, B), block = 256.  Runs once per context, not per step.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void xattn_pack_vt_kernel(const T* __restrict__ v, int ldv, T* __restrict__ vt, int heads, int Lc) {
  __shared__ T tile[128][72];
  const int t = threadIdx.x, h = blockIdx.x, b = blockIdx.y;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int key = i * 32 + (t >> 3), j = t & 7;
    uint4 u = make_uint4(0, 0, 0, 0);
    if (key < Lc) u = *reinterpret_cast<const uint4*>(v + ((size_t)b * Lc + key) * ldv + h * 64 + j * 8);
    *reinterpret_cast<uint4*>(&tile[key][j * 8]) = u;
  }
  __syncthreads();
  const int d = t >> 2, g = t & 3;
  T r[32];
#pragma unroll
  for (int sl = 0; sl < 32; ++sl) {
    const int fq = sl >> 3, i = sl & 7;
    r[sl] = tile[32 * g + 16 * (i >> 2) + 4 * fq + (i & 3)][d];
  }
  T* dst = vt + ((((size_t)b * heads + h) * 2 + (g >> 1)) * 64 + d) * 64 + (g & 1) * 32;
#pragma unroll
  for (int q = 0; q < 4; ++q) *reinterpret_cast<uint4*>(dst + 8 * q) = *reinterpret_cast<const uint4*>(&r[8 * q]);
}

template <typename T>
static int xattn_pack_vt_t(const lr_half* v, int ldv, lr_half* vt, int B, int heads, int Lc, lr_stream_t s) {
  if (!v || !vt || B <= 0 || heads <= 0 || Lc <= 0) return LR_E_ARG;
  if (Lc > 96) return LR_E_UNSUPPORTED;
  if (ldv % 8 || (((uintptr_t)v | (uintptr_t)vt) & 15)) return LR_E_ALIGN;
  hipLaunchKernelGGL(xattn_pack_vt_kernel<T>, dim3(heads, B), dim3(256), 0, (hipStream_t)s, (const T*)v, ldv, (T*)vt, heads, Lc);
  return lr_launch_status();
}

template <typename T>
static int xattn_block_t(const lr_xattn_args* a, lr_stream_t s) {
  if (!a || !a->x || !a->wq || !a->bq || !a->k || !a->vt || !a->wo || !a->bo || !a->out) return LR_E_ARG;
  if (a->M <= 0 || a->HW <= 0 || a->Lc <= 0 || a->ldk <= 0) return LR_E_ARG;
  const bool wide = a->C == 640 && a->heads == 10;      // level 1: xattn_block640.hip (64-row blocks)
  if ((!wide && (a->C != XA_C || a->heads != XA_HEADS)) || a->Lc > 96) return LR_E_UNSUPPORTED;
  if (!wide && (a->M % XA_ROWS || a->HW % XA_ROWS || a->M % a->HW)) return LR_E_UNSUPPORTED;     // a block stays inside one sample
  if (a->ldk % 8) return LR_E_ALIGN;
  if (((uintptr_t)a->x | (uintptr_t)a->wq | (uintptr_t)a->bq | (uintptr_t)a->k | (uintptr_t)a->vt | (uintptr_t)a->wo |
       (uintptr_t)a->bo | (uintptr_t)a->out) & 15)
    return LR_E_ALIGN;
  if (a->stats_out && ((uintptr_t)a->stats_out & 7)) return LR_E_ALIGN;
  if (a->pre_a) {
    if (!a->pre_w || !a->pre_b) return LR_E_ARG;
    if (((uintptr_t)a->pre_a | (uintptr_t)a->pre_w | (uintptr_t)a->pre_b) & 15) return LR_E_ALIGN;
  }
  if (wide) return lr_xattn640_launch(a, std::is_same<T, bf16>::value ? 1 : 0, s);
  const int B = a->M / a->HW;
  const int64_t kb = (int64_t)B * a->Lc * a->ldk * 2, vb = (int64_t)B * XA_HEADS * 2 * 64 * 128;
  if (kb >= ((int64_t)1 << 31) || vb >= ((int64_t)1 << 31)) return LR_E_UNSUPPORTED;
  XattnParams P;
  P.x = a->x; P.wq = a->wq; P.bq = a->bq; P.k = a->k; P.vt = a->vt; P.wo = a->wo; P.bo = a->bo; P.out = a->out;
  P.st_out = a->stats_out;
  P.M = a->M; P.HW = a->HW; P.Lc = a->Lc; P.ldk = a->ldk; P.nblocks = a->M / XA_ROWS;
  P.eps = a->ln_eps; P.c = a->scale * 1.44269504088896340736f;
  P.k_bytes = (unsigned)kb; P.vt_bytes = (unsigned)vb;
#ifdef LR_XATTN_TRACE
  P.trace = g_xa_trace;
#endif
  const size_t smem = 3 * XA_SLOT + 3 * XA_C * sizeof(float);
  const bool six = a->Lc > 80, pre = a->pre_a != nullptr;
  P.pre_a = a->pre_a; P.pre_w = a->pre_w; P.pre_b = a->pre_b;
  const void* fns[4] = {reinterpret_cast<const void*>(xattn_block_kernel<T, 5, false>), reinterpret_cast<const void*>(xattn_block_kernel<T, 6, false>),
                        reinterpret_cast<const void*>(xattn_block_kernel<T, 5, true>), reinterpret_cast<const void*>(xattn_block_kernel<T, 6, true>)};
  static unsigned long long attr_done[4] = {0, 0, 0, 0};
  const int v = (pre ? 2 : 0) + (six ? 1 : 0);
  if (lr_attr_needed(&attr_done[v])) {
    hipFuncSetAttribute(fns[v], hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  const dim3 grid(P.nblocks), block(XA_THREADS);
  if (v == 0) hipLaunchKernelGGL((xattn_block_kernel<T, 5, false>), grid, block, smem, (hipStream_t)s, P);
  else if (v == 1) hipLaunchKernelGGL((xattn_block_kernel<T, 6, false>), grid, block, smem, (hipStream_t)s, P);
  else if (v == 2) hipLaunchKernelGGL((xattn_block_kernel<T, 5, true>), grid, block, smem, (hipStream_t)s, P);
  else hipLaunchKernelGGL((xattn_block_kernel<T, 6, true>), grid, block, smem, (hipStream_t)s, P);
  return lr_launch_status();
}

extern "C" int lr_xattn_block_f16(const lr_xattn_args* a, lr_stream_t s) { return xattn_block_t<f16>(a, s); }
extern "C" int lr_xattn_block_bf16(const lr_xattn_args* a, lr_stream_t s) { return xattn_block_t<bf16>(a, s); }
extern "C" int lr_xattn_pack_vt_f16(const lr_half* v, int ldv, lr_half* vt, int B, int heads, int Lc, lr_stream_t s) { return xattn_pack_vt_t<f16>(v, ldv, vt, B, heads, Lc, s); }
extern "C" int lr_xattn_pack_vt_bf16(const lr_half* v, int ldv, lr_half* vt, int B, int heads, int Lc, lr_stream_t s) { return xattn_pack_vt_t<bf16>(v, ldv, vt, B, heads, Lc, s); }
