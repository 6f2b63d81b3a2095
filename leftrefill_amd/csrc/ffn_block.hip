// Fused feed-forward block of BasicTransformerBlock (reference ldm/modules/attention.py:51-78 + 282):
//
//     out = x + W2 ( u * gelu_erf(g) ) + b2,      [u | g] = LayerNorm(x) W1^T + b1          (FeedForward with GEGLU, glu=True)
//
// as ONE kernel for C = 320 (level 0 of the SD2 UNet, hidden width H = 1280): the [M, 4C] hidden activation never exists in
// memory.  Unfused, the GEGLU projection writes 168 MB and the second Linear reads them back at M = 65536 -- both GEMMs have
// K-loops of 5 / 20 steps under epilogues as long as their main loops.
//
// Register-chained like xattn_block.hip (16x16x32 MFMAs in the swapped form: weights = A operand from LDS, rows = B operand from
// registers), but two waves share 32 rows: block = 8 waves = 4 pairs = 128 rows; a wave computes half of every hidden chunk and owns
// half of the output columns FOR BOTH 16-row tiles, so each weight fragment read from LDS feeds two MFMAs (the kernel is bound by
// the LDS -> register fragment stream: with one row tile per wave it ran 23 % slower); the halves of the gated hidden chunk cross
// between the two waves through a 4 KB LDS hand-off area (8 bytes per lane and row tile), under the step barrier that exists anyway:
//     [u | g]^T [64 x 16] = W1_c [64 x 320] . xn^T      chunk c = 32 hidden units, rows interleaved [u16 | g16 | u16 | g16]
//                                                        (packing.pack_geglu: value and gate land in the same lane)
//     h^T       [32 x 16] = (u + bu) * gelu(g + bg)      in the accumulator registers, packed to fp16
//     out^T   [320 x 16] += W2[:, chunk] . h^T           B = the h accumulators: k-slot (fq, i) holds hidden unit
//                                                        32 c + 16 (i >> 2) + 4 fq + (i & 3)  (W2's columns are stored in that order)
// Weight ring: 3 LDS slots of 40 KB, one piece per step, loads two steps ahead, counted vmcnt + one barrier per step; per 64
// hidden units:  W1 chunk 2j (5 sub-tiles [64 x 64])  ->  W1 chunk 2j + 1  ->  W2[:, 64 j .. + 63] ([320 x 64]).  W2 is stored piece by
// piece ([H / 64][320][64]) so that every piece is 40 KB of consecutive addresses like the W1 pieces: with the Linear's own [320][H]
// layout the 128-byte rows of a piece sit 2 H bytes apart and all CUs of an XCD hammer half of its L2 channels at the same time.
// LayerNorm: gamma / beta folded into W1 / b1 (packing.fold_layernorm), rows normalised in registers (two-pass).
// Epilogue as in xattn_block.hip: + bias -> fp16 -> wave-private LDS rows -> 16-byte pieces: + x, store, optional row statistics.
#include "chain_common.h"

#define FF_C 320
#define FF_ROWS 128
#define FF_THREADS 512
#define FF_SLOT 40960
#define FF_PITCH 336          // bytes per staged row of a wave's 160 output columns (320 + 16: rows start in distinct banks)
#define FF_PAR_BYTES ((2 * FF_MAX_H + 2 * FF_C) * 4)      // bias rows behind the ring
#define FF_DEPTH 2            // MFMA groups a fragment read runs ahead of its use (3 measured the same)
#define FF_MAX_H 2048          // hidden units whose bias rows fit the LDS region behind the ring

struct FfnParams {
  const void* x; const void* w1; const float* b1; const void* w2; const float* b2; void* out; float* st_out;
  const void* post_w; const float* post_b; const void* post_resid; float* gs_out;      // POST (see the kernel)
  float* gp_out; int gp_hw, gp_chunks;      // POST: per-group sums of the output, [sample][chunk = 128-row block][32][2]
  int M, H, nblocks;
  float eps;
#ifdef LR_FFN_TRACE
  unsigned long long* trace;   // developer build only: shader-clock stamps [block][8 waves][16] (tools/trace_ffn.py)
#endif
};

#ifdef LR_FFN_TRACE
#define FF_STAMP(k) do { if (P.trace && lane == 0) P.trace[((size_t)blockIdx.x * 8 + w) * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
static unsigned long long* g_ff_trace = nullptr;
extern "C" void lr_ffn_set_trace(void* p) { g_ff_trace = (unsigned long long*)p; }
#else
#define FF_STAMP(k) do { } while (0)
#endif

// POST: the Linear that follows the block (SpatialTransformer.proj_out, attention.py:412-419) runs behind it in the same launch:
//     x3 = x + ff(LayerNorm(x));   out = x3 Wp^T + bp + x_in        (Wp = post_w as 64-column pieces in k-slot order, x_in = post_resid)
// x3 never goes to memory: rounded to fp16 like the unfused path stores it, each wave's half of the columns becomes five B operands
// per row tile, the two halves of a pair cross through LDS once, then five more ring steps.  With gs_out the block also emits the
// per-channel (sum, sumsq) of its 128 output rows -- the statistics of the GroupNorm that consumes `out` (lr_groupnorm_finalize, R = 128).
template <typename T, bool POST>
__global__ __launch_bounds__(FF_THREADS) void ffn_block_kernel(const FfnParams P) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int C = FF_C, KL = C / 64;
  constexpr int NTW = C / 32;          // output tiles (16 columns) per wave: half of the row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* par = reinterpret_cast<float*>(smem + 3 * FF_SLOT);      // [2 H] b1 (interleaved like W1's rows) | [C] b2

  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int pr = w >> 1, role = w & 1;   // wave pair (32 rows) and which half of the work this wave does for it
  const int fr = lane & 15, fq = lane >> 4;
  int bid = blockIdx.x;
  {
    const int q = P.nblocks >> 3, r = P.nblocks & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int m_p0 = bid * FF_ROWS + pr * 32;      // first row of the pair
  const int H2 = 2 * P.H;
  char* xch = smem + 3 * FF_SLOT + FF_PAR_BYTES + pr * 4096;      // the pair's hand-off area [chunk 2][row tile 2][lane 64][16 B]
  FF_STAMP(0);

  // ---- the pair's 32 rows in B-operand form (both waves hold them): lane (fr, fq), row tile rt holds
  //      x[m_p0 + 16 rt + fr][64 t5 + 32 u + 8 fq .. + 7]  (weight fragment of lane group fq = chunk 4 u + fq: conflict-free reads,
  //      see xattn_block.hip)
  vec8<T> xf[2][KL][2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const T* xrow = reinterpret_cast<const T*>(P.x) + (size_t)(m_p0 + 16 * rt + fr) * C + 8 * fq;
#pragma unroll
    for (int t5 = 0; t5 < KL; ++t5)
#pragma unroll
      for (int u = 0; u < 2; ++u) xf[rt][t5][u] = *reinterpret_cast<const vec8<T>*>(xrow + 64 * t5 + 32 * u);
  }
  for (int i = t; i < (H2 + (POST ? 2 : 1) * C) / 4; i += FF_THREADS) {      // biases -> LDS (no register loads inside the loop below)
    const float* src = i < H2 / 4 ? P.b1 + 4 * i : i < (H2 + C) / 4 ? P.b2 + 4 * (i - H2 / 4) : P.post_b + 4 * (i - (H2 + C) / 4);
    *reinterpret_cast<f32x4*>(par + 4 * i) = *reinterpret_cast<const f32x4*>(src);
  }

  // ---- weight ring
  const __amdgpu_buffer_rsrc_t rs1 = uniform_rsrc(P.w1, (size_t)H2 * C * 2);
  const __amdgpu_buffer_rsrc_t rs2 = uniform_rsrc(P.w2, (size_t)C * P.H * 2);
  const int lrow = w * 8 + (lane >> 3);
  const int lchunk = (lane & 7) ^ ((lrow >> 1) & 7);
  auto issue_w1 = [&](int slot, int c, int i) __attribute__((always_inline)) {   // rows 64 c .. + 63 of W1 as 5 sub-tiles [64 x 64 k]
    const unsigned v0 = (unsigned)(((c * 64 + lrow) * C + lchunk * 8) * 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lptr_t)(smem + slot * FF_SLOT + (i * 64 + w * 8) * 128), 16, v0, i * 128, 0, 0);
  };
  auto issue_w2 = [&](int slot, int j, int i) __attribute__((always_inline)) {   // piece j of W2 ([H / 64][320][64]: 40 KB contiguous); rows 64 i + 8 w ..
    const unsigned v0 = (unsigned)(((j * C + lrow) * 64 + lchunk * 8) * 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, (lptr_t)(smem + slot * FF_SLOT + (i * 64 + w * 8) * 128), 16, v0, i * 64 * 128, 0, 0);
  };
#pragma unroll
  for (int i = 0; i < KL; ++i) issue_w1(0, 0, i);
#pragma unroll
  for (int i = 0; i < KL; ++i) issue_w1(1, 1, i);

  // ---- LayerNorm of the rows in registers (two-pass); gamma / beta live in W1 / b1
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    __builtin_amdgcn_sched_barrier(0);      // one row tile at a time (register pressure)
    float s = 0.f;
#pragma unroll
    for (int t5 = 0; t5 < KL; ++t5)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) s += (float)xf[rt][t5][u][i];
    const float mean = xa_row4_sum(s) * (1.0f / C);
    float q2 = 0.f;
#pragma unroll
    for (int t5 = 0; t5 < KL; ++t5)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = (float)xf[rt][t5][u][i] - mean; q2 = fmaf(d, d, q2); }
    const float rstd = rsqrtf(xa_row4_sum(q2) * (1.0f / C) + P.eps);
    const float nmr = -mean * rstd;
#pragma unroll
    for (int t5 = 0; t5 < KL; ++t5)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) xf[rt][t5][u][i] = (T)fmaf((float)xf[rt][t5][u][i], rstd, nmr);
  }

  FF_STAMP(1);
  f32x4 acc[NTW][2];
#pragma unroll
  for (int j = 0; j < NTW; ++j) { acc[j][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[j][1] = acc[j][0]; }
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const int sw = (fr >> 1) & 7;
  auto frag = [&](const char* base, int row, int chunk) -> vec8<T> {
    return *reinterpret_cast<const vec8<T>*>(base + row * 128 + ((chunk ^ sw) << 4));
  };
#define FF_FENCE() __builtin_amdgcn_sched_barrier(0)
  // This wave's half of chunk c (32 hidden units): tiles (u, g) of units 16 role .. 16 role + 15 for both row tiles -- every W1
  // fragment read from LDS feeds TWO MFMAs.  The gated result goes to the pair's hand-off area as 4 halves per lane and row tile;
  // the 16 units of the partner come from there in step C.  `issue(i)`: the i-th LDS-DMA instruction of the piece this step prefetches.
  auto proj_chunk = [&](int slot, int c, int cb, auto&& issue) __attribute__((always_inline)) {
    const char* Ws = smem + slot * FF_SLOT;
    f32x4 pa[2][2] = {{z4, z4}, {z4, z4}};      // [u | g][row tile]
    vec8<T> fa[FF_DEPTH + 1][2];              // fragment reads run FF_DEPTH MFMA groups ahead of their use
    auto rd = [&](int ks, vec8<T> (&f)[2]) __attribute__((always_inline)) {
#pragma unroll
      for (int ug = 0; ug < 2; ++ug) f[ug] = frag(Ws + (ks >> 1) * 64 * 128, (2 * role + ug) * 16 + fr, 4 * (ks & 1) + fq);
    };
#pragma unroll
    for (int d = 0; d < FF_DEPTH; ++d) rd(d, fa[d]);
#pragma unroll
    for (int ks = 0; ks < 2 * KL; ++ks) {
      if (ks + FF_DEPTH < 2 * KL) rd(ks + FF_DEPTH, fa[(ks + FF_DEPTH) % (FF_DEPTH + 1)]);
      FF_FENCE();
#pragma unroll
      for (int ug = 0; ug < 2; ++ug)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) pa[ug][rt] = lr_mfma16(fa[ks % (FF_DEPTH + 1)][ug], xf[rt][ks >> 1][ks & 1], pa[ug][rt]);
      if (ks < KL) issue(ks);
      FF_FENCE();
    }
    const f32x4 bu = *reinterpret_cast<const f32x4*>(par + c * 64 + (2 * role) * 16 + 4 * fq);
    const f32x4 bg = *reinterpret_cast<const f32x4*>(par + c * 64 + (2 * role + 1) * 16 + 4 * fq);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const f32x4 u = pa[0][rt] + bu, g = pa[1][rt] + bg;
      const f32x2_t a = lr_gelu_erf2((f32x2_t){g[0], g[1]}), b = lr_gelu_erf2((f32x2_t){g[2], g[3]});
      vec4<T> hv = {(T)(u[0] * a[0]), (T)(u[1] * a[1]), (T)(u[2] * b[0]), (T)(u[3] * b[1])};
      *reinterpret_cast<vec4<T>*>(xch + ((cb * 2 + rt) * 64 + lane) * 16 + role * 8) = hv;
    }
  };

  const int nsuper = P.H / 64;
#pragma unroll 1
  for (int j = 0; j < nsuper; ++j) {
    const bool more = j + 1 < nsuper;
    // ---- step A1: chunk 2j (slot 0); W2 piece j goes to slot 2
    xa_wait_vmcnt<5>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (j < 3) FF_STAMP(2 + 4 * j);
    proj_chunk(0, 2 * j, 0, [&](int i) __attribute__((always_inline)) { issue_w2(2, j, i); });
    // ---- step A2: chunk 2j + 1 (slot 1); the next super-chunk's first W1 piece goes to slot 0
    if (j < 3) FF_STAMP(3 + 4 * j);
    xa_wait_vmcnt<5>();
    __builtin_amdgcn_s_barrier();
    if (j < 3) FF_STAMP(4 + 4 * j);
    proj_chunk(1, 2 * j + 1, 1, [&](int i) __attribute__((always_inline)) { if (more) issue_w1(0, 2 * j + 2, i); });
    // ---- step C: out^T += W2[own 160 columns, 64 j ..] h^T (slot 2); the next super-chunk's second W1 piece goes to slot 1
    if (more) xa_wait_vmcnt<5>(); else xa_wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's halves of h are in the hand-off area
    __builtin_amdgcn_s_barrier();
    if (j < 3) FF_STAMP(5 + 4 * j);
    {
      const char* Os = smem + 2 * FF_SLOT;
      vec8<T> hb[2];                        // [row tile] of the current chunk: k-slots 0-3 = role 0's units, 4-7 = role 1's (W2's column order)
      auto rd_h = [&](int cb) __attribute__((always_inline)) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) hb[rt] = *reinterpret_cast<const vec8<T>*>(xch + ((cb * 2 + rt) * 64 + lane) * 16);
      };
      rd_h(0);
      vec8<T> fa[FF_DEPTH + 1][2];
      // group g: k-step (chunk) g / 5, output tiles 2 (g % 5), + 1 of this wave's 10
      auto rd = [&](int g, vec8<T> (&f)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 2; ++q) f[q] = frag(Os, (role * NTW + 2 * (g % 5) + q) * 16 + fr, 4 * (g / 5) + fq);
      };
#pragma unroll
      for (int d = 0; d < FF_DEPTH; ++d) rd(d, fa[d]);
#pragma unroll
      for (int g = 0; g < 10; ++g) {
        if (g + FF_DEPTH < 10) rd(g + FF_DEPTH, fa[(g + FF_DEPTH) % (FF_DEPTH + 1)]);
        FF_FENCE();
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int rt = 0; rt < 2; ++rt)
            acc[2 * (g % 5) + q][rt] = lr_mfma16(fa[g % (FF_DEPTH + 1)][q], hb[rt], acc[2 * (g % 5) + q][rt]);
        if (more && g < KL) issue_w1(1, 2 * j + 3, g);
        if (g == 4) rd_h(1);
        FF_FENCE();
      }
    }
  }
  if constexpr (POST) {
    // ================= post: x3 = acc + b2 + x (fp16), out^T = Wp x3^T ==============================================
    const __amdgpu_buffer_rsrc_t rsW = uniform_rsrc(P.post_w, (size_t)C * C * 2);
    auto issue_wp = [&](int slot, int pc, int i) __attribute__((always_inline)) {   // piece pc of Wp ([5][320][64]); rows 64 i + 8 w ..
      const unsigned v0 = (unsigned)(((pc * C + lrow) * 64 + lchunk * 8) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lptr_t)(smem + slot * FF_SLOT + (i * 64 + w * 8) * 128), 16, v0, i * 64 * 128, 0, 0);
    };
    // the residual x of this wave's tiles in accumulator layout (its registers were the B operands of the projection until here)
    vec4<T> xd[NTW][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const T* xr = reinterpret_cast<const T*>(P.x) + (size_t)(m_p0 + 16 * rt + fr) * C + role * (C / 2) + 4 * fq;
#pragma unroll
      for (int j = 0; j < NTW; ++j) xd[j][rt] = *reinterpret_cast<const vec4<T>*>(xr + 16 * j);
    }
    __syncthreads();                          // every wave is done with the ring
#pragma unroll
    for (int i = 0; i < KL; ++i) issue_wp(0, 0, i);
    // own half of x3 as B operands: k-step q of this wave = its tiles (2 q, 2 q + 1) = global k-step 5 role + q
    char* xo = smem + FF_SLOT + w * 10240;
    const float* pb = par + H2 + role * (C / 2);
    vec8<T> ball[2 * (NTW / 2)][2];           // all ten k-steps of the pair's rows (this wave's five, the partner's five)
#pragma unroll
    for (int q = 0; q < NTW / 2; ++q)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        f32x4 v0 = acc[2 * q][rt] + *reinterpret_cast<const f32x4*>(pb + (2 * q) * 16 + 4 * fq);
        f32x4 v1 = acc[2 * q + 1][rt] + *reinterpret_cast<const f32x4*>(pb + (2 * q + 1) * 16 + 4 * fq);
#pragma unroll
        for (int r = 0; r < 4; ++r) {         // the unfused kernels round ff's output, then x + that
          v0[r] = (float)(T)v0[r] + (float)xd[2 * q][rt][r];
          v1[r] = (float)(T)v1[r] + (float)xd[2 * q + 1][rt][r];
        }
        const vec8<T> bq_ = xa_pack<T>(v0, v1);
        *reinterpret_cast<vec8<T>*>(xo + ((q * 2 + rt) * 64 + lane) * 16) = bq_;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int p = 0; p < 2 * (NTW / 2); ++p)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const int ow = (w & ~1) | (p / (NTW / 2));          // the wave of the pair that owns global k-step p
        ball[p][rt] = *reinterpret_cast<const vec8<T>*>(smem + FF_SLOT + ow * 10240 + (((p % (NTW / 2)) * 2 + rt) * 64 + lane) * 16);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();              // everyone has its operands: slots 1, 2 are free again
#pragma unroll
    for (int i = 0; i < KL; ++i) issue_wp(1, 1, i);
#pragma unroll
    for (int i = 0; i < KL; ++i) issue_wp(2, 2, i);
#pragma unroll
    for (int j = 0; j < NTW; ++j) { acc[j][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[j][1] = acc[j][0]; }
#pragma unroll
    for (int pc = 0; pc < KL; ++pc) {
      if (pc == 0) xa_wait_vmcnt<10>(); else if (pc + 1 < KL) xa_wait_vmcnt<5>(); else xa_wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      const char* Os = smem + (pc % 3) * FF_SLOT;
      vec8<T> fa[2][2];
      auto rd = [&](int g, vec8<T> (&f)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 2; ++q) f[q] = frag(Os, (role * NTW + 2 * (g % 5) + q) * 16 + fr, 4 * (g / 5) + fq);
      };
      rd(0, fa[0]);
#pragma unroll
      for (int g = 0; g < 10; ++g) {
        if (g + 1 < 10) rd(g + 1, fa[(g + 1) & 1]);
        FF_FENCE();
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int rt = 0; rt < 2; ++rt)
            acc[2 * (g % 5) + q][rt] = lr_mfma16(fa[g & 1][q], ball[2 * pc + g / 5][rt], acc[2 * (g % 5) + q][rt]);
        if (pc >= 1 && pc + 2 < KL && g < KL) issue_wp((pc + 2) % 3, pc + 2, g);
        FF_FENCE();
      }
    }
  }
#undef FF_FENCE

  // ---- epilogue: (acc + bias) -> fp16 -> this wave's LDS area [32 rows][160 columns] -> 16-byte pieces: + residual, store, statistics
  FF_STAMP(14);
  __syncthreads();
  char* stg = smem + w * (32 * FF_PITCH);
  const float* pb2 = par + H2 + (POST ? C : 0) + role * (C / 2);
#pragma unroll
  for (int j = 0; j < NTW; ++j)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const f32x4 v = acc[j][rt] + *reinterpret_cast<const f32x4*>(pb2 + j * 16 + 4 * fq);
      vec4<T> hv;
#pragma unroll
      for (int r = 0; r < 4; ++r) hv[r] = (T)v[r];
      *reinterpret_cast<vec4<T>*>(stg + (16 * rt + fr) * FF_PITCH + (j * 16 + 4 * fq) * 2) = hv;
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int sub = lane & 3;
  constexpr int NP = C / 64;                  // 16-byte pieces per lane and row (4 lanes per row, 20 pieces of the wave's 160 columns)
#pragma unroll
  for (int rh = 0; rh < 2; ++rh) {
    const int row = 16 * rh + (lane >> 2);
    const T* xr = reinterpret_cast<const T*>(POST ? P.post_resid : P.x) + (size_t)(m_p0 + row) * C + role * (C / 2);
    T* orow = reinterpret_cast<T*>(P.out) + (size_t)(m_p0 + row) * C + role * (C / 2);
    float s1 = 0.f, s2 = 0.f;
    uint4 rx[NP];
#pragma unroll
    for (int it = 0; it < NP; ++it) rx[it] = *reinterpret_cast<const uint4*>(xr + (sub + 4 * it) * 8);
#pragma unroll
    for (int it = 0; it < NP; ++it) {
      const int piece = sub + 4 * it;
      float a[8], e[8];
      lr_unpack8<T>(*reinterpret_cast<const uint4*>(stg + row * FF_PITCH + piece * 16), a);
      lr_unpack8<T>(rx[it], e);
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] += e[i];
      const uint4 pk = lr_pack8<T>(a);
      *reinterpret_cast<uint4*>(orow + piece * 8) = pk;
      if constexpr (POST) { if (P.gs_out || P.gp_out) *reinterpret_cast<uint4*>(stg + row * FF_PITCH + piece * 16) = pk; }      // final values for the column sums
      lr_unpack8<T>(pk, a);
#pragma unroll
      for (int i = 0; i < 8; ++i) { s1 += a[i]; s2 = fmaf(a[i], a[i], s2); }
    }
    if (P.st_out) {      // one partial per wave of the pair: [M][2][2]
      s1 += __shfl_xor(s1, 1, 64); s2 += __shfl_xor(s2, 1, 64);
      s1 += __shfl_xor(s1, 2, 64); s2 += __shfl_xor(s2, 2, 64);
      if (sub == 0) {
        float2 o; o.x = s1; o.y = s2;
        *reinterpret_cast<float2*>(P.st_out + ((size_t)(m_p0 + row) * 2 + role) * 2) = o;
      }
    }
  }
  if constexpr (POST) {
    if (P.gs_out || P.gp_out) {      // per-channel (sum, sumsq) over the block's 128 rows, fixed order: wave -> its 32 rows, then the four pairs
      float* cs = reinterpret_cast<float*>(smem + 8 * 32 * FF_PITCH);      // [8 waves][160][2]
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      for (int c = lane; c < C / 2; c += 64) {
        float a1 = 0.f, a2 = 0.f;
#pragma unroll 8
        for (int r = 0; r < 32; ++r) {
          const float v = (float)*reinterpret_cast<const T*>(stg + r * FF_PITCH + c * 2);
          a1 += v; a2 = fmaf(v, v, a2);
        }
        cs[(w * (C / 2) + c) * 2] = a1;
        cs[(w * (C / 2) + c) * 2 + 1] = a2;
      }
      __syncthreads();
      float a1 = 0.f, a2 = 0.f;
      if (t < C) {
        const int rl = t / (C / 2), cc = t % (C / 2);
#pragma unroll
        for (int p4 = 0; p4 < 4; ++p4) { a1 += cs[((2 * p4 + rl) * (C / 2) + cc) * 2]; a2 += cs[((2 * p4 + rl) * (C / 2) + cc) * 2 + 1]; }
        float2 o; o.x = a1; o.y = a2;
        if (P.gs_out) *reinterpret_cast<float2*>(P.gs_out + ((size_t)bid * C + t) * 2) = o;
      }
      if (P.gp_out) {      // the 32 groups of 10 channels (block-uniform branch)
        __syncthreads();
        if (t < C) { cs[t * 2] = a1; cs[t * 2 + 1] = a2; }
        __syncthreads();
        if (t < 32) {
          float g1 = 0.f, g2 = 0.f;
#pragma unroll
          for (int c = 0; c < C / 32; ++c) { g1 += cs[(t * (C / 32) + c) * 2]; g2 += cs[(t * (C / 32) + c) * 2 + 1]; }
          const int m0 = bid * FF_ROWS;
          const int smp = m0 / P.gp_hw, chunk = (m0 - smp * P.gp_hw) / FF_ROWS;
          float2 o; o.x = g1; o.y = g2;
          *reinterpret_cast<float2*>(P.gp_out + (((size_t)smp * P.gp_chunks + chunk) * 32 + t) * 2) = o;
        }
      }
    }
  }
#ifdef LR_FFN_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  FF_STAMP(15);
#endif
#endif
}

template <typename T>
static int ffn_block_t(const lr_ffn_args* a, lr_stream_t s) {
  if (!a || !a->x || !a->w1 || !a->b1 || !a->w2 || !a->b2 || !a->out) return LR_E_ARG;
  if (a->M <= 0 || a->H <= 0) return LR_E_ARG;
  if (a->C != FF_C || a->H % 64 || a->H > FF_MAX_H || a->M % FF_ROWS) return LR_E_UNSUPPORTED;
  if (((uintptr_t)a->x | (uintptr_t)a->w1 | (uintptr_t)a->b1 | (uintptr_t)a->w2 | (uintptr_t)a->b2 | (uintptr_t)a->out) & 15) return LR_E_ALIGN;
  if (a->stats_out && ((uintptr_t)a->stats_out & 7)) return LR_E_ALIGN;
  FfnParams P;
  P.x = a->x; P.w1 = a->w1; P.b1 = a->b1; P.w2 = a->w2; P.b2 = a->b2; P.out = a->out; P.st_out = a->stats_out;
  P.M = a->M; P.H = a->H; P.nblocks = a->M / FF_ROWS; P.eps = a->ln_eps;
#ifdef LR_FFN_TRACE
  P.trace = g_ff_trace;
#endif
  const size_t smem = 3 * FF_SLOT + FF_PAR_BYTES + 4 * 4096;      // ring | bias rows | hand-off areas of the four wave pairs
  const bool post = a->post_w != nullptr;
  if (post) {
    if (!a->post_b || !a->post_resid) return LR_E_ARG;
    if (((uintptr_t)a->post_w | (uintptr_t)a->post_b | (uintptr_t)a->post_resid | (uintptr_t)a->gn_stats_out) & 15) return LR_E_ALIGN;
  } else if (a->gn_stats_out || a->gn_group_out) return LR_E_ARG;
  P.post_w = a->post_w; P.post_b = a->post_b; P.post_resid = a->post_resid; P.gs_out = a->gn_stats_out;
  P.gp_out = a->gn_group_out; P.gp_hw = 1; P.gp_chunks = 0;
  if (P.gp_out) {
    if (a->gn_hw <= 0 || a->gn_hw % FF_ROWS || a->M % a->gn_hw || ((uintptr_t)P.gp_out & 7)) return LR_E_ARG;
    P.gp_hw = a->gn_hw; P.gp_chunks = a->gn_hw / FF_ROWS;
  }
  static unsigned long long attr_done[2] = {0, 0};
  if (lr_attr_needed(&attr_done[post])) {
    hipFuncSetAttribute(post ? reinterpret_cast<const void*>(ffn_block_kernel<T, true>) : reinterpret_cast<const void*>(ffn_block_kernel<T, false>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  if (post) hipLaunchKernelGGL((ffn_block_kernel<T, true>), dim3(P.nblocks), dim3(FF_THREADS), smem, (hipStream_t)s, P);
  else hipLaunchKernelGGL((ffn_block_kernel<T, false>), dim3(P.nblocks), dim3(FF_THREADS), smem, (hipStream_t)s, P);
  return lr_launch_status();
}

extern "C" int lr_ffn_block_f16(const lr_ffn_args* a, lr_stream_t s) { return ffn_block_t<f16>(a, s); }
extern "C" int lr_ffn_block_bf16(const lr_ffn_args* a, lr_stream_t s) { return ffn_block_t<bf16>(a, s); }
