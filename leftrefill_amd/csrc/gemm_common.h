// Shared pieces of the implicit-GEMM kernels (gemm_conv.hip, conv_halo.hip): kernel parameters, the register epilogue,
// GroupNorm / LayerNorm statistics helpers.  Device code only; included once per translation unit.
#pragma once
#include "common.h"
#include <type_traits>

#define BK 64
#define GEMM_THREADS 256
#define BM 128

struct GemmParams {
  const f16* p1; const f16* p2; const f16* wt; const float* bias; const f16* rowvec; const f16* resid; f16* out;
  int C1, C2, H, W, Hs, Ws, taps, stride, up, pad, zins, N, M, K, ld_rowvec, ld_resid, ld_out, geglu, gelu, rows_per_batch;
  int ntiles_n, nblocks;
  int splits; float* ws;   // split-K: blockIdx.y = K slice, fp32 partial tiles -> ws[split][M][N]
  int ntiles_m, m_fastest; // tile order inside an XCD's contiguous chunk (see tile_order())
  unsigned hw_mul, hw_sh, w_mul, w_sh;   // magic numbers of the unsigned divisions by H*W and by W (see lr_udiv)
  // LayerNorm folded into this GEMM (A = raw x, Wt = W * gamma): out = rstd[m] * (acc - mean[m] * ln_cs[n]) + bias'[n];
  // (mean, rstd) of row m come from the producer's per-row partial (sum, sumsq): ln_part[m][ln_parts][2]
  const float* ln_part; const float* ln_cs; int ln_parts; float ln_eps, ln_invc;
  // per-row (sum, sumsq) of THIS GEMM's fp16 output over each wave's column range: st_out[m][st_parts][2]
  float* st_out; int st_parts;
  // per-channel (sum, sumsq) of this GEMM's fp16 output over each wave's rows: gs_out[row block][N][2], row block =
  // m / (rows per wave tile); feeds the GroupNorm of the consumer (lr_groupnorm_finalize) instead of a statistics pass
  float* gs_out;
  // per-GROUP (sum, sumsq) of this GEMM's fp16 output over each tile's rows, for the GroupNorm(32) of a single-source consumer:
  // gp_out[sample][gp_chunks][32][2], chunk = row tile inside the sample (gp_hw rows per sample, a multiple of the tile's rows),
  // gp_cg = N / 32 channels per group (divides the tile width).  lr_groupnorm_apply_n reads it directly: no finalize launch.
  float* gp_out; int gp_cg, gp_chunks, gp_hw;
  int gs_store;   // split-K reduce: write gs_out (0 when gs_out only carries the "statistics wanted" flag for gp_out)
  // per-sample weights / bias (GroupNorm folded into a pointwise GEMM, lr_gn_fold_weights_f16): sample = m / rows_per_batch
  int wt_bstride, bias_bstride;
  // weight layout: 0 = [N][K] (a K-step's 128-byte pieces of consecutive rows lie 2 K bytes apart), 1 = piece-major [K / 64][N][64]
  // (the 128-byte pieces of a K-step are consecutive: a tile's weight slice of one K-step is ONE contiguous run of tile_n * 128
  // bytes -- DRAM pages and L2 channels see a stream instead of a 2K-byte stride)
  int wt_pm;
  // c16: 3x3 conv over a 16-channel source (the UNet's 9-channel input padded to 16: 32 bytes per pixel).  A K-step of 64 covers FOUR
  // taps (k = tap * 16 + c, 9 taps -> 144, zero-padded to K = 192 = 3 K-steps): the 16-byte chunk c of a gathered row comes from tap
  // 4 kt + (c >> 1), half c & 1, so the per-lane gather offset changes every K-step.  Replaces the 64-channel padding of round 1-3
  // (K = 576, 86 % zero work).
  int c16;
  // K extension by a pointwise term over a second (virtually concatenated) pair of sources with the output's resolution:
  //   out = conv3x3([p1 | p2]) + W_s [p3 | p4],   wt = [N][taps (C1 + C2) + C3 + C4]  (the 3x3 part first)
  // -- the ResBlock's skip_connection (a 1x1 conv of the block input) accumulated into the block's last conv instead of running as its
  // own GEMM whose output travels through HBM to this conv's residual epilogue.  The extra K-steps gather tap (0, 0) of p3 / p4.
  // Stride 1, no upsample; the main part may itself be pointwise (taps == 1): out = W_a [p1 | p2] + W_s [p3 | p4].
  const f16* p3; const f16* p4; int C3, C4;
  int bf16;   // 16-bit type of activations / weights / outputs: 0 = fp16, 1 = bf16
  // in-launch split-K reduce (lr_gemm_args.splitk_mode == 1): per-tile arrival counters (zero on entry, left zero); NULL = the partials
  // are reduced by splitk_reduce_kernel in a second launch
  unsigned* sk_cnt;
#ifdef LR_GEMM_STAGGER
  int stagger;   // developer build only: shader-clock cycles the SECOND co-resident block of a CU waits before it starts (env LR_GEMM_STAGGER)
#endif
#ifdef LR_GEMM_TRACE
  unsigned long long* trace;   // developer build only: per-block shader-clock stamps [block][8] (tools/trace_gemm.py)
#endif
};

#ifdef LR_GEMM_TRACE
#define LR_STAMP(k) do { if (P.trace && threadIdx.x == 0) P.trace[(size_t)(blockIdx.y * P.nblocks + lr_trace_tile) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define LR_STAMP(k) do { } while (0)
#endif

// Blocks are handed to XCDs in contiguous logical chunks (bijective remap of blockIdx).  Inside a chunk the order is
//   grouped (m_fastest = G > 1): panels of G row tiles, m-fastest inside a panel (see launch_pipe_t);
//   m-fastest: neighbours share the WEIGHT slice (BN x K) -- right when that slice is MBs (3x3 convs at 1280 channels:
//              3.7 MB per N-tile, 29 MB in total, far beyond one XCD's 4 MB L2) ;
//   n-fastest: neighbours share the activation rows -- right when the weights are small and fit L2 anyway.
__device__ __forceinline__ void tile_order(const GemmParams& P, int bid, int& tile_m, int& tile_n) {
  if (P.m_fastest > 1) {      // grouped: panels of m_fastest consecutive row tiles, inside a panel m-fastest over all column tiles -- the ~32 tiles
    // an XCD runs at once then share m_fastest x (32 / m_fastest) row / column slices instead of 2 x 16 (both operands fit its L2)
    const int G = P.m_fastest, per = G * P.ntiles_n;
    const int grp = bid / per, first = grp * G;
    const int gsz = min(G, P.ntiles_m - first), r = bid - grp * per;
    tile_m = first + r % gsz; tile_n = r / gsz;
  } else if (P.m_fastest) { tile_m = bid % P.ntiles_m; tile_n = bid / P.ntiles_m; }
  else { tile_n = bid % P.ntiles_n; tile_m = bid / P.ntiles_n; }
}

typedef __attribute__((address_space(3))) void* lptr_t;


// =====================================================================================================================
// Register epilogue shared by both kernels: no LDS staging, no barriers, every wave drains its own accumulators.
//
// With the swapped product a lane (fr = lane & 15, fq = lane >> 4) holds D[n = 16 j + 4 fq + r][m = 16 i + fr], r < 4:
// four consecutive channels of one output row per MFMA tile.  `v_permlane16_swap` of two tiles (A, B) exchanges the odd
// 16-lane rows of A with the even rows of B, after which the lane owns EIGHT consecutive channels (16 bytes of fp16) of
// tile (fq & 1 ? B : A): residual / per-sample row vector come in as one 16-byte load, the result leaves as one 16-byte
// store, and a wave-wide store covers 16 rows x 64 contiguous bytes (tile pairs adjacent in n); the other half of each 128-byte line
// comes from a later unit (with plain stores ~20 % of a level-0 conv's bytes reached memory twice, see LR_OUT_AUX below).  An odd leftover tile column is paired along m instead.  Loads of the next unit are
// independent of the current one, so the residual latency overlaps across units and across the block's waves.
//   optional: LayerNorm fold (see GemmParams), per-row (sum, sumsq) of the rounded output for the NEXT LayerNorm.
// =====================================================================================================================
__device__ __forceinline__ void swap_rows16(f32x4& a, f32x4& b) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float ar = a[r], br = b[r];   // plain floats: __builtin_bit_cast of a vector-element lvalue reads element 0
    const auto s = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, ar), __builtin_bit_cast(unsigned, br),
                                                    false, false);
    a[r] = __builtin_bit_cast(float, (unsigned)s[0]);
    b[r] = __builtin_bit_cast(float, (unsigned)s[1]);
  }
#endif
}

// rs : LDS [rows of the block][2] = (mean, rstd), already offset to this wave's first row (LayerNorm fold only)
// par: LDS [2][PAR_LD] floats = bias and ln_colsum of the block's columns (zeros where absent), already offset to this
//      wave's first column; filled by LDS-DMA in the kernel prologue (stage_params) so that the epilogue issues NO
//      per-lane parameter loads -- a VGPR load in the unit loop makes hipcc wait vmcnt(0), which on gfx950 also waits
//      for every store already issued (one full store round trip per unit: 65k cycles per 256x320 tile, measured).
//
// Units: one unit = one swapped tile pair = one 16-byte store per lane.  TE = emitted tile columns per wave (TN, or TN/2
// with GEGLU where tiles (2 jo, 2 jo + 1) = (value, gate) of output tile jo).  Units 0 .. TM*(TE/2)-1 pair columns
// (2 jp, 2 jp + 1) of row tile i; an odd last column is paired along m: rows (2 ip, 2 ip + 1).
// Residual / row-vector come through buffer descriptors (absent operand = 0 records = zeros, row / column tails =
// out-of-range offsets: no branches around memory ops, so the compiler's counted vmcnt stays exact); the loads run G
// units ahead of their use and the stores never block.
//
// MODE (compile time, keeps the unrolled epilogue small: the erf polynomial is only instantiated where it is used):
//   0 plain | 1 GEGLU (value * gelu(gate)) | 2 erf-GELU of (acc + bias [+ rowvec]) before the residual
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// Cache policy of the epilogue's output stores: sc1 (write-through).  MEASURED, whole UNet step, same box, alternating runs of builds that
// differ in this constant only: sc1 -0.07 ... -0.21 ms in six pairs (mean -0.12); nt (2) +0.1 ms; nt + sc1 (18) +0.25 ms; training step
// -0.07 / -0.10 ms; multi-view step unchanged within its run-to-run spread.  The same policy tried on the other large writers, one site at a
// time: attention outputs +0.04 ms, gn_apply / fused blocks / split-K reduce outputs +-0.02, split-K partials +0.44 ms -- so only these
// stores carry it.
// What it changes, MEASURED (rocprofv3 --pmc WRITE_SIZE per dispatch, tools/pmc_write_probe.py; bench.py's roofline.traffic): with plain stores
// a level-0 conv_halo<320> launch moves 43.5-54.5 MB (mean 51.5) out of the L2s for 41.9 MB of output + 1.4 MB of statistics; written
// through it moves 43.3 MB, every launch.  GEMM family over the step (one box, tools/pmc_step.py after its selection fix): WRITE_SIZE 46.3 -> 40.8 MB per call (-0.76 GB per UNet step), FETCH
// x2 127.1 -> 123.9 MB.  So a level-0 conv's plain stores wrote ~20 % of its bytes twice (12 % over the family, whose split-K calls write
// partials either way); why (half-filled 128-byte lines evicted and written again is a guess) was not measured.  The hypothesis this experiment STARTED from -- output bytes left dirty and written back at the kernel boundary (the
// guide's "boundary" row) -- made the opposite prediction (a plain-store conv showing LESS than its output in its own counter window, the
// rest on its successor) and is not supported: own bytes >= output in both builds, the following gn_apply shows its own 41.9 MB in both.
#ifndef LR_OUT_AUX
#define LR_OUT_AUX 16
#endif

// Which 16-column tile of the block's EMITTED column space wave `wn` accumulates in its je-th tile column (TE per wave,
// WNW waves along n).  Contiguous ranges, except TE = 5 (80-column wave tiles of the 160 / 320-wide blocks): 80 columns
// are 160 bytes of fp16, so contiguous ranges would start in the middle of 128-byte lines and every 64-byte store
// segment of waves 1 and 3 would straddle two lines.  Instead wave wn takes tiles 4 wn .. 4 wn + 3 (one whole line)
// plus tile 4 WNW + wn of the last line: the paired stores are 64-byte aligned halves of a line owned by one wave.
// The permutation only changes which B_s rows a wave reads (weights are staged in natural row order).
template <int TE, int WNW>
__device__ __forceinline__ constexpr int emit_tile(const int wn, const int je) {
  return TE == 5 ? (je < 4 ? 4 * wn + je : 4 * WNW + wn) : wn * TE + je;
}
// accumulator tile column j of wave wn -> 16-row tile of the staged weight rows
template <int TN, int WNW, bool GEGLU>
__device__ __forceinline__ constexpr int weight_tile(const int wn, const int j) {
  return GEGLU ? 2 * emit_tile<TN / 2, WNW>(wn, j >> 1) + (j & 1) : emit_tile<TN, WNW>(wn, j);
}

// HALO (conv_halo.hip): the wave's TM row tiles are 16-pixel segments of TM consecutive image lines -- row tile i starts at
// m_w0 + i * P.W instead of m_w0 + 16 i -- and the row-block index of gs_out is passed in (rb_halo: wave tiles numbered tile-major).
template <int TM, int TN, int MODE, int PAR_LD, int WNW, typename T, bool HALO = false>
__device__ __forceinline__ void epilogue_units(const GemmParams& P, f32x4 (&acc)[TN][TM], const int m_w0, const int n0,
                                               const int wn, const int lane, const float* rs, const float* par,
                                               const int part, float* gsl = nullptr, const int rb_halo = 0) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr bool GEGLU = MODE == 1;
  constexpr int TE = GEGLU ? TN / 2 : TN;
  constexpr int NPJ = TE / 2;
  constexpr int NUJ = TM * NPJ;
  constexpr int NU = NUJ + (TE & 1) * (TM / 2);
  // Addend loads in flight per wave (16 B per lane each).  The ring carries ONE operand per unit -- the residual when there
  // is one, else the per-sample row vector (no layer of the model has both; if a caller passes both, the row vector is
  // loaded synchronously) -- so 8 units fit in the registers 4 two-operand units took: the 40-tile epilogue was bound by
  // bytes in flight (32 KiB per CU at ~2.5 us loaded latency = 3.3 TB/s chip-wide), not by bandwidth.
  constexpr int G = TM * TN >= 40 ? 8 : (NU < 16 ? NU : 16);
  const int fr = lane & 15, fq = lane >> 4;
  const int odd = fq & 1, ch8 = (fq >> 1) * 8;
  const int N_out = GEGLU ? P.N >> 1 : P.N;
  const int no0 = GEGLU ? n0 >> 1 : n0;      // block's first emitted column
  const bool ln = P.ln_part != nullptr;
  const bool fin = P.splits == 1;
  const unsigned OOB = 0x80000000u;
  const int mstep = HALO ? P.W : 16;      // distance in m between the wave's consecutive 16-row tiles
  const __amdgpu_buffer_rsrc_t rsR = uniform_rsrc(P.resid ? (const void*)P.resid : (const void*)P.out,
                                                  (P.resid && fin) ? ((size_t)(P.M - 1) * P.ld_resid + N_out) * 2 : 0);
  const __amdgpu_buffer_rsrc_t rsV = uniform_rsrc(P.rowvec ? (const void*)P.rowvec : (const void*)P.out,
                                                  (P.rowvec && fin) ? ((size_t)((P.M - 1) / P.rows_per_batch) * P.ld_rowvec + N_out) * 2 : 0);
  const __amdgpu_buffer_rsrc_t rsO = uniform_rsrc(P.out, fin ? ((size_t)(P.M - 1) * P.ld_out + N_out) * 2 : 0);
  const __amdgpu_buffer_rsrc_t rsWS = uniform_rsrc(P.sk_cnt ? (const void*)P.ws : (const void*)P.out,
                                                   P.sk_cnt ? (size_t)P.splits * P.M * P.N * 4 : 0);
  // one register array serves both kinds of output statistics (a GEMM feeds a LayerNorm or a GroupNorm, never both):
  // row mode (P.st_out): s1[i] = sg[i], s2[i] = sg[SGH + i];  channel mode (P.gs_out): g1[q] = sg[q], g2[q] = sg[SGH + q]
  constexpr int SGH = TM > 8 ? TM : 8;
  float sg[2 * SGH];
#pragma unroll
  for (int i = 0; i < 2 * SGH; ++i) sg[i] = 0.f;
#define s1(i) sg[(i)]
#define s2(i) sg[SGH + (i)]
#define g1(q) sg[(q)]
#define g2(q) sg[SGH + (q)]
  // GroupNorm statistics of the consumer: per-channel sums over the wave's rows.  All units of a column group (the TM
  // units of a tile-column pair, or the TM/2 units of the odd last column) put the SAME eight channels in a lane, so the
  // lane accumulates over them and the group is reduced over the 16 row lanes (4 DPP adds per value) when it completes.
  // gsl: LDS [block columns][2] of this wave's row block -- the per-channel sums also go there when the block reduces them to
  // per-group sums afterwards (gn_group_reduce)
  const bool gstat = !GEGLU && fin && (P.gs_out != nullptr || P.gp_out != nullptr) && P.st_out == nullptr;
  auto row16_sum = [&](float v) -> float {   // sum over the 16 lanes of a DPP row, result in every lane
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    return v;
  };
  auto flush_group = [&](const int n_lane, const bool pair_fq) {
    const int rb = HALO ? rb_halo : m_w0 / (TM * 16);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float a = row16_sum(g1(q)), b = row16_sum(g2(q));
      if (pair_fq) {   // odd last column: lane rows (fq, fq ^ 1) hold the same channels for different output rows
        const auto sa = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, a), false, false);
        a = __builtin_bit_cast(float, (unsigned)sa[0]) + __builtin_bit_cast(float, (unsigned)sa[1]);
        const auto sb = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, b), false, false);
        b = __builtin_bit_cast(float, (unsigned)sb[0]) + __builtin_bit_cast(float, (unsigned)sb[1]);
      }
      g1(q) = a; g2(q) = b;
    }
    if (fr == 0 && (!pair_fq || !odd) && n_lane < N_out) {
      if (P.gs_out != nullptr && m_w0 < P.M) {
        float* dst = P.gs_out + ((size_t)rb * N_out + n_lane) * 2;
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
          const f32x4 o = {g1(q), g2(q), g1(q + 1), g2(q + 1)};
          *reinterpret_cast<f32x4*>(dst + 2 * q) = o;
        }
      }
      if (P.gp_out != nullptr) {      // rows past M contributed zeros
        float* dst = gsl + (n_lane - no0) * 2;
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
          const f32x4 o = {g1(q), g2(q), g1(q + 1), g2(q + 1)};
          *reinterpret_cast<f32x4*>(dst + 2 * q) = o;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) { g1(q) = 0.f; g2(q) = 0.f; }
  };
  auto rowstat = [&](const int i) -> float2 { return *reinterpret_cast<const float2*>(rs + 2 * (i * 16 + fr)); };
  auto par4 = [&](const int which, const int col) -> f32x4 { return *reinterpret_cast<const f32x4*>(par + which * PAR_LD + col); };
  // accumulator tile (i, je) of the EMITTED grid, with LayerNorm fold, bias and the GEGLU gate applied (pre-swap layout:
  // lane holds channels 4 fq .. 4 fq + 3 of the tile)
  auto tile = [&](const int i, const int je) -> f32x4 {
    if constexpr (!GEGLU) {
      const int c = weight_tile<TN, WNW, false>(wn, je) * 16 + fq * 4;
      f32x4 r = acc[je][i];
      if (ln) { const float2 mr = rowstat(i); r = (r - par4(1, c) * mr.x) * mr.y; }
      if (fin) r += par4(0, c);
      return r;
    } else {
      const int c = weight_tile<TN, WNW, true>(wn, 2 * je) * 16 + fq * 4;
      f32x4 u = acc[2 * je][i], g = acc[2 * je + 1][i];
      if (ln) {
        const float2 mr = rowstat(i);
        u = (u - par4(1, c) * mr.x) * mr.y;
        g = (g - par4(1, c + 16) * mr.x) * mr.y;
      }
      u += par4(0, c); g += par4(0, c + 16);
      const f32x2_t g01 = lr_gelu_erf2((f32x2_t){g[0], g[1]}), g23 = lr_gelu_erf2((f32x2_t){g[2], g[3]});
      const f32x4 o = {u[0] * g01[0], u[1] * g01[1], u[2] * g23[0], u[3] * g23[1]};
      return o;
    }
  };
  // lane's output coordinates for unit u
  auto coords = [&](const int u, int& m, int& n) {
    if (u < NUJ) {
      const int jp = u / TM, i = u - jp * TM;
      m = m_w0 + i * mstep + fr;
      n = no0 + emit_tile<TE, WNW>(wn, 2 * jp + odd) * 16 + ch8;
    } else {
      const int ip = u - NUJ;
      m = m_w0 + (2 * ip + odd) * mstep + fr;
      n = no0 + emit_tile<TE, WNW>(wn, TE - 1) * 16 + ch8;
    }
  };
  // sample index of row m for the per-sample row vector: one division per wave, then boundary compares
  const int b_w0 = P.rowvec ? m_w0 / P.rows_per_batch : 0;
  const bool has_res = P.resid != nullptr && fin;
  const bool both = has_res && P.rowvec != nullptr;
  auto rowvec_offset = [&](const int m, const int n, const bool ok) -> unsigned {
    int b = b_w0;
    for (int lim = (b_w0 + 1) * P.rows_per_batch; m >= lim; lim += P.rows_per_batch) ++b;
    return ok ? (unsigned)(((size_t)b * P.ld_rowvec + n) * 2) : OOB;
  };
  const __amdgpu_buffer_rsrc_t rsX = has_res ? rsR : rsV;     // the operand the ring carries
  auto fetch = [&](const int u, u32x4& r) {
    int m, n;
    coords(u, m, n);
    const bool ok = m < P.M && n < N_out;
    const unsigned off = has_res ? (ok ? (unsigned)(((size_t)m * P.ld_resid + n) * 2) : OOB)
                                 : (P.rowvec ? rowvec_offset(m, n, ok) : OOB);
    r = __builtin_amdgcn_raw_buffer_load_b128(rsX, off, 0, 0);
  };
  auto finish = [&](const int u, const u32x4& rx) {
    f32x4 a, b;
    int ia, ib;          // statistics slots of the even / odd lane rows
    if (u < NUJ) {
      const int jp = u / TM, i = u - jp * TM;
      a = tile(i, 2 * jp); b = tile(i, 2 * jp + 1);
      ia = i; ib = i;
    } else {
      const int ip = u - NUJ;
      a = tile(2 * ip, TE - 1); b = tile(2 * ip + 1, TE - 1);
      ia = 2 * ip; ib = 2 * ip + 1;
    }
    swap_rows16(a, b);
    int m, n;
    coords(u, m, n);
    const bool ok = m < P.M && n < N_out;
    if (!fin) {   // raw fp32 partial; bias / row vector / residual are applied by splitk_reduce_kernel
      if (ok) {
        float* dst = P.ws + ((size_t)blockIdx.y * P.M + m) * P.N + n;
        if (P.sk_cnt) {      // read by OTHER blocks of this launch: write-through (sc1) stores, no release fence needed (guide G16, R1)
          const unsigned off = (unsigned)((((size_t)blockIdx.y * P.M + m) * P.N + n) * 4);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, a), rsWS, off, 0, 16);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, b), rsWS, off + 16, 0, 16);
        } else {
          *reinterpret_cast<f32x4*>(dst) = a;      // (plain stores: the reduce launch finds most of the partials in the L2s; written through: +0.44 ms per step)
          *reinterpret_cast<f32x4*>(dst + 4) = b;
        }
      }
      return;
    }
    float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    float e[8];
    lr_unpack8<T>(__builtin_bit_cast(uint4, rx), e);
    if (both) {                        // row vector next to a residual: not prefetched (no layer of the model does this)
      float e2[8];
      const u32x4 r2 = __builtin_amdgcn_raw_buffer_load_b128(rsV, rowvec_offset(m, n, ok), 0, 0);
      lr_unpack8<T>(__builtin_bit_cast(uint4, r2), e2);
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] += e2[q];
    }
    if constexpr (MODE == 2) {         // erf-GELU sits between the row vector and the residual
      if (!has_res) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] += e[q];
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = lr_gelu_erf(v[q]);
      if (has_res) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] += e[q];
      }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] += e[q];
    }
    const uint4 pk = lr_pack8<T>(v);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pk), rsO,
                                           ok ? (unsigned)(((size_t)m * P.ld_out + n) * 2) : OOB, 0, LR_OUT_AUX);
    if (gstat) {
      float f[8];
      lr_unpack8<T>(pk, f);
#pragma unroll
      for (int q = 0; q < 8; ++q) { const float x = ok ? f[q] : 0.f; g1(q) += x; g2(q) = fmaf(x, x, g2(q)); }
      if (u < NUJ) { if (u % TM == TM - 1) flush_group(n, false); }
      else if (u == NU - 1) flush_group(n, true);
    }
    if (P.st_out) {
      float f[8], t1 = 0.f, t2 = 0.f;
      lr_unpack8<T>(pk, f);
#pragma unroll
      for (int q = 0; q < 8; ++q) { t1 += f[q]; t2 = fmaf(f[q], f[q], t2); }
      if (!ok) { t1 = 0.f; t2 = 0.f; }
      if (ia == ib) { s1(ia) += t1; s2(ia) += t2; }
      else {
        s1(ia) += odd ? 0.f : t1; s2(ia) += odd ? 0.f : t2;
        s1(ib) += odd ? t1 : 0.f; s2(ib) += odd ? t2 : 0.f;
      }
    }
  };

  // software pipeline: the loads of unit u + G are issued right after unit u is finished (a ring of G register sets)
  u32x4 rr[G];
#pragma unroll
  for (int k = 0; k < G; ++k)
    if (k < NU) fetch(k, rr[k]);
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    finish(u, rr[u % G]);
    if (u + G < NU) fetch(u + G, rr[u % G]);
  }
  if (P.st_out) {   // row sums over this wave's column range: the four fq lanes of an fr hold pieces of the same row
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float a = s1(i), b = s2(i);
      a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
      a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
      const int m = m_w0 + i * mstep + fr;
      if (fq == 0 && m < P.M) {
        float2 o; o.x = a; o.y = b;
        *reinterpret_cast<float2*>(P.st_out + ((size_t)m * P.st_parts + part) * 2) = o;
      }
    }
  }
#undef s1
#undef s2
#undef g1
#undef g2
#endif
}

// Per-group sums of a tile from the per-channel sums its waves left in LDS: gsl[row block][BN_][2] -> gp_out.  One thread
// per group of the tile, fixed order (row blocks outer, channels inner).  Called by all threads after a block-wide barrier.
template <int BN_, int NROWBLK, int BM_>
__device__ __forceinline__ void gn_group_reduce(const GemmParams& P, const float* gsl, const int m0, const int n0, const int t) {
  const int cg = P.gp_cg;
  if (t < BN_ / cg && m0 < P.M) {
    const int g = n0 / cg + t;
    if (g < 32) {
      float s = 0.f, q = 0.f;
      for (int rb = 0; rb < NROWBLK; ++rb) {
        const float2* src = reinterpret_cast<const float2*>(gsl) + rb * BN_ + t * cg;
        for (int c = 0; c < cg; ++c) { const float2 v = src[c]; s += v.x; q += v.y; }
      }
      const int smp = m0 / P.gp_hw, chunk = (m0 - smp * P.gp_hw) / BM_;
      float2 o; o.x = s; o.y = q;
      *reinterpret_cast<float2*>(P.gp_out + (((size_t)smp * P.gp_chunks + chunk) * 32 + g) * 2) = o;
    }
  }
}

// Kernel prologue: bias and ln_colsum of the block's BN columns -> LDS par[2][PAR_LD] by 4-byte LDS-DMA (wave w covers
// columns 64 w .. 64 w + 63; a missing operand or a column >= N reads as 0 through the descriptor's bounds check).
// Issued BEFORE the first K stage, so every later counted vmcnt wait covers it.
template <int BN, int PAR_LD>
__device__ __forceinline__ void stage_params(const GemmParams& P, float* par, const int n0, const int w, const int lane,
                                             const int smp = 0) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (w < (BN + 63) / 64) {
    const __amdgpu_buffer_rsrc_t rb = uniform_rsrc(P.bias ? (const void*)(P.bias + (size_t)smp * P.bias_bstride) : (const void*)P.wt,
                                                   (P.bias && P.splits == 1) ? (size_t)P.N * 4 : 0);
    const __amdgpu_buffer_rsrc_t rc = uniform_rsrc(P.ln_cs ? (const void*)P.ln_cs : (const void*)P.wt,
                                                   P.ln_cs ? (size_t)P.N * 4 : 0);
    const unsigned off = (unsigned)(n0 + w * 64 + lane) * 4u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lptr_t)(par + w * 64), 4, off, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rc, (lptr_t)(par + PAR_LD + w * 64), 4, off, 0, 0, 0);
  }
#endif
}

// (mean, rstd) of the block's rows from the producer's partial sums -> LDS rs[row][2]; one thread per row.
__device__ __forceinline__ void ln_rows_to_lds(const GemmParams& P, float* rs, const int m0, const int rows, const int t) {
  if (t < rows) {
    const int m = m0 + t;
    double s = 0.0, q = 0.0;
    if (m < P.M) {
      const float2* p = reinterpret_cast<const float2*>(P.ln_part) + (size_t)m * P.ln_parts;
      for (int k = 0; k < P.ln_parts; ++k) { const float2 v = p[k]; s += (double)v.x; q += (double)v.y; }
    }
    // E[x^2] - mean^2 in fp64: with |mean| >> std (outlier tokens) the fp32 difference cancels to 0 and rstd jumps to 1/sqrt(eps)
    const double mean = s * (double)P.ln_invc;
    const double var = fmax(q * (double)P.ln_invc - mean * mean, 0.0);
    rs[2 * t] = (float)mean;
    rs[2 * t + 1] = (float)(1.0 / sqrt(var + (double)P.ln_eps));
  }
}

// n / d for any 32-bit n with a host-made (mul, sh) pair (Granlund-Montgomery round-up method): 4 VALU ops instead of
// the ~40-instruction expansion of a runtime integer division
__device__ __forceinline__ unsigned lr_udiv(unsigned n, unsigned mul, unsigned sh) {
  const unsigned t = __umulhi(n, mul);
  return (t + ((n - t) >> (sh ? 1 : 0))) >> (sh ? sh - 1 : 0);
}
static inline void lr_udiv_magic(unsigned d, unsigned* mul, unsigned* sh) {
  unsigned l = 0;
  while ((1ull << l) < d) ++l;                       // ceil(log2 d)
  *mul = (unsigned)((((1ull << l) - d) << 32) / d + 1);
  *sh = l;                                           // d == 1: l = 0, mul = 1 -> t = 0, q = n
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// =====================================================================================================================
// In-launch split-K reduce (lr_gemm_args.splitk_mode = 1).  Every K-slice block of an output tile writes its fp32
// partial tile with write-through stores, arrives on the tile's counter, waits (bounded) until all `splits` slices have arrived and
// then reduces ITS SHARE of the tile -- 32-row sub-blocks j = slice, slice + splits, ... -- summing the slices in the fixed order
// 0 .. splits-1 and running the epilogue splitk_reduce_kernel would run (bias, row vector, GELU, residual, 16-bit store, the
// consumer GroupNorm's statistics in the same layouts): deterministic, no second launch, the partials are read back from L2 /
// Infinity Cache.  Protocol per MI355X guide G16: payload sc1 stores, every storing wave drains vmcnt, ONE relaxed agent-scope
// counter per tile, ONE relaxed poll loop (one lane, s_sleep), ONE agent acquire, then plain loads.  The blocks of a tile spin on
// each other, so the host only selects this mode when the whole grid is resident at once (tiles x splits <= CUs, one block per
// CU for these kernels); the spin is bounded all the same (lr_sk_error counts timeouts; results are then wrong, not hung).
// The counter is self-resetting: the slices arrive a second time after their share, the last one stores 0.
// =====================================================================================================================
static __device__ unsigned lr_sk_error = 0;

#ifdef LR_DEV_VARIANTS
// one 32-row x 160-column block: sum of the partials + epilogue + statistics.  >= 320 threads; red: >= 16 * 336 floats of LDS.
template <typename T, typename RowFn>
__device__ __forceinline__ void sk_reduce_block(const GemmParams& P, RowFn row_m, const int rb, const int mbv, const int nbase,
                                                float* red, const int tid) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int LD = 16 * 20 + 16;
  const int cg = tid % 20, rr = tid / 20;      // (row of the 16-row pass, 8-channel group); tid < 320 active
  const bool act = tid < 320;
  const bool stats = P.gs_out != nullptr || P.gp_out != nullptr;
  float ssum = 0.f;
  for (int pass = 0; pass < 2; ++pass) {
    const int m = act ? row_m(pass * 16 + rr) : -1;
    const int n = nbase + cg * 8;
    const bool ok = act && m >= 0 && m < P.M && n < P.N;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.f;
    if (ok) {
      const float* src = P.ws + (size_t)m * P.N + n;
      const size_t slice = (size_t)P.M * P.N;
      int sidx = 0;
      for (; sidx + 4 <= P.splits; sidx += 4) {      // four slices in flight, summed in slice order
        f32x4 a[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          a[k] = *reinterpret_cast<const f32x4*>(src + (sidx + k) * slice);
          b[k] = *reinterpret_cast<const f32x4*>(src + (sidx + k) * slice + 4);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          v[0] += a[k][0]; v[1] += a[k][1]; v[2] += a[k][2]; v[3] += a[k][3];
          v[4] += b[k][0]; v[5] += b[k][1]; v[6] += b[k][2]; v[7] += b[k][3];
        }
      }
      for (; sidx < P.splits; ++sidx) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(src + sidx * slice), b = *reinterpret_cast<const f32x4*>(src + sidx * slice + 4);
        v[0] += a[0]; v[1] += a[1]; v[2] += a[2]; v[3] += a[3]; v[4] += b[0]; v[5] += b[1]; v[6] += b[2]; v[7] += b[3];
      }
      if (P.bias) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += P.bias[n + i];
      }
      if (P.rowvec) {
        float e[8];
        lr_unpack8<T>(*reinterpret_cast<const uint4*>(P.rowvec + (size_t)(m / P.rows_per_batch) * P.ld_rowvec + n), e);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += e[i];
      }
      if (P.gelu) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = lr_gelu_erf(v[i]);
      }
      if (P.resid) {
        float e[8];
        lr_unpack8<T>(*reinterpret_cast<const uint4*>(P.resid + (size_t)m * P.ld_resid + n), e);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += e[i];
      }
      const uint4 pk = lr_pack8<T>(v);
      *reinterpret_cast<uint4*>(P.out + (size_t)m * P.ld_out + n) = pk;
      if (stats) lr_unpack8<T>(pk, v);       // statistics of what the consumer will read
    }
    if (stats) {      // block-uniform
      if (act) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float x = ok ? v[i] : 0.f;
          red[(2 * i) * LD + rr * 20 + cg] = x; red[(2 * i + 1) * LD + rr * 20 + cg] = x * x;
        }
      }
      __syncthreads();
      if (act) {      // thread (pr = rr, cg): value pr of channel group cg, summed over the pass's 16 rows in a fixed order
#pragma unroll 8
        for (int k = 0; k < 16; ++k) ssum += red[rr * LD + k * 20 + cg];
      }
      __syncthreads();
    }
  }
  if (stats) {
    const int pr = rr;
    const int nn = nbase + cg * 8;
    if (act && nn < P.N && P.gs_store) P.gs_out[((size_t)rb * P.N + nn) * 2 + pr] = ssum;      // [row block][N][2], (sum, sumsq) interleaved
    if (P.gp_out) {      // per-group sums of the block's 160 channels over its 32 rows
      if (act) red[(pr & 1) * LD + cg * 8 + (pr >> 1)] = ssum;
      __syncthreads();
      const int gcg = P.gp_cg;
      if (tid < 2 * (160 / gcg)) {
        const int gl = tid >> 1, j = tid & 1;
        const int g = nbase / gcg + gl;
        if (g < 32) {
          float a = 0.f;
          for (int c = 0; c < gcg; ++c) a += red[j * LD + gl * gcg + c];
          const int smp = mbv / P.gp_hw, chunk = (mbv - smp * P.gp_hw) / 32;
          P.gp_out[(((size_t)smp * P.gp_chunks + chunk) * 32 + g) * 2 + j] = a;
        }
      }
      __syncthreads();
    }
  }
#endif
}

// tile_id: index of the output tile's counter; nsub: 32-row sub-blocks of the tile; row_fn(j, r) -> m of row r of sub-block j (or -1);
// rb_fn(j) -> statistics row-block index; mbv_fn(j) -> first (virtual linear) row of sub-block j; BN_: tile width (multiple of 160)
template <typename T, int BN_, typename RowFn, typename RbFn, typename MbvFn>
__device__ __forceinline__ void sk_fused_tail(const GemmParams& P, const int tile_id, const int nsub, RowFn row_fn, RbFn rb_fn, MbvFn mbv_fn,
                                              const int n0, float* red, const int tid) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(BN_ % 160 == 0, "the reduce block is 160 columns wide");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its write-through stores
  __syncthreads();
  unsigned* cnt = P.sk_cnt + tile_id;
  if (tid == 0) {
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)P.splits) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1u << 22)) { __hip_atomic_fetch_add(&lr_sk_error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  for (int j = blockIdx.y; j < nsub; j += P.splits) {
#pragma unroll
    for (int h = 0; h < BN_ / 160; ++h)
      sk_reduce_block<T>(P, [&](const int r) { return row_fn(j, r); }, rb_fn(j), mbv_fn(j), n0 + h * 160, red, tid);
  }
  __syncthreads();
  if (tid == 0) {      // second arrival: the last slice to finish its share re-arms the counter for the next launch
    const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == 2u * (unsigned)P.splits - 1u) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#endif
}
#endif  // LR_DEV_VARIANTS

// conv_halo.hip: the LR_PIPE_HALO instances (3x3 stride-1 conv with an LDS-resident 18 x 18 pixel patch); tile_n = 160 | 320
int lr_launch_conv_halo(const GemmParams& P, int tile_n, hipStream_t st);
unsigned lr_halo_sk_timeouts();      // conv_halo.hip's copy of lr_sk_error (synchronises the device)
