// Implicit-GEMM convolution / linear layer on the gfx950 matrix cores.
//
//   C[m][n] = sum_k A[m][k] * Wt[n][k]        m = output pixel / token, n = output channel, k = (tap, in-channel)
//
// * A is never materialised: for K-step kt the block gathers, per output row, the 128 contiguous bytes
//   (64 fp16 channels) of the shifted source pixel of tap kt / (C/64) straight from the NHWC activation into LDS
//   with 16-byte LDS-DMA (`buffer_load_dwordx4 ... lds` through a buffer descriptor: per-lane byte offset in a VGPR,
//   the K-step's channel offset in an SGPR); out-of-image taps use an out-of-range offset, which the hardware bounds
//   check turns into zeros.  Stride-2, the nearest-2x upsample and the channel concat [p1 | p2] are pure index
//   arithmetic in that gather, redone only when the tap or the concat source changes.
// * Tile 128 x BN (BN = 64 | 128), BK = 64, 4 waves (2 x 2), each wave owns 64 x BN/2 as 16x16x32 f16 MFMA tiles
//   with fp32 accumulators; LDS double-buffered, one barrier per K-step.
// * LDS rows are 128 B; the 16-byte slot of chunk c in row r is c ^ ((r >> 1) & 7).  LDS-DMA writes lane-linear, so
//   the permutation is applied to the per-lane SOURCE address and again on the ds_read_b128 side: every 16-lane
//   group of a fragment read touches 16 distinct slots of the 256-byte bank row (conflict-free).
// * MFMA operands are swapped (A-op = weight rows, B-op = activation rows) so that each lane ends up with four
//   consecutive output channels of ONE output row: the fp32 tile goes to LDS with 16-byte writes, and the epilogue
//   (bias, per-sample time-embedding row, residual, GEGLU gate) streams it out as whole fp16 lines.
#include "gemm_common.h"
#include <atomic>

#ifdef LR_GEMM_TRACE
static unsigned long long* g_trace = nullptr;
extern "C" void lr_gemm_set_trace(void* p) { g_trace = (unsigned long long*)p; }
#endif


// 2nd launch-bounds argument = waves per SIMD the register allocation must leave room for (= resident blocks per CU here)
template <int BN, int MODE, typename T>
__global__ __launch_bounds__(GEMM_THREADS, BN == 64 ? 4 : BN == 128 ? 3 : 2) void gemm_conv_kernel(const GemmParams P) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int TM = 4;            // 16-row MFMA tiles per wave along M (64 rows)
  constexpr int TN = BN / 32;      // 16-col MFMA tiles per wave along N (BN/2 cols)
  constexpr int A_BYTES = BM * 128;
  constexpr int B_BYTES = BN * 128;
  constexpr int STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w >> 1, wn = w & 1;
#ifdef LR_GEMM_STAGGER
  // Two co-resident 4-wave blocks start together and stay in lockstep: both load, both multiply, both store.  Experiment: the block
  // whose waves sit in the odd wave slot of their SIMD (HW_ID.wave_id bit 0) starts P.stagger cycles late, so that its memory phases
  // fall into the other block's matrix phase.  Measured (profiles/r04_stagger_experiment.txt): no shape gets faster, the delay only adds its length.
  if (P.stagger) {
    const unsigned slot_id = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11));      // HW_REG_HW_ID, bits [3:0] = wave_id
    if (slot_id & 1) {
      const unsigned long long t0 = __builtin_readcyclecounter();
      while (__builtin_readcyclecounter() - t0 < (unsigned long long)P.stagger) __builtin_amdgcn_s_sleep(16);
    }
  }
#endif

  // XCD-aware bijective remap: consecutive logical tiles run on the same XCD (shared A rows / halos stay in its L2)
  int bid = blockIdx.x;
  {
    const int q = P.nblocks >> 3, r = P.nblocks & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int tile_m, tile_n;
  tile_order(P, bid, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- per-thread gather metadata: 4 A rows, fixed for the whole K loop
  const int HW = P.H * P.W;
  int rb[4], ry[4], rx[4];
  const int slot = lane & 7;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (i * 4 + w) * 8 + (lane >> 3);
    const int m = m0 + row;
    if (m < P.M) {
      const int b = (int)lr_udiv((unsigned)m, P.hw_mul, P.hw_sh), rem = m - b * HW;
      const int y = (int)lr_udiv((unsigned)rem, P.w_mul, P.w_sh), x = rem - y * P.W;
      rb[i] = b * P.Hs * P.Ws;
      ry[i] = y * P.stride;
      rx[i] = x * P.stride;
    } else {
      rb[i] = 0; ry[i] = -(1 << 20); rx[i] = -(1 << 20);
    }
  }
  const int Hlim = P.Hs << P.up, Wlim = P.Ws << P.up;
  const int cpt = (P.C1 + P.C2) >> 6;   // 64-channel chunks per tap
  const int cpt1 = P.C1 >> 6;
  const int nk_main = P.taps * cpt;
  [[maybe_unused]] const bool simple = P.taps == 1 && P.p2 == nullptr && P.C3 + P.C4 == 0;      // block-uniform (kernel arguments)
  const int cpt3 = P.C3 >> 6;           // K-steps of the pointwise extension (GemmParams.p3 / p4) behind the taps
  const int nk_all = nk_main + ((P.C3 + P.C4) >> 6);
  const int k_per = (nk_all + P.splits - 1) / P.splits;
  const int k_begin = blockIdx.y * k_per;
  const int nk = min(nk_all, k_begin + k_per);   // this block runs K-steps [k_begin, nk)

  // LDS-DMA through buffer descriptors (see gemm_conv256_kernel): per-lane voffset fixed within a tap, the K-step's
  // channel offset in an SGPR, out-of-range voffset == zero padding.
  const unsigned OOB = 0x80000000u;
  const size_t a1_bytes = (size_t)P.Hs * P.Ws * P.C1 * 2 * (P.M / HW);
  const size_t a2_bytes = P.p2 ? (size_t)P.Hs * P.Ws * P.C2 * 2 * (P.M / HW) : 0;
  const size_t a3_bytes = (size_t)P.Hs * P.Ws * P.C3 * 2 * (P.M / HW);
  const size_t a4_bytes = (size_t)P.Hs * P.Ws * P.C4 * 2 * (P.M / HW);
  const int smp = P.wt_bstride ? m0 / P.rows_per_batch : 0;     // per-sample weights: the tile lies inside one sample
  const __amdgpu_buffer_rsrc_t rsW = uniform_rsrc(P.wt + (size_t)smp * P.wt_bstride, (size_t)P.N * P.K * 2);
  unsigned wvo[BN / 32];
#pragma unroll
  for (int i = 0; i < BN / 32; ++i) {
    const int row = (i * 4 + w) * 8 + (lane >> 3);
    const int n = n0 + row;
    const int chunk = slot ^ ((row >> 1) & 7);
    wvo[i] = n < P.N ? (unsigned)(((size_t)n * (P.wt_pm ? 64 : P.K) + chunk * 8) * 2) : OOB;
  }
  const unsigned kstep_bytes = P.wt_pm ? (unsigned)P.N * 128u : 128u;
  unsigned avo[4] = {OOB, OOB, OOB, OOB};
  int seg_tap = -1, seg_src = -1;
  [[maybe_unused]] int pk_kt = -2, pk_tap = 0, pk_cc = 0;      // (tap, chunk) of the last K-step staged
  auto stage = [&](int buf, int kt) {
    char* As = smem + buf * STAGE;
    char* Bs = As + A_BYTES;
    int tap, cc, srcsel;
#ifdef LR_GEMM_NO_PREP     // timing-decomposition build: no per-step bookkeeping at all (pointwise single-source shapes only)
    tap = 0; cc = kt; srcsel = 0;
    if (false) {}
#else
    if (simple) { tap = 0; cc = kt; srcsel = 0; }      // pointwise, one source: K-step kt is channel chunk kt (no tap / source arithmetic)
#endif
    else if (kt < nk_main) {
      // K-steps arrive in order: the (tap, chunk) pair is advanced, not divided out (the scalar division sat on every wave's path
      // once per K-step, between the barrier and the next fragment reads)
      if (kt == pk_kt + 1) { if (++pk_cc == cpt) { pk_cc = 0; ++pk_tap; } }
      else { pk_tap = kt / cpt; pk_cc = kt - pk_tap * cpt; }
      pk_kt = kt;
      tap = pk_tap; cc = pk_cc; srcsel = cc < cpt1 ? 0 : 1;
    }
    else { tap = 16; cc = kt - nk_main; srcsel = cc < cpt3 ? 2 : 3; }      // pointwise extension: tap (0, 0) of sources 3 / 4
    if (tap != seg_tap || srcsel != seg_src) {      // wave-uniform: new tap or crossing a source boundary
      seg_tap = tap; seg_src = srcsel;
      int dy = 0, dx = 0;
      if (P.taps == 9 && tap < 9) { dy = tap / 3 - P.pad; dx = tap - (tap / 3) * 3 - P.pad; }
      const int cs = srcsel == 0 ? P.C1 : srcsel == 1 ? P.C2 : srcsel == 2 ? P.C3 : P.C4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = (i * 4 + w) * 8 + (lane >> 3);
        const int iy = ry[i] + dy, ix = rx[i] + dx;
        const bool ok = (unsigned)iy < (unsigned)Hlim && (unsigned)ix < (unsigned)Wlim && !(((iy | ix) & 1) & P.zins);
        const int sy = iy >> P.up, sx = ix >> P.up;
        const int chunk = slot ^ ((row >> 1) & 7);
        avo[i] = ok ? (unsigned)(((size_t)(rb[i] + sy * P.Ws + sx) * cs + chunk * 8) * 2) : OOB;
      }
    }
    const unsigned coff = (unsigned)((srcsel == 1 ? cc - cpt1 : srcsel == 3 ? cc - cpt3 : cc) * 128);
    const __amdgpu_buffer_rsrc_t rsA =
        srcsel == 0 ? uniform_rsrc((const void*)P.p1, a1_bytes) : srcsel == 1 ? uniform_rsrc((const void*)P.p2, a2_bytes)
      : srcsel == 2 ? uniform_rsrc((const void*)P.p3, a3_bytes) : uniform_rsrc((const void*)P.p4, a4_bytes);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(As + ((i * 4 + w) * 8) * 128), 16, avo[i], coff, 0, 0);
    const unsigned koff = (unsigned)kt * kstep_bytes;
#pragma unroll
    for (int i = 0; i < BN / 32; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lptr_t)(Bs + ((i * 4 + w) * 8) * 128), 16, wvo[i], koff, 0, 0);
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  constexpr int PAR_LD = ((BN + 63) / 64) * 64;
  float* rs = reinterpret_cast<float*>(smem + 2 * STAGE);
  float* par = rs + 2 * BM;
  stage_params<BN, PAR_LD>(P, par, n0, w, lane, smp);
  if (k_begin < nk) stage(0, k_begin);
  __syncthreads();
  int cur = 0;
  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = k_begin; kt < nk; ++kt) {
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const char* As = smem + cur * STAGE;
    const char* Bs = As + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      vec8<T> xf[TM], wf[TN];
      const int kc = ks * 4 + fq;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * 64 + i * 16 + fr;
        xf[i] = *reinterpret_cast<const vec8<T>*>(As + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = weight_tile<TN, 2, MODE == 1>(wn, j) * 16 + fr;
        wf[j] = *reinterpret_cast<const vec8<T>*>(Bs + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
          acc[j][i] = lr_mfma16(wf[j], xf[i], acc[j][i]);
    }
    __syncthreads();  // all reads of buf[cur] done; all LDS-DMA into buf[cur^1] landed (vmcnt(0) precedes the barrier)
    cur ^= 1;
  }

  // ---- epilogue straight from the accumulators (see epilogue_direct)
  if (P.ln_part) {
    ln_rows_to_lds(P, rs, m0, BM, t);
    __syncthreads();
  }
  // (the K loop ended with a barrier: the stage buffers are free for the per-channel sums of gn_group_reduce)
  float* gsl = reinterpret_cast<float*>(smem);
  epilogue_units<TM, TN, MODE, PAR_LD, 2, T>(P, acc, m0 + wm * 64, n0, wn, lane, rs + 2 * (wm * 64), par, tile_n * 2 + wn,
                                             gsl + wm * (BN * 2));
  if constexpr (MODE == 0) {
    if (P.gp_out != nullptr && P.splits == 1) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      gn_group_reduce<BN, 2, BM>(P, gsl, m0, n0, t);
    }
  }
#endif  // __HIP_DEVICE_COMPILE__
}

// =====================================================================================================================
// 256 x BN tile, 8 waves (4 x 2), 3-stage LDS-DMA pipeline with counted vmcnt and raw barriers.
//
// Why: at 128 x 128 x 64 the block moves 32 KB L2->LDS per 2.1 MFLOP; at full MFMA rate that is ~36 TB/s chip-wide,
// i.e. the per-XCD L2s, not the matrix cores, are the roof.  256 x 160 moves 52 KB per 5.2 MFLOP (1.55x less traffic
// per flop; 2.3x less than the 128 x 64 tile used for N = 320) and BN = 160 divides every channel count of the model.
// One 8-wave block per CU cannot hide a vmcnt(0)+barrier drain behind another block, so the loads run two K-steps ahead:
//   wait(vmcnt = loads of the NEXT stage still in flight) -> s_barrier -> issue stage kt+2 -> ds_read/MFMA on stage kt.
// The gather addresses are kept per row as a base pointer + validity bit and only recomputed when the tap or the
// concat source changes (every C/64 K-steps); inside a tap the K-step offset is a scalar add.
// =====================================================================================================================


// BM2 x BN tile, NW waves of which WMW along M (the other NW / WMW along N); NSTAGE = depth of the LDS-DMA ring.
// Instances <BM2, NW, BN, WMW, NSTAGE>:
//   <256, 8, 128, 4, 3>, <256, 8, 160, 4, 3>: wave tile 64 x BN/2, 3-stage ring (loads two K-steps ahead)
//   <256, 8, 320, 2, 2>: wave tile 128 x 80 (40 accumulator tiles): 28 % fewer LDS bytes per MFMA and 31 % less
//                        L2->LDS traffic per flop than 256 x 160; 2-stage ring (144 KB); N = 320 is ONE tile wide.
//   <256, 8, 256, 2, 2>: wave tile 128 x 64 for N = 256 / 512 (the VAE's widths) and the GEGLU projections;
//   <256, 8, 320, 4, 2>: wave tile 64 x 160 (even number of N tiles per wave, needed by the GEGLU u|g pairing).
//   <128, 8, 128, 4, 4>, <128, 8, 160, 4, 4>: the small-M levels (M = 4096 / 1024 rows: 128 x 160 tiles of
//                        4096 x 1280 are exactly 256 blocks).  The block's K-step is only 2.6 MFLOP, so the loads run
//                        THREE K-steps ahead (4-stage ring, 147 KB, one block per CU) and the 8 waves (4 x 2, wave tile
//                        32 x BN/2) give every SIMD two waves to overlap ds_reads, LDS-DMA issue and MFMAs -- with the
//                        2-stage ring of gemm_conv_kernel every K-step of these shapes waits out a full memory latency,
//                        and a 4-wave version of this instance spent 2080 cycles per K-step for 680 cycles of MFMA.
template <int BM2, int NW, int BN, int WMW, int NSTAGE, int MODE, typename T>
__global__ __launch_bounds__(NW * 64) void gemm_conv_pipe_kernel(const GemmParams P) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int WNW = NW / WMW;
  constexpr int TM = BM2 / WMW / 16;    // 16-row MFMA tiles per wave along M
  constexpr int TN = BN / WNW / 16;     // 16-col MFMA tiles per wave along N
  constexpr bool DB = TM * TN <= 20;    // double-buffer the fragments in registers when the accumulators leave room
  constexpr int A_BYTES = BM2 * 128;
  constexpr int B_BYTES = BN * 128;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int NA = BM2 / (NW * 8);    // A staging instructions per wave (8 rows of 128 B each)
  constexpr int NB_FULL = BN / (NW * 8);   // B staging instructions issued by every wave
  constexpr bool B_TAIL = (BN % (NW * 8)) != 0;  // one more instruction for the first waves (rows 128..159 when BN = 160, NW = 8)
  constexpr int TAIL_WAVES = (BN % (NW * 8)) / 8;
  static_assert(NSTAGE >= 2 && NSTAGE <= 4, "ring depth");
  static_assert((NSTAGE - 1) * (NA + NB_FULL + 1) <= 63, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w / WNW, wn = w % WNW;

  // XCD-aware bijective remap: hardware block b lives on XCD b & 7; XCD x owns the contiguous logical tiles
  // [t_begin, t_begin + t_cnt) (shared A rows / halos stay in its L2).
  // (Persistent variants were built and measured twice in round 2.  (1) min(tiles, CUs) blocks that prefetch the next
  // tile's first stages between main loop and epilogue: slower on every shape (level-0 GEGLU 220 vs 184 us; the next
  // tile's gather state stays live across an epilogue at the 256-VGPR limit, and the tile start drained the stores).
  // (2) ONE continuous LDS ring over a block's tiles -- the refill slots of a tile's last K-steps take the next tile's
  // first stages, the gather state is switched in place, stores stay in flight across the boundary; bit-identical, no
  // spills on 256 x {128, 256}: level-0 GEGLU 181 vs 184 us (256 x 256), 203 vs 213 (256 x 128), levels 1 / 2 +-1 %.
  // The prologue is not what these short-K tiles wait for; neither variant was kept.)
  const int xcd = blockIdx.x & 7;
  const int q8 = P.nblocks >> 3, r8 = P.nblocks & 7;
  const int t_begin = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int tl = blockIdx.x >> 3;       // this block's tile, local to the XCD's range
  [[maybe_unused]] int lr_trace_tile = 0;

  const int HW = P.H * P.W;
  const int slot = lane & 7;
  const unsigned OOB = 0x80000000u;
  int tile_m = 0, tile_n = 0, m0 = 0, n0 = 0, smp = 0;
  unsigned wvo[NB_FULL + 1];      // weight rows (fixed over K): instr i covers rows (i*NW + w)*8 + lane/8
  // gather state: per-row byte offset of the current (tap, source) segment, recomputed only when the tap or the concat
  // source changes
  unsigned avo[NA];
  int seg_tap = -1, seg_src = -1;
  [[maybe_unused]] int pk_kt = -2, pk_tap = 0, pk_cc = 0;      // (tap, chunk) of the last K-step staged
  // descriptors of the current A segment and of the weights (zero-length for a stage past the end of K); rebuilt from
  // the kernel arguments when the segment / tile changes instead of holding 4 SGPRs per operand
  __amdgpu_buffer_rsrc_t rsA = uniform_rsrc((const void*)P.wt, 0), rsB = rsA;
  auto setup_tile = [&](int tloc) __attribute__((always_inline)) {
    tile_order(P, t_begin + tloc, tile_m, tile_n);
    lr_trace_tile = t_begin + tloc;
    m0 = tile_m * BM2; n0 = tile_n * BN;
#pragma unroll
    for (int i = 0; i < NA; ++i) avo[i] = OOB;
#pragma unroll
    for (int i = 0; i < NB_FULL + 1; ++i) {
      const int row = (i * NW + w) * 8 + (lane >> 3);
      const int n = n0 + row;
      const int chunk = slot ^ ((row >> 1) & 7);
      wvo[i] = (row < BN && n < P.N) ? (unsigned)(((size_t)n * (P.wt_pm ? 64 : P.K) + chunk * 8) * 2) : OOB;
    }
    seg_tap = -1; seg_src = -1;
    smp = P.wt_bstride ? m0 / P.rows_per_batch : 0;      // per-sample weights: the tile lies inside one sample
    rsB = uniform_rsrc(P.wt + (size_t)smp * P.wt_bstride, (size_t)P.N * P.K * 2);
  };
  const int Hlim = P.Hs << P.up, Wlim = P.Ws << P.up;
  const int cpt = P.c16 ? 1 : (P.C1 + P.C2) >> 6;
  const int cpt1 = P.c16 ? 1 : P.C1 >> 6;
  const int nk_main = P.c16 ? 3 : P.taps * cpt;
  [[maybe_unused]] const bool simple = P.taps == 1 && !P.c16 && P.p2 == nullptr && P.C3 + P.C4 == 0;      // block-uniform (kernel arguments)
  const int cpt3 = P.C3 >> 6;                           // K-steps of the pointwise extension (GemmParams.p3 / p4): source 3, then 4
  const int nk_all = nk_main + ((P.C3 + P.C4) >> 6);
  const int k_per = (nk_all + P.splits - 1) / P.splits;
  const int k_begin = blockIdx.y * k_per;
  const int nk = min(nk_all, k_begin + k_per);

  // LDS-DMA through buffer descriptors: address = SRD base + per-lane voffset (VGPR, fixed within a tap) + soffset
  // (SGPR, the K-step's channel offset).  A K-step therefore issues its 6-7 `buffer_load_dwordx4 ... lds` with NO
  // per-lane address arithmetic, and padding / tails need no zero page: an out-of-range voffset reads as 0.
  const size_t a1_bytes = (size_t)P.Hs * P.Ws * P.C1 * 2 * (P.M / HW);
  const size_t a2_bytes = P.p2 ? (size_t)P.Hs * P.Ws * P.C2 * 2 * (P.M / HW) : 0;
  const size_t a3_bytes = (size_t)P.Hs * P.Ws * P.C3 * 2 * (P.M / HW);
  const size_t a4_bytes = (size_t)P.Hs * P.Ws * P.C4 * 2 * (P.M / HW);
  setup_tile(tl);
  // A K-step's staging is split in two: stage_prepare (wave-uniform control flow: new gather offsets when the tap or
  // the concat source changes; the descriptor / scalar offsets of the step) and stage_issue (a straight line of
  // NA + NB_FULL `buffer_load ... lds`), so that the main loop can spread the issue over the MFMAs of a half K-step:
  // an LDS-DMA instruction costs its wave ~60 cycles of issue among MFMAs but 100-185 in a back-to-back burst next to
  // the ds_reads (MI355X_MICROARCH.md, "LDS-DMA piece issue cost").
  // a stage past the end of this block's K range is issued all the same, through zero-length descriptors (every lane
  // out of range: no memory traffic, zeros into an LDS buffer nobody reads any more) -- the number of LDS-DMA
  // instructions in flight is then the same in every iteration and all vmcnt waits are constants
  unsigned coff = 0, koff = 0;
  auto stage_prepare = [&](int buf, int kt) __attribute__((always_inline)) {
    if (kt >= nk) {      // (no real stage follows in this tile; setup_tile / the next segment change restore them)
      rsA = uniform_rsrc((const void*)P.wt, 0); rsB = rsA;
    } else if (P.c16) {
      // 16-channel source: this lane's chunk of each gathered row belongs to tap 4 kt + (chunk >> 1) (taps >= 9: zero padding of K)
      rsA = uniform_rsrc((const void*)P.p1, a1_bytes);
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int row = (i * NW + w) * 8 + (lane >> 3);
        const int chunk = slot ^ ((row >> 1) & 7);
        const int tap = kt * 4 + (chunk >> 1);
        const int t3 = (tap * 11) >> 5;                        // tap / 3 for tap < 12
        const int dy = t3 - P.pad, dx = tap - 3 * t3 - P.pad;
        const unsigned m = (unsigned)(m0 + row);
        const unsigned b = lr_udiv(m, P.hw_mul, P.hw_sh), rem = m - b * (unsigned)HW;
        const unsigned y = lr_udiv(rem, P.w_mul, P.w_sh), x = rem - y * (unsigned)P.W;
        const int iy = (int)y + dy, ix = (int)x + dx;
        const bool ok = (int)m < P.M && tap < 9 && (unsigned)iy < (unsigned)Hlim && (unsigned)ix < (unsigned)Wlim;
        avo[i] = ok ? (unsigned)(((size_t)((int)b * P.Hs * P.Ws + iy * P.Ws + ix) * 16 + (chunk & 1) * 8) * 2) : OOB;
      }
      seg_tap = -1; seg_src = -1;
      coff = 0;
      koff = (unsigned)kt * (P.wt_pm ? (unsigned)P.N * 128u : 128u);
    } else {
    int tap, cc, srcsel;
#ifdef LR_GEMM_NO_PREP     // timing-decomposition build: no per-step bookkeeping at all (pointwise single-source shapes only)
    tap = 0; cc = kt; srcsel = 0;
    if (false) {}
#else
    if (simple) { tap = 0; cc = kt; srcsel = 0; }      // pointwise, one source: K-step kt is channel chunk kt (no tap / source arithmetic)
#endif
    else if (kt < nk_main) {
      // K-steps arrive in order: the (tap, chunk) pair is advanced, not divided out (the scalar division sat on every wave's path
      // once per K-step, between the barrier and the next fragment reads)
      if (kt == pk_kt + 1) { if (++pk_cc == cpt) { pk_cc = 0; ++pk_tap; } }
      else { pk_tap = kt / cpt; pk_cc = kt - pk_tap * cpt; }
      pk_kt = kt;
      tap = pk_tap; cc = pk_cc; srcsel = cc < cpt1 ? 0 : 1;
    }
    else { tap = 16; cc = kt - nk_main; srcsel = cc < cpt3 ? 2 : 3; }      // pointwise extension: tap (0, 0) of sources 3 / 4
    if (tap != seg_tap || srcsel != seg_src) {      // wave-uniform
      seg_tap = tap; seg_src = srcsel;
      int dy = 0, dx = 0;
      if (P.taps == 9 && tap < 9) { dy = tap / 3 - P.pad; dx = tap - (tap / 3) * 3 - P.pad; }
      const int cs = srcsel == 0 ? P.C1 : srcsel == 1 ? P.C2 : srcsel == 2 ? P.C3 : P.C4;
      rsA = srcsel == 0 ? uniform_rsrc((const void*)P.p1, a1_bytes) : srcsel == 1 ? uniform_rsrc((const void*)P.p2, a2_bytes)
          : srcsel == 2 ? uniform_rsrc((const void*)P.p3, a3_bytes) : uniform_rsrc((const void*)P.p4, a4_bytes);
      // (sample, y, x) of each gathered row are re-derived here (a few dozen VALU ops per tap) rather than kept in 3 * NA
      // registers for the whole tile: the persistent loop keeps the next tile's gather state live across the epilogue
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int row = (i * NW + w) * 8 + (lane >> 3);
        const unsigned m = (unsigned)(m0 + row);
        const unsigned b = lr_udiv(m, P.hw_mul, P.hw_sh), rem = m - b * (unsigned)HW;
        const unsigned y = lr_udiv(rem, P.w_mul, P.w_sh), x = rem - y * (unsigned)P.W;
        const int iy = (int)y * P.stride + dy, ix = (int)x * P.stride + dx;
        const bool ok = (int)m < P.M && (unsigned)iy < (unsigned)Hlim && (unsigned)ix < (unsigned)Wlim &&
                        !(((iy | ix) & 1) & P.zins);
        const int sy = iy >> P.up, sx = ix >> P.up;
        const int chunk = slot ^ ((row >> 1) & 7);
        avo[i] = ok ? (unsigned)(((size_t)((int)b * P.Hs * P.Ws + sy * P.Ws + sx) * cs + chunk * 8) * 2) : OOB;
      }
    }
    coff = (unsigned)((srcsel == 1 ? cc - cpt1 : srcsel == 3 ? cc - cpt3 : cc) * 128);
    koff = (unsigned)kt * (P.wt_pm ? (unsigned)P.N * 128u : 128u);
    }
    if (B_TAIL && w < TAIL_WAVES)     // the odd weight rows (first waves only): issued here, ahead of the step's other loads
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lptr_t)(smem + buf * STAGE + A_BYTES + ((NB_FULL * NW + w) * 8) * 128), 16,
                                               wvo[NB_FULL], koff, 0, 0);
  };
  auto stage_issue = [&](int buf) __attribute__((always_inline)) {
    char* As = smem + buf * STAGE;
    char* Bs = As + A_BYTES;
#ifdef LR_GEMM_NO_DMA      // timing-decomposition build (tools/sweep_decomp.sh): no operand traffic, the loop multiplies whatever the LDS holds
    return;
#endif
#pragma unroll
    for (int i = 0; i < NA; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(As + ((i * NW + w) * 8) * 128), 16, avo[i], coff, 0, 0);
#pragma unroll
    for (int i = 0; i < NB_FULL; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lptr_t)(Bs + ((i * NW + w) * 8) * 128), 16, wvo[i], koff, 0, 0);
  };
  auto stage = [&](int buf, int kt) __attribute__((always_inline)) { stage_prepare(buf, kt); stage_issue(buf); };

  f32x4 acc[TN][TM];

  // ---- main loop.  Fragments are double-buffered in registers: while the 20 MFMAs of one half K-step (k = 32) run,
  // the ds_read_b128 of the NEXT half step are already in flight, so the matrix pipe never waits on LDS latency and the
  // eight waves do not all hit the LDS port at once.  One barrier per K-step, placed between the two halves: at that
  // point every wave has finished reading stage kt (-> its buffer is refilled with stage kt+3) and stage kt+1 has landed.
  const int fr = lane & 15, fq = lane >> 4;
  auto read_frags = [&](vec8<T> (&xf)[TM], vec8<T> (&wf)[TN], int buf, int ks) {
    const char* As = smem + buf * STAGE;
    const char* Bs = As + A_BYTES;
    const int kc = ks * 4 + fq;
#ifdef LR_GEMM_NO_READS    // timing-decomposition build: fragments from registers (no ds_read traffic)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) xf[i][e] = (T)(float)(buf + ks + i);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) wf[j][e] = (T)(float)(buf - ks + j);
    return;
#endif
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = wm * (TM * 16) + i * 16 + fr;
      xf[i] = *reinterpret_cast<const vec8<T>*>(As + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row = weight_tile<TN, WNW, MODE == 1>(wn, j) * 16 + fr;
      wf[j] = *reinterpret_cast<const vec8<T>*>(Bs + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
    }
  };
  auto mma = [&](const vec8<T> (&xf)[TM], const vec8<T> (&wf)[TN]) {
#ifdef LR_GEMM_SETPRIO      // developer A/B build: raise the wave's priority over its SIMD partner while it issues matrix work
    __builtin_amdgcn_s_setprio(1);
#endif
#ifdef LR_GEMM_NO_MFMA     // timing-decomposition build: the fragments are consumed by one VALU add per accumulator tile instead of an MFMA
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i) acc[j][i][0] += (float)wf[j][0] + (float)xf[i][0];
#else
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i)
        acc[j][i] = lr_mfma16(wf[j], xf[i], acc[j][i]);
#endif
#ifdef LR_GEMM_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
  };
  // wait until all of this wave's LDS-DMA except the newest INFLIGHT stages has landed
  auto wait_stages = [&](auto inflight) __attribute__((always_inline)) {
    constexpr int F = decltype(inflight)::value;
    constexpr int L0 = NA + NB_FULL, L1 = L0 + 1;   // LDS-DMA instructions per stage of a wave without / with the tail
    if (B_TAIL && w < TAIL_WAVES) wait_vmcnt<F * L1>();
    else wait_vmcnt<F * L0>();
  };
  using std::integral_constant;

  const int nsteps = nk - k_begin;
  constexpr int PAR_LD = ((BN + 63) / 64) * 64;
  float* rs = reinterpret_cast<float*>(smem + NSTAGE * STAGE);
  float* par = rs + 2 * BM2;
  // DB: the whole ring is filled up front and K-step `it` refills its own buffer (stage it + NSTAGE) from the middle of
  // the step on; !DB: NSTAGE - 1 stages up front, stage it + NSTAGE - 1 goes out during the first half of K-step `it`
  // (its buffer was released by the barrier that ended step it - 1).
  constexpr int NPRO = DB ? NSTAGE : NSTAGE - 1;
  constexpr int NDMA = NA + NB_FULL;
  constexpr int MFMA_PER = (TM * TN) / NDMA > 0 ? (TM * TN) / NDMA : 1;
  // MFMAs of one half K-step with the prepared stage's LDS-DMA instructions spread between them
  auto mma_issue = [&](const vec8<T> (&xf)[TM], const vec8<T> (&wf)[TN], int buf) {
    stage_issue(buf);
    mma(xf, wf);
#pragma unroll
    for (int g = 0; g < NDMA; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x8, MFMA_PER, 0);
      __builtin_amdgcn_sched_group_barrier(0x10, 1, 0);
    }
  };
  LR_STAMP(0);
  stage_params<BN, PAR_LD>(P, par, n0, w, lane, smp);
#pragma unroll
  for (int sidx = 0; sidx < NPRO; ++sidx) stage(sidx, k_begin + sidx);
  if constexpr (!DB) stage_prepare(NSTAGE - 1, k_begin + NSTAGE - 1);
  LR_STAMP(1);
  {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    vec8<T> xa[TM], wa[TN];
    wait_stages(integral_constant<int, NPRO - 1>{});        // stage 0 (and the column parameters) landed
    __builtin_amdgcn_s_barrier();
    if constexpr (DB) { if (nsteps > 0) read_frags(xa, wa, 0, 0); }
    LR_STAMP(2);
    int cur = 0;
    if constexpr (DB) {
      vec8<T> xb[TM], wb[TN];
      for (int it = 0; it + 1 < nsteps; ++it) {
        const int nxt = cur == NSTAGE - 1 ? 0 : cur + 1;
        // sched_barrier(0) pins the issue order [reads of the next half] -> [MFMAs of the current half]
        read_frags(xb, wb, cur, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(xa, wa);
        __builtin_amdgcn_sched_barrier(0);
        wait_stages(integral_constant<int, NSTAGE - 2>{});   // stage it+1 landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my reads of stage `it` are in registers
        __builtin_amdgcn_s_barrier();
        stage_prepare(cur, k_begin + it + NSTAGE);           // refill the buffer every wave has finished with ...
        read_frags(xa, wa, nxt, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma_issue(xb, wb, cur);                              // ... between the MFMAs of the second half
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
      }
      if (nsteps > 0) {
        read_frags(xb, wb, cur, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(xa, wa);
        __builtin_amdgcn_sched_barrier(0);
        mma(xb, wb);
      }
    } else {
      int pbuf = NSTAGE - 1;      // buffer of the prepared, not yet issued stage
      for (int it = 0; it + 1 < nsteps; ++it) {
        const int nxt = cur == NSTAGE - 1 ? 0 : cur + 1;
        read_frags(xa, wa, cur, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma_issue(xa, wa, pbuf);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(xa, wa, cur, 1);
        mma(xa, wa);
        wait_stages(integral_constant<int, NSTAGE - 2>{});   // stage it+1 landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stage_prepare(cur, k_begin + it + NSTAGE);
        pbuf = cur;
        cur = nxt;
      }
      if (nsteps > 0) {
        read_frags(xa, wa, cur, 0);
        mma(xa, wa);
        read_frags(xa, wa, cur, 1);
        mma(xa, wa);
      }
    }
    LR_STAMP(3);
    // ---- epilogue straight from the accumulators (see epilogue_units)
    if (P.ln_part) {
      ln_rows_to_lds(P, rs, m0, BM2, t);
      __syncthreads();
    }
    LR_STAMP(4);
    float* gsl = reinterpret_cast<float*>(smem);
    const bool gp = MODE == 0 && P.gp_out != nullptr && P.splits == 1;      // block-uniform
    if (gp) {      // every wave has left the K loop and its (zero-length) trailing LDS-DMA has landed: the stage buffers are free
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    epilogue_units<TM, TN, MODE, PAR_LD, WNW, T>(P, acc, m0 + wm * (TM * 16), n0, wn, lane, rs + 2 * (wm * TM * 16), par,
                                              tile_n * WNW + wn, gsl + wm * (BN * 2));
    if constexpr (MODE == 0) {
      if (gp) {      // only the LDS writes of the per-channel sums have to be visible: the epilogue's global stores stay in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        gn_group_reduce<BN, WMW, BM2>(P, gsl, m0, n0, t);
      }
    }
#ifdef LR_DEV_VARIANTS
    if constexpr (MODE != 1 && BN % 160 == 0) {
      if (P.sk_cnt != nullptr && P.splits > 1)      // in-launch split-K reduce (gemm_common.h): this slice's share of the tile
        sk_fused_tail<T, BN>(P, tile_m * P.ntiles_n + tile_n, BM2 / 32,
                             [&](const int j, const int r) { return m0 + 32 * j + r; },
                             [&](const int j) { return (m0 >> 5) + j; }, [&](const int j) { return m0 + 32 * j; }, n0,
                             reinterpret_cast<float*>(smem), t);
    }
#endif
    LR_STAMP(5);
#ifdef LR_GEMM_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LR_STAMP(6);
#endif
  }
#endif  // __HIP_DEVICE_COMPILE__
}

template <int BM2, int NW, int BN, int WMW, int NSTAGE, int MODE, typename T>
static int launch_pipe_t(const GemmParams& P0, hipStream_t st) {
  GemmParams P = P0;
  P.ntiles_n = (P.N + BN - 1) / BN;
  const int ntm = (P.M + BM2 - 1) / BM2;
  P.ntiles_m = ntm;
  // Tile order inside an XCD's contiguous range (round 5): panels of G row tiles, m-fastest inside a panel, so that the ~32 tiles an XCD
  // runs at once (one 8-wave block per CU) are G x (32 / G) row / column slices instead of 2 x 16 (n-fastest) -- wide-N GEMMs whose
  // CUs run several tiles back to back (GEGLU / qkv projections: 4 rounds) drift out of lockstep, and with n-fastest an XCD then streams
  // the whole weight matrix per two row tiles.  Measured: 16384 x 5120 x 640 GEGLU 152 -> 140 us, 4096 x 10240 x 1280 129 -> 117 us,
  // 16384 x 1920 x 640 74 -> 68 us, UNet step -0.25 ... -0.32 ms on three boxes (G = 8 / 16; G = 32: -0.07; plain m-fastest was
  // measured slower in round 1: 928 vs 1038 TFLOP/s).  Pure scheduling: same tiles, same bits.
  // (developer build, LR_DEV: LR_GEMM_GROUP_M / LR_GEMM_GROUP_M2 override G for wide / narrow N, 0 = n-fastest; LR_GEMM_GROUP_CONV groups the
  // 3x3 gathers too; LR_GATHER_MFASTEST = 1: split-K 3x3 gathers walk m-fastest -- measured: fetch DOWN at 1024 rows (1024 x 1280 x 11520:
  // 169 -> 96 MB) but UP at 4096 rows (378 -> 492 MB; this kernel re-gathers its rows per tap), step time unchanged)
  {
    const int g_env = LR_DEV("LR_GEMM_GROUP_M", -1), g_env2 = LR_DEV("LR_GEMM_GROUP_M2", -1);
    int G = 0;
    if (P.ntiles_n > 1) G = P.ntiles_n >= 4 ? 8 : 16;
    if (g_env >= 0 && P.ntiles_n >= 4) G = g_env;
    if ((g_env2 >= 0 || g_env >= 0) && P.ntiles_n > 1 && P.ntiles_n < 4) G = g_env2 >= 0 ? g_env2 : g_env;
    if (P.taps != 1 && !LR_DEV("LR_GEMM_GROUP_CONV", 0)) G = 0;      // 3x3 gathers re-read their rows per tap: column-tile neighbours on one XCD are worth more there (measured 0 ... +2 %)
    P.m_fastest = (G > 1 && ntm > 1) ? G : 0;
    if (LR_DEV("LR_GATHER_MFASTEST", 0) && P.taps != 1 && P.splits > 1 && ntm > 1) P.m_fastest = 1;
  }
  P.nblocks = P.ntiles_n * ntm;
  // stages + (mean, rstd) rows + (bias, ln_colsum) columns
  const size_t smem = NSTAGE * (size_t)(BM2 + BN) * 128 + BM2 * 2 * sizeof(float) + 2 * (((BN + 63) / 64) * 64) * sizeof(float);
  static unsigned long long attr_done = 0;
  if (lr_attr_needed(&attr_done)) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_conv_pipe_kernel<BM2, NW, BN, WMW, NSTAGE, MODE, T>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  const int gx = P.nblocks;
  hipLaunchKernelGGL((gemm_conv_pipe_kernel<BM2, NW, BN, WMW, NSTAGE, MODE, T>), dim3(gx, P.splits), dim3(NW * 64), smem, st, P);
  return lr_launch_status();
}

// Fixed-order reduction of the split-K partials + the fused epilogue (deterministic: no atomics).
// Block = RED_GROUPS channel groups (8 channels = 16 B of output each) x RED_ROWS = 32 output rows, one (row, group) per thread;
// the partial loads of a thread are independent and issued four splits at a time (the round-1 kernel walked the splits
// with one dependent load pair per iteration: 11 us for 20-30 MB).  With gs_out the block also emits the per-channel
// (sum, sumsq) of its 32 rows of the ROUNDED output -- the GroupNorm statistics of the consumer -- so a split-K producer
// no longer needs the stand-alone statistics pass.
#define RED_ROWS 32
#define RED_GROUPS 20     // 8-channel groups per block: 160 channels x 32 rows = 640 threads; 1024 x 1280 outputs -> 256 blocks
                          // (160 = a whole number of GroupNorm groups for C = 320 / 640 / 1280: the block also emits per-GROUP sums)
template <typename T>
__global__ __launch_bounds__(RED_ROWS * RED_GROUPS) void splitk_reduce_kernel(const GemmParams P) {
  // [value j of 16][row][group] with 16 floats of padding per j: the writes of a half-wave (16 groups x 2 rows) and the column
  // reads of a half-wave (16 groups x 2 values) both touch 32 distinct banks (the round-2 [row][group][17] layout put two of the
  // four channel groups a wave reads on the same banks: 33 % conflict cycles, profiles/r02_pmc_util.txt)
  __shared__ float red[16][RED_ROWS * RED_GROUPS + 16];
  const int tx = threadIdx.x % RED_GROUPS, ty = threadIdx.x / RED_GROUPS;
  const int n = (blockIdx.x * RED_GROUPS + tx) * 8;
  const int m = blockIdx.y * RED_ROWS + ty;
  const bool ok = n < P.N && m < P.M;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = 0.f;
  if (ok) {
    const float* src = P.ws + (size_t)m * P.N + n;
    const size_t slice = (size_t)P.M * P.N;
    int sidx = 0;
    if (P.splits > 4 && P.splits <= 8) {      // every partial of the thread in flight at once (one latency instead of two)
      f32x4 a[8], b[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int kk = k < P.splits ? k : 0;
        a[k] = *reinterpret_cast<const f32x4*>(src + kk * slice);
        b[k] = *reinterpret_cast<const f32x4*>(src + kk * slice + 4);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {     // same order as a plain loop over the splits
        if (k < P.splits) {
          v[0] += a[k][0]; v[1] += a[k][1]; v[2] += a[k][2]; v[3] += a[k][3];
          v[4] += b[k][0]; v[5] += b[k][1]; v[6] += b[k][2]; v[7] += b[k][3];
        }
      }
      sidx = P.splits;
    }
    for (; sidx + 4 <= P.splits; sidx += 4) {
      f32x4 a[4], b[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        a[k] = *reinterpret_cast<const f32x4*>(src + (sidx + k) * slice);
        b[k] = *reinterpret_cast<const f32x4*>(src + (sidx + k) * slice + 4);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {     // same order as a plain loop over the splits
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
    if (P.gs_out || P.gp_out) lr_unpack8<T>(pk, v);       // statistics of what the consumer will read
  }
  if (P.gs_out || P.gp_out) {      // block-uniform; gs_store: write the per-channel form too
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float x = ok ? v[i] : 0.f;
      red[2 * i][ty * RED_GROUPS + tx] = x; red[2 * i + 1][ty * RED_GROUPS + tx] = x * x;
    }
    __syncthreads();
    // RED_GROUPS channel groups x 16 values, summed over the 32 rows in a fixed order by the first 16 RED_GROUPS threads
    float s = 0.f;
    const int pr = threadIdx.x / RED_GROUPS, cg = threadIdx.x % RED_GROUPS;
    if (threadIdx.x < 16 * RED_GROUPS) {
#pragma unroll 8
      for (int k = 0; k < RED_ROWS; ++k) s += red[pr][k * RED_GROUPS + cg];
      const int nn = (blockIdx.x * RED_GROUPS + cg) * 8;
      if (nn < P.N && P.gs_store) P.gs_out[((size_t)blockIdx.y * P.N + nn) * 2 + pr] = s;      // [row block][N][2], (sum, sumsq) interleaved
    }
    if (P.gp_out) {      // per-group sums of the block's 160 channels over its 32 rows (block-uniform)
      __syncthreads();
      // value pr = 2 i + j of channel group cg = (sum | sumsq)[j] of channel 8 cg + i  ->  red[j][channel]
      if (threadIdx.x < 16 * RED_GROUPS) red[pr & 1][cg * 8 + (pr >> 1)] = s;
      __syncthreads();
      const int gcg = P.gp_cg, t = threadIdx.x;
      if (t < 2 * (8 * RED_GROUPS / gcg)) {
        const int gl = t >> 1, j = t & 1;
        const int g = (blockIdx.x * 8 * RED_GROUPS) / gcg + gl;
        if (g < 32) {
          float a = 0.f;
          for (int c = 0; c < gcg; ++c) a += red[j][gl * gcg + c];
          const int m0 = blockIdx.y * RED_ROWS;
          const int smp = m0 / P.gp_hw, chunk = (m0 - smp * P.gp_hw) / RED_ROWS;
          P.gp_out[(((size_t)smp * P.gp_chunks + chunk) * 32 + g) * 2 + j] = a;
        }
      }
    }
  }
}

template <int BN, int MODE, typename T>
static int launch_gemm_t(const GemmParams& P0, hipStream_t st) {
  GemmParams P = P0;
  P.ntiles_n = (P.N + BN - 1) / BN;
  const int ntm = (P.M + BM - 1) / BM;
  P.ntiles_m = ntm;
  P.m_fastest = 0;   // measured on MI355X: n-fastest wins even for 3.7 MB weight slices (1038 vs 928 TFLOP/s)
  P.nblocks = P.ntiles_n * ntm;
  // stages + (mean, rstd) rows + (bias, ln_colsum) columns
  const size_t smem = 2 * (size_t)(BM + BN) * 128 + BM * 2 * sizeof(float) + 2 * (((BN + 63) / 64) * 64) * sizeof(float);
  static unsigned long long attr_done = 0;
  if (lr_attr_needed(&attr_done)) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_conv_kernel<BN, MODE, T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)smem);
  }
  hipLaunchKernelGGL((gemm_conv_kernel<BN, MODE, T>), dim3(P.nblocks, P.splits), dim3(GEMM_THREADS), smem, st, P);
  return lr_launch_status();
}

// dtype dispatch (fp16 | bf16; the erf-GELU mode of the text tower is fp16 only)
template <int BM2, int NW, int BN, int WMW, int NSTAGE, int MODE>
static int launch_pipe(const GemmParams& P, hipStream_t st) {
  if constexpr (MODE != 2) { if (P.bf16) return launch_pipe_t<BM2, NW, BN, WMW, NSTAGE, MODE, bf16>(P, st); }
  return launch_pipe_t<BM2, NW, BN, WMW, NSTAGE, MODE, f16>(P, st);
}
template <int BN, int WMW, int NSTAGE, int MODE>
static int launch_gemm256(const GemmParams& P, hipStream_t st) { return launch_pipe<256, 8, BN, WMW, NSTAGE, MODE>(P, st); }
template <int BN, int MODE>
static int launch_gemm(const GemmParams& P, hipStream_t st) {
  if constexpr (MODE != 2) { if (P.bf16) return launch_gemm_t<BN, MODE, bf16>(P, st); }
  return launch_gemm_t<BN, MODE, f16>(P, st);
}

static int launch_reduce(const GemmParams& P, hipStream_t st) {
  const dim3 grid((P.N + 8 * RED_GROUPS - 1) / (8 * RED_GROUPS), (P.M + RED_ROWS - 1) / RED_ROWS);
  if (grid.y > 65535) return LR_E_UNSUPPORTED;
  if (P.bf16) hipLaunchKernelGGL(splitk_reduce_kernel<bf16>, grid, dim3(RED_ROWS * RED_GROUPS), 0, st, P);
  else hipLaunchKernelGGL(splitk_reduce_kernel<f16>, grid, dim3(RED_ROWS * RED_GROUPS), 0, st, P);
  return lr_launch_status();
}

// split-K heuristic: only when the tile grid cannot fill the chip (256 CUs x 2 resident blocks) and K is long.
// geglu: the GEGLU / erf-GELU epilogue modes, which are not instantiated for the 160-column tiles
static void choose_tile(int M, int N, int geglu, int* tm, int* tn) {
  // explicit requests win; otherwise: the 256-row 8-wave kernel for large M when BN divides N, else the 128-row one
  if (*tm == 0 && *tn == 0) {
    // static fallback (the Python front end normally autotunes): the 256-row kernel needs >= ~1 block per CU
    const int t160 = ((M + 255) / 256) * ((N + 159) / 160), t128 = ((M + 255) / 256) * ((N + 127) / 128);
    if (!geglu && N % 160 == 0 && t160 >= 224) { *tm = 256; *tn = 160; }
    else if (!geglu && N % 128 == 0 && t128 >= 224) { *tm = 256; *tn = 128; }
    else { *tm = 128; *tn = (N % 128 == 0) ? 128 : 64; }
  } else if (*tm == 0) {
    *tm = 128;
  } else if (*tn == 0) {
    if (*tm == 256) *tn = (!geglu && N % 160 == 0) ? 160 : 128;
    else *tn = (N % 128 == 0) ? 128 : 64;
  }
}

static int choose_splits(int M, int N, int K, int tm, int tn, int geglu, int stages) {
  if (geglu) return 1;
  const int tiles = ((M + tm - 1) / tm) * ((N + tn - 1) / tn);
  const int nk = K / BK;
  if (stages == LR_PIPE_HALO) {                               // the halo-tile conv splits by whole 64-channel chunks (9 K-steps each)
    const int t_ = ((M + 255) / 256) * ((N + tn - 1) / tn), chunks = K / 576;
    if (t_ * 10 > 256 * 6 || chunks < 8) return 1;
    int s_ = (256 + t_ / 2) / t_;
    if (s_ > 8) s_ = 8;
    if (s_ > chunks / 2) s_ = chunks / 2;
    return s_ < 1 ? 1 : s_;
  }
  const int slots = (tm == 256 || stages == 4) ? 256 : 512;   // resident blocks on the chip
  if (tiles * 10 > slots * 6 || nk < 32) return 1;   // > 60 % of the resident slots filled: do not split
  int s = (slots + tiles / 2) / tiles;               // round to the nearest whole number of waves
  if (s > 8) s = 8;
  if (s > nk / 8) s = nk / 8;
  return s < 1 ? 1 : s;
}

// pipeline variant (lr_gemm_args.pipe) of the instance that serves tile (tm, tn): for 128-row tiles 2 = the 4-wave 2-stage
// kernel (several blocks per CU cover each other's waits) and, for tn = 128 | 160, LR_PIPE_W8_DEEP (4: 8 waves, 4-stage ring,
// one block per CU, loads three K-steps ahead); for 256-row tiles the ring depth of the only instance.
// (An 8-wave 2-stage variant with two blocks per CU was measured too: 2-3 us faster on the short-K linears in isolation,
// no difference in the step -- not kept.)
static int choose_stages(int tm, int tn, int stages) {
  if (stages == LR_PIPE_HALO && tm == 256 && (tn == 160 || tn == 320)) return LR_PIPE_HALO;
  if (tm == 128) return (stages == 4 && (tn == 128 || tn == 160)) ? 4 : 2;
  return (tn == 128 || tn == 160) ? 3 : 2;
}

// waves along N of the instance that serves tile (tm, tn): each writes one (sum, sumsq) partial per row
static int tile_wnw(int tm, int tn, int geglu) {
  if (tm == 128) return 2;
  if (tn == 256) return 4;
  if (tn == 320) return geglu ? 2 : 4;
  return 2;   // 256 x {128, 160}: 4 x 2 waves
}

extern "C" int lr_gemm_plan(const lr_gemm_args* a, int32_t* plan) {
  if (!a || !plan) return LR_E_ARG;
  const int M = a->B * a->H * a->W;
  const int K = a->taps * (a->C1 + (a->p2 ? a->C2 : 0)) + (a->skip1 ? a->Cs1 + (a->skip2 ? a->Cs2 : 0) : 0);
  int tn = a->tile_n, tm = a->tile_m;
  choose_tile(M, a->N, a->geglu != 0, &tm, &tn);
  plan[0] = tm; plan[1] = tn;
  plan[2] = a->splits ? a->splits : choose_splits(M, a->N, K, tm, tn, a->geglu == 1, a->pipe);
  plan[3] = choose_stages(tm, tn, a->pipe);
  return 0;
}

// rows per wave tile of the instance that serves tile (tm, tn): the row-block size of gn_stats_out
static int tile_wave_rows(int tm, int tn, int geglu, int stages = 0) {
  if (tm == 128) return choose_stages(tm, tn, stages) == 4 ? 32 : 64;     // 8-wave instance: 4 x 2 waves, wave tile 32 x BN/2
  if (tn == 256) return 128;
  if (tn == 320) return geglu ? 64 : 128;
  return 64;   // 256 x {128, 160}: 4 x 2 waves
}

extern "C" int lr_gemm_gn_rows(const lr_gemm_args* a) {
  if (!a) return 0;
  int tn = a->tile_n, tm = a->tile_m;
  const int M = a->B * a->H * a->W;
  choose_tile(M, a->N, a->geglu != 0, &tm, &tn);
  const int K = a->taps * (a->C1 + (a->p2 ? a->C2 : 0)) + (a->skip1 ? a->Cs1 + (a->skip2 ? a->Cs2 : 0) : 0);
  const int splits = a->splits ? a->splits : choose_splits(M, a->N, K, tm, tn, a->geglu == 1, a->pipe);
  if (splits > 1) return RED_ROWS;     // the statistics come out of the split-K reduce kernel
  return tile_wave_rows(tm, tn, a->geglu == 1, a->pipe);
}

// chunks per sample of gn_group_out for this call (0: the plan cannot produce per-group sums -- tile width not a whole number
// of groups, tiles straddling samples, GEGLU): rows per chunk = the tile's rows (32 behind a split-K reduce)
static int gn_group_chunks(const lr_gemm_args* a, int tm, int tn, int splits) {
  if (a->geglu || a->N % 32) return 0;
  const int cg = a->N / 32;
  const int hw = a->gn_hw > 0 ? a->gn_hw : a->H * a->W;
  const int M = a->B * a->H * a->W;
  if (hw <= 0 || M % hw) return 0;
  const int rows = splits > 1 ? RED_ROWS : tm;
  const int width = splits > 1 ? 8 * RED_GROUPS : tn;
  if (width % cg || hw % rows) return 0;
  return hw / rows;
}

extern "C" int lr_gemm_gn_group_chunks(const lr_gemm_args* a) {
  if (!a) return 0;
  int tn = a->tile_n, tm = a->tile_m;
  const int M = a->B * a->H * a->W;
  choose_tile(M, a->N, a->geglu != 0, &tm, &tn);
  const int K = a->taps * (a->C1 + (a->p2 ? a->C2 : 0)) + (a->skip1 ? a->Cs1 + (a->skip2 ? a->Cs2 : 0) : 0);
  const int splits = a->splits ? a->splits : choose_splits(M, a->N, K, tm, tn, a->geglu == 1, a->pipe);
  return gn_group_chunks(a, tm, tn, splits);
}

extern "C" int lr_gemm_stats_parts(const lr_gemm_args* a) {
  if (!a) return 0;
  int tn = a->tile_n, tm = a->tile_m;
  choose_tile(a->B * a->H * a->W, a->N, a->geglu != 0, &tm, &tn);
  return ((a->N + tn - 1) / tn) * tile_wnw(tm, tn, a->geglu == 1);
}

extern "C" int64_t lr_gemm_workspace_bytes(const lr_gemm_args* a) {
  if (!a) return 0;
  const int M = a->B * a->H * a->W;
  const int K = a->taps * (a->C1 + (a->p2 ? a->C2 : 0)) + (a->skip1 ? a->Cs1 + (a->skip2 ? a->Cs2 : 0) : 0);
  int tn = a->tile_n, tm = a->tile_m;
  choose_tile(M, a->N, a->geglu != 0, &tm, &tn);
  int splits = a->splits ? a->splits : choose_splits(M, a->N, K, tm, tn, a->geglu == 1, a->pipe);
  return splits > 1 ? (int64_t)splits * M * a->N * (int64_t)sizeof(float) : 0;
}

// Arrival counters of the in-launch split-K reduce: 16 slots x 2048 tiles, zero at module load and left zero by every launch
// (self-resetting protocol, gemm_common.h).  Consecutive launches rotate through the slots, so two launches that overlap on the
// device (graph branches, other streams) do not share counters unless 16 of them are in flight at once.
static __device__ unsigned lr_sk_counters[16][2048];
static unsigned* sk_counter_slot() {
  static unsigned* base[64] = {nullptr};
  static std::atomic<unsigned> next{0};      // (host threads may launch concurrently)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  dev &= 63;
  if (!base[dev]) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(lr_sk_counters)) != hipSuccess) return nullptr;
    base[dev] = (unsigned*)p;
  }
  return base[dev] + (size_t)(next.fetch_add(1, std::memory_order_relaxed) & 15) * 2048;
}

extern "C" int lr_gemm_splitk_timeouts(void) {
  unsigned v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(lr_sk_error), sizeof(v)) != hipSuccess) return -1;
  const unsigned h = lr_halo_sk_timeouts();
  return h == 0xFFFFFFFFu ? -1 : (int)(v + h);
}

extern "C" int lr_gemm_conv_f16(const lr_gemm_args* a, lr_stream_t s) {
  if (!a || !a->p1 || !a->wt || !a->out) return LR_E_ARG;
  GemmParams P;
  P.p1 = (const f16*)a->p1; P.C1 = a->C1;
  P.p2 = (const f16*)a->p2; P.C2 = a->p2 ? a->C2 : 0;
  P.c16 = (P.C1 == 16 && P.C2 == 0 && a->taps == 9 && a->stride == 1 && a->up == 0 && !a->asym) ? 1 : 0;
  if (P.C1 <= 0 || (!P.c16 && P.C1 % 64) || P.C2 % 64) return LR_E_ALIGN;
  if (a->taps != 1 && a->taps != 9) return LR_E_UNSUPPORTED;
  if (a->stride != 1 && a->stride != 2) return LR_E_UNSUPPORTED;
  if (a->up < 0 || a->up > 2) return LR_E_UNSUPPORTED;
  if (a->B <= 0 || a->H <= 0 || a->W <= 0 || a->Hs <= 0 || a->Ws <= 0 || a->N <= 0) return LR_E_ARG;
  P.H = a->H; P.W = a->W; P.Hs = a->Hs; P.Ws = a->Ws;
  P.taps = a->taps; P.stride = a->stride; P.up = a->up;
  P.zins = (a->up == 2) ? 1 : 0;      // up == 2: zero-insertion upsample (dgrad of a stride-2 conv)
  if (P.zins) P.up = 1;
  P.pad = a->asym ? 0 : 1;   // asym: F.pad(x, (0,1,0,1)) + conv padding 0 (VAE Downsample)
  P.wt = (const f16*)a->wt; P.N = a->N; P.bias = a->bias;
  P.M = a->B * a->H * a->W;
  P.p3 = (const f16*)a->skip1; P.C3 = a->skip1 ? a->Cs1 : 0;
  P.p4 = a->skip1 ? (const f16*)a->skip2 : nullptr; P.C4 = P.p4 ? a->Cs2 : 0;
  if (P.p3) {      // pointwise K extension (the ResBlock's skip_connection inside its last conv): see GemmParams.p3
    if (P.c16 || a->stride != 1 || a->up != 0 || a->asym || a->geglu || a->ln_stats || a->wt_bstride ||
        a->Hs != a->H || a->Ws != a->W)
      return LR_E_UNSUPPORTED;
    if (P.C3 <= 0 || P.C3 % 64 || P.C4 % 64 || (((uintptr_t)P.p3 | (uintptr_t)P.p4) & 15)) return LR_E_ALIGN;
  }
  P.K = P.c16 ? 192 : a->taps * (P.C1 + P.C2) + P.C3 + P.C4;      // c16: wt is [N][192] (k = tap * 16 + c, taps 9..11 zero)
  // 32-bit byte offsets in the gather (bit 31 marks out-of-range): every operand must stay below 2 GiB
  const int64_t lim = (int64_t)1 << 31;
  const int64_t src_rows = (int64_t)a->B * a->Hs * a->Ws;
  if (src_rows * P.C1 * 2 >= lim || src_rows * P.C2 * 2 >= lim || src_rows * P.C3 * 2 >= lim || src_rows * P.C4 * 2 >= lim ||
      (int64_t)P.N * P.K * 2 >= lim ||
      (int64_t)a->B * a->H * a->W >= lim / 2)
    return LR_E_UNSUPPORTED;
  P.rowvec = (const f16*)a->rowvec; P.ld_rowvec = a->ld_rowvec;
  P.resid = (const f16*)a->resid; P.ld_resid = a->ld_resid;
  P.out = (f16*)a->out; P.ld_out = a->ld_out;
  // the epilogue addresses out / resid / rowvec through buffer descriptors with 32-bit byte offsets (bit 31 = out of range)
  if ((int64_t)P.M * a->ld_out * 2 >= lim || (a->resid && (int64_t)P.M * a->ld_resid * 2 >= lim) ||
      (a->rowvec && (int64_t)a->B * a->ld_rowvec * 2 >= lim))
    return LR_E_UNSUPPORTED;
  if (a->geglu < 0 || a->geglu > 2) return LR_E_ARG;
  P.geglu = a->geglu == 1 ? 1 : 0;
  P.gelu = a->geglu == 2 ? 1 : 0;      // plain erf-GELU of (acc + bias [+ rowvec]), applied before the residual
  P.rows_per_batch = a->H * a->W;
  lr_udiv_magic((unsigned)(a->H * a->W), &P.hw_mul, &P.hw_sh);
  lr_udiv_magic((unsigned)a->W, &P.w_mul, &P.w_sh);
  const int N_out = P.geglu ? P.N / 2 : P.N;
  if (P.N % 8 || N_out % 8 || P.ld_out % 8 || (P.resid && P.ld_resid % 8) || (P.rowvec && P.ld_rowvec % 8))
    return LR_E_ALIGN;
  if (P.geglu && P.N % 32) return LR_E_ALIGN;
  if (((uintptr_t)P.p1 | (uintptr_t)P.p2 | (uintptr_t)P.wt | (uintptr_t)P.out | (uintptr_t)P.resid |
       (uintptr_t)P.rowvec | (uintptr_t)P.bias) & 15)
    return LR_E_ALIGN;
  int tn = a->tile_n, tm = a->tile_m;
  choose_tile(P.M, P.N, P.geglu || P.gelu, &tm, &tn);
  int splits = a->splits;
  if (P.c16) {      // only the pipelined 256-row instances gather 16-channel taps; three K-steps: never split
    if (tm != 256 || (a->splits > 1) || P.geglu || P.gelu || a->wt_bstride) return LR_E_UNSUPPORTED;
    splits = 1;
  }
  if (splits == 0) splits = choose_splits(P.M, P.N, P.K, tm, tn, P.geglu, a->pipe);
  if (splits > 1 && P.geglu) return LR_E_UNSUPPORTED;
  if (splits > 1) {
    const int64_t need = (int64_t)splits * P.M * P.N * (int64_t)sizeof(float);
    if (!a->workspace || a->workspace_bytes < need) {
      if (a->splits > 1) return LR_E_ARG;   // explicitly requested but no room
      splits = 1;
    }
  }
  P.splits = splits;
  P.ws = a->workspace;
  P.sk_cnt = nullptr;
  if (a->splitk_mode != 0 && a->splitk_mode != 1) return LR_E_ARG;
#ifndef LR_DEV_VARIANTS
  if (a->splitk_mode == 1) return LR_E_UNSUPPORTED;      // the in-launch reduce (measured slower, profiles/r05_splitk_inlaunch.txt) is compiled in developer builds only
#endif
  // LayerNorm fold (pointwise, single source, K = normalised width) and per-row output statistics
  P.ln_part = a->ln_stats; P.ln_cs = a->ln_colsum; P.ln_parts = a->ln_parts; P.ln_eps = a->ln_eps;
  P.ln_invc = 1.0f / (float)P.K;
  if (P.ln_part) {
    if (!P.ln_cs || P.ln_parts <= 0 || P.taps != 1 || P.C2 || splits > 1) return LR_E_ARG;
    if (((uintptr_t)P.ln_part & 7) || ((uintptr_t)P.ln_cs & 15)) return LR_E_ALIGN;
  }
#ifdef LR_GEMM_TRACE
  P.trace = g_trace;
#endif
  if (a->dtype != LR_DTYPE_F16 && a->dtype != LR_DTYPE_BF16) return LR_E_ARG;
  P.bf16 = a->dtype == LR_DTYPE_BF16;
#ifdef LR_GEMM_STAGGER
  P.stagger = LR_DEV("LR_GEMM_STAGGER", 0);
#endif
  if (P.bf16 && P.gelu) return LR_E_UNSUPPORTED;
  P.gs_out = a->gn_stats_out;
  if (P.gs_out && (P.geglu || ((uintptr_t)P.gs_out & 15))) return LR_E_ARG;
  P.gs_store = P.gs_out != nullptr;
  P.gp_out = a->gn_group_out; P.gp_cg = 1; P.gp_chunks = 0; P.gp_hw = 1;
  if (P.gp_out) {
    P.gp_chunks = gn_group_chunks(a, tm, tn, splits);
    if (P.gp_chunks <= 0 || ((uintptr_t)P.gp_out & 7) || a->stats_out) return LR_E_ARG;
    P.gp_cg = P.N / 32;
    P.gp_hw = a->gn_hw > 0 ? a->gn_hw : a->H * a->W;
  }
  // per-sample weights (GroupNorm of the SpatialTransformer folded into proj_in): pointwise, one tile = one sample, no split
  P.wt_pm = a->wt_pm ? 1 : 0;
  if (P.wt_pm && (a->wt_bstride || P.K % 64)) return LR_E_ARG;
  P.wt_bstride = a->wt_bstride; P.bias_bstride = a->bias_bstride;
  if (P.wt_bstride) {
    if (a->taps != 1 || splits > 1 || P.rows_per_batch % tm || P.wt_bstride < 0 || P.bias_bstride < 0 || (P.wt_bstride & 7)) return LR_E_ARG;
    if ((int64_t)a->B * P.wt_bstride * 2 >= lim) return LR_E_UNSUPPORTED;
  }
  P.st_out = a->stats_out;
  P.st_parts = ((P.N + tn - 1) / tn) * tile_wnw(tm, tn, P.geglu);
  if (P.st_out && (splits > 1 || ((uintptr_t)P.st_out & 7))) return LR_E_ARG;
  hipStream_t st = (hipStream_t)s;
  int rc;
  const int mode = P.geglu ? 1 : P.gelu ? 2 : 0;
  if (a->pipe != 0 && a->pipe != choose_stages(tm, tn, a->pipe)) return LR_E_UNSUPPORTED;
  const bool deep = tm == 128 && choose_stages(tm, tn, a->pipe) == 4;
#ifdef LR_DEV_VARIANTS
  // in-launch split-K reduce: the 8-wave instances with 160- / 320-column tiles, the whole grid resident at once (one block per CU)
  if (P.splits > 1 && a->splitk_mode == 1 && mode != 1 && (tn == 160 || tn == 320) && (tm == 256 || deep || a->pipe == LR_PIPE_HALO)) {
    const int64_t tiles = (int64_t)((P.M + tm - 1) / tm) * ((P.N + tn - 1) / tn);
    if (tiles * P.splits <= 256 && tiles <= 2048 && (int64_t)P.splits * P.M * P.N * 4 < lim && P.N % 8 == 0)
      P.sk_cnt = sk_counter_slot();
  }
#endif
  if (a->pipe == LR_PIPE_HALO) {      // 3x3 stride-1 conv with the input patch resident in LDS (conv_halo.hip)
    if (mode != 0 || a->asym) return LR_E_UNSUPPORTED;
    rc = lr_launch_conv_halo(P, tn, st);
    if (rc || P.splits == 1 || P.sk_cnt) return rc;
    return launch_reduce(P, st);
  }
  if (mode == 0) {
    if (deep && tn == 128) rc = launch_pipe<128, 8, 128, 4, 4, 0>(P, st);
    else if (deep && tn == 160) rc = launch_pipe<128, 8, 160, 4, 4, 0>(P, st);

    else if (tm == 128 && tn == 128) rc = launch_gemm<128, 0>(P, st);
    else if (tm == 128 && tn == 64) rc = launch_gemm<64, 0>(P, st);
    else if (tm == 128 && tn == 160) rc = launch_gemm<160, 0>(P, st);
    else if (tm == 256 && tn == 128) rc = launch_gemm256<128, 4, 3, 0>(P, st);
    else if (tm == 256 && tn == 160) rc = launch_gemm256<160, 4, 3, 0>(P, st);
    else if (tm == 256 && tn == 256) rc = launch_gemm256<256, 2, 2, 0>(P, st);                 // wave tile 128 x 64
    else if (tm == 256 && tn == 320) rc = launch_gemm256<320, 2, 2, 0>(P, st);                 // wave tile 128 x 80
    else return LR_E_UNSUPPORTED;
  } else if (mode == 1) {
    if (deep && tn == 128) rc = launch_pipe<128, 8, 128, 4, 4, 1>(P, st);
    else if (deep) return LR_E_UNSUPPORTED;
    else if (tm == 128 && tn == 128) rc = launch_gemm<128, 1>(P, st);
    else if (tm == 128 && tn == 64) rc = launch_gemm<64, 1>(P, st);
    else if (tm == 256 && tn == 128) rc = launch_gemm256<128, 4, 3, 1>(P, st);
    else if (tm == 256 && tn == 256) rc = launch_gemm256<256, 2, 2, 1>(P, st);
    else if (tm == 256 && tn == 320) rc = launch_gemm256<320, 4, 2, 1>(P, st);                 // wave tile 64 x 160: even TN
    else return LR_E_UNSUPPORTED;
  } else {   // erf-GELU epilogue (text tower MLP): the small-tile instances only
    if (deep) return LR_E_UNSUPPORTED;
    if (tm == 128 && tn == 128) rc = launch_gemm<128, 2>(P, st);
    else if (tm == 128 && tn == 64) rc = launch_gemm<64, 2>(P, st);
    else if (tm == 256 && tn == 128) rc = launch_gemm256<128, 4, 3, 2>(P, st);
    else return LR_E_UNSUPPORTED;
  }
  if (rc || P.splits == 1 || P.sk_cnt) return rc;
  return launch_reduce(P, st);
}
