// 3x3 stride-1 convolution on the gfx950 matrix cores with an LDS-resident input patch ("halo tile").
//
// replaces: the conv_nd(dims, ch, out, 3, padding=1) layers of ResBlock.in_layers / out_layers (reference
//           ldm/modules/diffusionmodules/openaimodel.py:200-231, 254-274) at the levels whose latent is a multiple of 16 x 16 pixels;
//           same arguments, epilogues and statistics outputs as lr_gemm_conv_f16 (this is its LR_PIPE_HALO instance).
//
// Why: gemm_conv_pipe_kernel gathers, for EVERY tap, the 128-byte channel slice of each shifted source pixel from L2 into LDS --
// nine L2 -> LDS copies of (nearly) the same pixels per 64-channel chunk, 256 x 128 B each, next to 9 x BN x 128 B of weights.  Its
// dominant instance (256 x 320) is co-limited by that fill stream (73.7 KB per K-step against ~10 TB/s chip-wide with the MFMAs running).
// Here the block owns a 16 x 16 PIXEL tile of one sample.  Per 64-channel chunk the (16 + 2) x (16 + 2) halo patch is copied ONCE
// (324 x 128 B = 41.5 KB instead of 9 x 32 KB) and the nine taps are shifted fragment reads of that patch; only the weights keep
// streaming per tap.  L2 -> LDS bytes per chunk of a 256 x 320 tile: 664 KB -> 410 KB; of a 256 x 160 tile: 472 KB -> 226 KB.
//
// * K order: chunk-major (chunk, tap) -- the weights stay in the packed [N][tap][Cin] K order, a K-step reads the 128-byte piece
//   k = tap * Cin + 64 chunk of every row.  With row-major weights [N][K] those pieces lie 2 K bytes apart and every CU of an XCD asks
//   its L2 for the same strided lines at the same time; piece-major weights [K / 64][N][64] (lr_gemm_args.wt_pm) make a tile's K-step
//   ONE contiguous run of BN x 128 bytes -- supported, bit-identical, and measured to make no difference here (the L2 serves both).  (gemm_conv_pipe_kernel walks (tap, chunk): the two kernels add the same products in a
//   different order, so they agree to fp32 rounding, not bit for bit; the plan is static per shape, reruns are bit-identical.)
// * Patch layout: pixel p = line * 18 + column owns the 128-byte LDS row p; the 16-byte slot of channel chunk c in row p is
//   c ^ (column & 6).  A fragment read is 16 CONSECUTIVE pixels (one output line segment shifted by the tap) at an arbitrary base, and
//   the 16-lane groups of ds_read_b128 mix two k-chunks (c, c ^ 1): with the swizzle on the 32-byte PAIR index every group touches 16
//   distinct 16-byte slots of the 256-byte bank row for every base (the GEMM's c ^ ((row >> 1) & 7) is conflict-free only for bases
//   that are multiples of 4; 18 is even, so row parity = column parity).  The key depends on the column only, so the eight line
//   segments of a wave differ by immediate offsets (18 rows) and a tap's ky by a scalar.  LDS-DMA writes lane-linear, so the
//   permutation sits on the per-lane source address.
// * Zero padding = out-of-range buffer offsets (hardware returns zeros); no branches in the loader.
// * Pipeline: ONE patch buffer + an NSTAGE-slot weight ring, one barrier per K-step placed in the middle of the step (see the main loop).
// * Epilogue: the register epilogue of the GEMM family (gemm_common.h) with row tile i = 16 pixels of image line i of the wave.
#include "gemm_common.h"

#ifndef HALO_EXP       // developer experiments (timing only, wrong results): 1 = never reload the patch, 3 = never reload the weights
#define HALO_EXP 0
#endif
#define HALO_PW 18          // patch width / height in pixels (16 + 2)
#define HALO_NPX 324
#define HALO_NQ 41          // LDS-DMA instructions per patch (8 pixel rows of 128 B each): 328 rows, the last 4 unused

// BN = 320: 2 x 4 waves, wave tile 8 lines x 80 columns (40 MFMA tiles), 2-slot weight ring  (level 0: N = 320 is one tile)
// BN = 160: 4 x 2 waves, wave tile 4 lines x 80 columns (20 MFMA tiles), 3-slot weight ring  (level 1: 16384 x 640 = 256 tiles)
template <int BN, int WMW, int NSTAGE, typename T>
__global__ __launch_bounds__(512) void conv_halo_kernel(const GemmParams P) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NW = 8, WNW = NW / WMW, TM = 16 / WMW, TN = BN / WNW / 16;
  constexpr int PATCH_BYTES = HALO_NQ * 1024;
  constexpr int B_BYTES = BN * 128;
  constexpr int NB_FULL = BN / (NW * 8);
  constexpr bool B_TAIL = (BN % (NW * 8)) != 0;
  constexpr int TAIL_WAVES = (BN % (NW * 8)) / 8;
  constexpr int PAR_LD = ((BN + 63) / 64) * 64;
  constexpr bool ROLL = TM * TN <= 20;      // rolling fragment schedule (needs a second fragment set's worth of registers in flight)
  static_assert(NSTAGE == 2 || NSTAGE == 3, "weight ring depth");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const patch = smem;
  char* const wring = smem + PATCH_BYTES;
  float* const par = reinterpret_cast<float*>(wring + NSTAGE * B_BYTES);

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w / WNW, wn = w % WNW;
  const int fr = lane & 15, fq = lane >> 4;
  const unsigned OOB = 0x80000000u;

  // XCD-aware bijective remap (as gemm_conv_pipe_kernel): XCD x owns a contiguous range of logical tiles, n-fastest inside
  const int xcd = blockIdx.x & 7;
  const int q8 = P.nblocks >> 3, r8 = P.nblocks & 7;
  const int bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
  [[maybe_unused]] const int lr_trace_tile = bid;
  const int tile_n = bid % P.ntiles_n, tile = bid / P.ntiles_n;      // tile: 16 x 16 pixel tiles numbered sample-major, line-major
  const int tiles_x = P.W >> 4, tps = (P.H >> 4) * tiles_x;
  const int smp = tile / tps, trem = tile - smp * tps;
  const int tyi = trem / tiles_x, txi = trem - tyi * tiles_x;
  const int y0 = tyi * 16, x0 = txi * 16;
  const int n0 = tile_n * BN;
  const int m_org = (smp * P.H + y0) * P.W + x0;

  // ---- patch loader: piece q = w + 8 k of the patch = LDS rows 8 q .. 8 q + 7; lane l fills slot l & 7 of row 8 q + (l >> 3) with channel
  // chunk (l & 7) ^ (column & 6) of the row's source pixel (out of the image / row >= 324: out-of-range offset -> zeros).  The addresses
  // are re-derived per chunk (~15 VALU per piece, 9 K-steps apart) instead of living in 6 registers through the MFMA loop.
  const int Ctot = P.C1 + P.C2;
  const int cpt1 = P.C1 >> 6, ncm = Ctot >> 6, cpt3 = P.C3 >> 6;
  const int nch = ncm + ((P.C3 + P.C4) >> 6);                        // chunks: the 3x3 part, then the pointwise extension (one tap each)
  const int nsteps = 9 * ncm + (nch - ncm);
  auto issue_patch = [&](const int ci) __attribute__((always_inline)) {
    const f16* src; int cs, ch;
    if (ci < cpt1) { src = P.p1; cs = P.C1; ch = ci; }
    else if (ci < ncm) { src = P.p2; cs = P.C2; ch = ci - cpt1; }
    else if (ci < ncm + cpt3) { src = P.p3; cs = P.C3; ch = ci - ncm; }
    else { src = P.p4; cs = P.C4; ch = ci - ncm - cpt3; }
    const __amdgpu_buffer_rsrc_t rsA = uniform_rsrc((const void*)src, (size_t)P.M * cs * 2);
    const unsigned coff = (unsigned)ch * 128u;
    int l = lane;
    asm volatile("" : "+v"(l));      // opaque: keeps the address arithmetic below inside the loop (not hoisted into registers)
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int q = w + 8 * k;
      if (q < HALO_NQ) {      // wave-uniform (only wave 0 has a sixth piece)
        const int px = q * 8 + (l >> 3);
        const int pl = (px * 3641) >> 16, pc = px - pl * HALO_PW;        // px / 18 for px < 328
        const int y = y0 - 1 + pl, x = x0 - 1 + pc;
        const bool ok = px < HALO_NPX && (unsigned)y < (unsigned)P.H && (unsigned)x < (unsigned)P.W;
        const int c = (l & 7) ^ (pc & 6);
        const unsigned vo = ok ? (unsigned)(((smp * P.H + y) * P.W + x) * cs + c * 8) * 2u : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(patch + q * 1024), 16, vo, coff, 0, 0);
      }
    }
  };

  // ---- weight loader (as gemm_conv_pipe_kernel): instruction i covers rows (i * NW + w) * 8 + lane / 8 of the tile
  const __amdgpu_buffer_rsrc_t rsB = uniform_rsrc((const void*)P.wt, (size_t)P.N * P.K * 2);
  const __amdgpu_buffer_rsrc_t rsZ = uniform_rsrc((const void*)P.wt, 0);
  __amdgpu_buffer_rsrc_t rsW = rsB;
  // (N is a multiple of BN here, so no row is out of range; rows 64 i + .. of the tile differ by a scalar offset: ONE address register.
  //  The swizzle key (row >> 1) & 7 = 4 (w & 1) + (lane >> 4) does not depend on i.)
  const unsigned wvo0 = (unsigned)(((size_t)(n0 + w * 8 + (lane >> 3)) * (P.wt_pm ? 64 : P.K) + (((lane & 7) ^ (((w & 1) << 2) + (lane >> 4))) & 7) * 8) * 2);
  const unsigned wrow64 = P.wt_pm ? 64u * 128u : (unsigned)P.K * 128u;      // bytes between weight rows n and n + 64
  // byte offset of the K-step (chunk ci, tap) inside a weight row
  auto wkoff = [&](const int ci, const int tap) -> unsigned {
    // piece-major weights [K / 64][N][64] (lr_gemm_args.wt_pm): the K-step's piece index is its position in the packed K order
    if (P.wt_pm) return (unsigned)(ci < ncm ? tap * ncm + ci : 9 * ncm + (ci - ncm)) * (unsigned)P.N * 128u;
    return (unsigned)((ci < ncm ? tap * Ctot + ci * 64 : 9 * Ctot + (ci - ncm) * 64) * 2);
  };
  auto issue_weights = [&](const int slot, const unsigned koff) __attribute__((always_inline)) {
    char* Bs = wring + slot * B_BYTES;
    if (B_TAIL && w < TAIL_WAVES)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lptr_t)(Bs + ((NB_FULL * NW + w) * 8) * 128), 16, wvo0, koff + NB_FULL * wrow64, 0, 0);
#pragma unroll
    for (int i = 0; i < NB_FULL; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lptr_t)(Bs + ((i * NW + w) * 8) * 128), 16, wvo0, koff + i * wrow64, 0, 0);
  };
  // piece i of a K-step's weights (i < NB_FULL: rows 64 i .. of the tile; i == NB_FULL: the tail rows, first waves only)
  static_assert(NB_FULL + (B_TAIL ? 1 : 0) <= TM, "one LDS-DMA piece per line group");
  auto issue_weight_piece = [&](const int slot, const unsigned koff, const int i) __attribute__((always_inline)) {
    char* Bs = wring + slot * B_BYTES;
    if (i < NB_FULL)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lptr_t)(Bs + ((i * NW + w) * 8) * 128), 16, wvo0, koff + i * wrow64, 0, 0);
    else if (i == NB_FULL && B_TAIL && w < TAIL_WAVES)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lptr_t)(Bs + ((NB_FULL * NW + w) * 8) * 128), 16, wvo0, koff + NB_FULL * wrow64, 0, 0);
  };
  auto advance = [&](int& ci, int& tap) __attribute__((always_inline)) {
    const int nt = ci < ncm ? 9 : 1;
    if (++tap == nt) { tap = 0; ++ci; }
  };

  // ---- fragments: activation = 16 consecutive patch pixels of line (wave line i + ky) starting at column kx; weights as the GEMM
  // (the swizzle keys do not depend on the line / the weight tile: one address per operand and k-half, everything else immediates)
  const int pb0 = wm * TM * HALO_PW + fr;
  auto x_addr = [&](const int ks, const int ky, const int kx) -> const char* {
    const int kc = ks * 4 + fq;
    return patch + (pb0 + kx) * 128 + ((kc ^ ((fr + kx) & 6)) << 4) + ky * (HALO_PW * 128);
  };
  auto rx = [&](const char* xs, const int i) -> vec8<T> { return *reinterpret_cast<const vec8<T>*>(xs + i * (HALO_PW * 128)); };
  auto w_addr = [&](const int slot, const int ks) -> const char* {
    const int kc = ks * 4 + fq;
    return wring + slot * B_BYTES + fr * 128 + ((kc ^ ((fr >> 1) & 7)) << 4);
  };
  auto rw = [&](const char* ws, const int j) -> vec8<T> {
    return *reinterpret_cast<const vec8<T>*>(ws + weight_tile<TN, WNW, false>(wn, j) * (16 * 128));
  };
  f32x4 acc[TN][TM];
  using std::integral_constant;

  LR_STAMP(0);
  stage_params<BN, PAR_LD>(P, par, n0, w, lane, 0);
  issue_patch(0);
  int wci = 0, wtap = 0;                 // K-step whose weights are issued next
#pragma unroll
  for (int sidx = 0; sidx < NSTAGE; ++sidx) {      // the whole ring up front
    rsW = sidx < nsteps ? rsB : rsZ;
    issue_weights(sidx, wkoff(wci, wtap));
    advance(wci, wtap);
  }
  LR_STAMP(1);
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  LR_STAMP(2);

  // ONE barrier per K-step, in the MIDDLE of the step, and a rolling fragment schedule (one fragment set; the MFMAs run line-major, so
  // the x fragment of line i is dead after TN MFMAs and its registers take what the next half step needs while the matrix pipe works):
  //   half A (k 0..31):  [TN MFMAs of line i | read x(i) of half B] ...; in the last line every MFMA is followed by the reload of the
  //                      weight fragment it just used (half B's).  After it nobody reads the step's weight slot or (at the last tap
  //                      of a chunk) the patch any more.
  //   barrier:           every wave's fragment reads are in registers and its LDS-DMA of step s + 1 (issued one whole step ago) has
  //                      landed -> the slot of step s is refilled with step s + NSTAGE; behind the last tap the patch is reloaded too.
  //   half B (k 32..63): [TN MFMAs of line i | read x(i) of the NEXT tap, half A -- the patch does not change inside a chunk]; in the
  //                      last line the weight fragments of step s + 1, half A (landed before the barrier above).
  // So a K-step's weights have a full step to arrive (the 2-slot ring of gemm_conv_pipe_kernel<256, 8, 320> issues them during the
  // first half of the step that precedes their use and waits for them at its end), the step boundary has no barrier and no bulk
  // fragment read, and only a patch reload (every 9th step) costs a second barrier.
  int ci = 0, tap = 0, cur = 0;
  vec8<T> xf[TM], wf[TN];
  if constexpr (ROLL) {
    const char* xs = x_addr(0, ncm > 0 ? 0 : 1, ncm > 0 ? 0 : 1);
    const char* ws = w_addr(0, 0);
#pragma unroll
    for (int j = 0; j < TN; ++j) wf[j] = rw(ws, j);
#pragma unroll
    for (int i = 0; i < TM; ++i) xf[i] = rx(xs, i);
  }
  for (int s = 0; s < nsteps; ++s) {
    const bool main_part = ci < ncm;
    const int ky = main_part ? (tap * 11) >> 5 : 1, kx = main_part ? tap - 3 * ky : 1;      // (tap * 11) >> 5 = tap / 3 for tap < 12
    const bool last_tap = !main_part || tap == 8;
    const bool pf_w = s + NSTAGE < nsteps;              // weights to prefetch in this step
    const bool pf_p = last_tap && ci + 1 < nch;         // patch to reload in this step
    const int nxt = cur + 1 == NSTAGE ? 0 : cur + 1;
    if constexpr (ROLL) {
      // first tap of the following step (x fragments are prefetched one half step ahead; behind a patch reload they are read again)
      const int ntap = last_tap ? 0 : tap + 1;
      const bool nmain = last_tap ? ci + 1 < ncm : true;
      const int nky = nmain ? (ntap * 11) >> 5 : 1, nkx = nmain ? ntap - 3 * nky : 1;
      const char* ws1 = w_addr(cur, 1);
      const char* xs1 = x_addr(1, ky, kx);
      // ---- half A  (sched_barrier(0) after every line group pins the order [TN MFMAs | refill])
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[j][i] = lr_mfma16(wf[j], xf[i], acc[j][i]);
          if (i == TM - 1) { wf[j] = rw(ws1, j); __builtin_amdgcn_sched_barrier(0); }
        }
        xf[i] = rx(xs1, i);
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- the step's barrier
      if (NSTAGE == 3) { if (B_TAIL && w < TAIL_WAVES) wait_vmcnt<NB_FULL + 1>(); else wait_vmcnt<NB_FULL>(); }
      else wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // (past the end of K the weights go through a zero-length descriptor: same instruction count in every step, constant waits)
      rsW = (pf_w && HALO_EXP != 3) ? rsB : rsZ;
      const unsigned koff = wkoff(wci, wtap);
      advance(wci, wtap);
      if (pf_p && HALO_EXP != 1) issue_patch(ci + 1);
      const char* ws0n = w_addr(nxt, 0);
      const char* xsn = x_addr(0, nky, nkx);
      // ---- half B  ([TN MFMAs | refill | one LDS-DMA piece])
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[j][i] = lr_mfma16(wf[j], xf[i], acc[j][i]);
          if (i == TM - 1) { wf[j] = rw(ws0n, j); __builtin_amdgcn_sched_barrier(0); }
        }
        xf[i] = rx(xsn, i);
        issue_weight_piece(cur, koff, i);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (pf_p) {      // the prefetched x fragments came from the patch that was being replaced: wait for the new one and read them again
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < TM; ++i) xf[i] = rx(xsn, i);
      }
    } else {
      // 40 accumulator tiles leave no room for fragments in flight: bulk reads per half step, the same mid-step barrier
      auto read_frags = [&](const int ks) __attribute__((always_inline)) {
        const char* xs = x_addr(ks, ky, kx);
        const char* ws = w_addr(cur, ks);
#pragma unroll
        for (int i = 0; i < TM; ++i) xf[i] = rx(xs, i);
#pragma unroll
        for (int j = 0; j < TN; ++j) wf[j] = rw(ws, j);
      };
      auto mma = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int i = 0; i < TM; ++i) acc[j][i] = lr_mfma16(wf[j], xf[i], acc[j][i]);
      };
      read_frags(0);
      __builtin_amdgcn_sched_barrier(0);
      mma();
      __builtin_amdgcn_sched_barrier(0);
      read_frags(1);
      wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      rsW = (pf_w && HALO_EXP != 3) ? rsB : rsZ;
      const unsigned koff = wkoff(wci, wtap);
      advance(wci, wtap);
      if (pf_p && HALO_EXP != 1) issue_patch(ci + 1);
#pragma unroll
      for (int i = 0; i <= NB_FULL; ++i) issue_weight_piece(cur, koff, i);
      mma();
#pragma unroll
      for (int g = 0; g < NB_FULL; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x8, (TM * TN) / NB_FULL, 0);
        __builtin_amdgcn_sched_group_barrier(0x10, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (pf_p) {      // the next chunk's patch has to land before its first fragment reads
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
      }
    }
    cur = nxt;
    advance(ci, tap);
  }
  // (the epilogue reuses the patch buffer for its column sums: every wave's trailing zero-length LDS-DMA and fragment reads are done)
  wait_vmcnt<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  LR_STAMP(3);
  LR_STAMP(4);

  // ---- epilogue straight from the accumulators (the loop ended with vmcnt(0) + barrier: the patch buffer is free for the sums)
  float* gsl = reinterpret_cast<float*>(smem);
  const bool gp = P.gp_out != nullptr;
  epilogue_units<TM, TN, 0, PAR_LD, WNW, T, true>(P, acc, m_org + wm * TM * P.W, n0, wn, lane, par, par, tile_n * WNW + wn,
                                                  gsl + wm * (BN * 2), tile * WMW + wm);
  if (gp) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    gn_group_reduce<BN, WMW, 256>(P, gsl, tile * 256, n0, t);
  }
  LR_STAMP(5);
#ifdef LR_GEMM_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  LR_STAMP(6);
#endif
#endif  // __HIP_DEVICE_COMPILE__
}

template <int BN, int WMW, int NSTAGE, typename T>
static int launch_halo_t(const GemmParams& P0, hipStream_t st) {
  GemmParams P = P0;
  P.ntiles_n = (P.N + BN - 1) / BN;
  P.ntiles_m = P.M / 256;
  P.m_fastest = 0;
  P.nblocks = P.ntiles_n * P.ntiles_m;
  const size_t smem = (size_t)HALO_NQ * 1024 + NSTAGE * (size_t)BN * 128 + 2 * (((BN + 63) / 64) * 64) * sizeof(float);
  static unsigned long long attr_done = 0;
  if (lr_attr_needed(&attr_done)) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo_kernel<BN, WMW, NSTAGE, T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)smem);
  }
  hipLaunchKernelGGL((conv_halo_kernel<BN, WMW, NSTAGE, T>), dim3(P.nblocks, 1), dim3(512), smem, st, P);
  return lr_launch_status();
}

// the LR_PIPE_HALO instances of lr_gemm_conv_f16 (called from gemm_conv.hip after its argument checks)
int lr_launch_conv_halo(const GemmParams& P, int tile_n, hipStream_t st) {
  if (P.taps != 9 || P.stride != 1 || P.up || P.zins || P.pad != 1 || P.c16 || P.splits != 1 || P.geglu || P.gelu || P.ln_part ||
      P.wt_bstride || P.st_out || (P.H & 15) || (P.W & 15) || P.Hs != P.H || P.Ws != P.W || P.N % tile_n)
    return LR_E_UNSUPPORTED;
  if (tile_n == 320) return P.bf16 ? launch_halo_t<320, 2, 2, bf16>(P, st) : launch_halo_t<320, 2, 2, f16>(P, st);
  if (tile_n == 160) return P.bf16 ? launch_halo_t<160, 4, 3, bf16>(P, st) : launch_halo_t<160, 4, 3, f16>(P, st);
  return LR_E_UNSUPPORTED;
}
