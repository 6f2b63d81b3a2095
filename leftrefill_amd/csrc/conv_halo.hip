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
// * K order: chunk-major (chunk, tap) -- the weights stay in the packed [N][tap][Cin] layout, a K-step reads the 128-byte piece
//   k = tap * Cin + 64 chunk of every row.  (gemm_conv_pipe_kernel walks (tap, chunk): the two kernels add the same products in a
//   different order, so they agree to fp32 rounding, not bit for bit; the plan is static per shape, reruns are bit-identical.)
//   Piece-major weights [K / 64][N][64] (lr_gemm_args.wt_pm) are accepted and bit-identical; measured to make no difference here.
// * Patch layout: pixel p = line * 18 + column owns the 128-byte LDS row p; the 16-byte slot of channel chunk c in row p is
//   c ^ (column & 6).  A fragment read is 16 CONSECUTIVE pixels (one output line segment shifted by the tap) at an arbitrary base, and
//   the 16-lane groups of ds_read_b128 mix two k-chunks (c, c ^ 1): with the swizzle on the 32-byte PAIR index every group touches 16
//   distinct 16-byte slots of the 256-byte bank row for every base (the GEMM's c ^ ((row >> 1) & 7) is conflict-free only for bases
//   that are multiples of 4; 18 is even, so row parity = column parity).  The key depends on the column only, so the eight line
//   segments of a wave differ by immediate offsets (18 rows) and a tap's ky by a scalar.  LDS-DMA writes lane-linear, so the
//   permutation sits on the per-lane source address.
// * Zero padding = out-of-range buffer offsets (hardware returns zeros); no branches in the loader.
// * Pipeline: ONE patch buffer + an NSTAGE-slot weight ring.  The weights of step s + NSTAGE - 1 are issued between the MFMAs of the first
//   half of step s; at the last tap of a chunk the patch is dead once every wave has read its second-half fragments, the next chunk's
//   patch is issued behind a mid-step barrier and lands under the second half's MFMAs.
//   Measured and NOT kept (profiles/r05_conv_halo_proto.txt): (a) a rolling fragment schedule (line-major MFMAs, every fragment
//   register refilled for the next half step as soon as its line is done; no bulk "read 13, multiply 40"): 102.3 vs 100.4 us at
//   BN = 320, 107.6 vs 110.6 at BN = 160; (b) ONE barrier per K-step in the middle of the step with a whole step for the weights to
//   land: 109 vs 103 us / 114 vs 112 us.  Ablations of this kernel (65536 x 320 x 2880, 100 us): weight stream with zero-length
//   descriptors (instructions issued, no bytes) 88 us, no patch reload 96 us, MFMAs + barriers only 81.6 us, without the barriers 81.1 us
//   -- what separates the kernel from its matrix-pipe time is bytes moved, not where the waits sit.
// * Epilogue: the register epilogue of the GEMM family (gemm_common.h) with row tile i = 16 pixels of image line i of the wave.
#include "gemm_common.h"

#ifndef HALO_EXP       // developer experiments (timing only, wrong results): 1 = never reload the patch, 3 = never reload the weights
#define HALO_EXP 0
#endif
#define HALO_PW 18          // patch width / height in pixels (16 + 2)
#define HALO_NQ 45          // LDS-DMA instructions per patch buffer (8 pixel rows of 128 B each): 41 in use for a 16-line image tile
                            // (18 x 18 = 324 rows), 45 for the two-segment tile of 8-line images (2 x 10 x 18 = 360 rows)

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
  int tile_n, tile;      // tile: 16 x 16 pixel tiles numbered sample-major, line-major
  tile_order(P, bid, tile, tile_n);
  // 8-line images (the 8 x 16 level): a tile is lines 0..7 of TWO consecutive samples -- in token order exactly a 16-line image of
  // the pair, so the output side needs nothing; the patch holds the two samples' 10 x 18 halo segments one after the other
  // (zero rows between them: each segment has its own padding) and a wave's lines lie in one segment
  const bool seg = P.H == 8;
  const int Hv = seg ? 16 : P.H;                                      // height of the (virtual) image the tile grid covers
  const int npx = seg ? 2 * 10 * HALO_PW : HALO_PW * HALO_PW, nq = (npx + 7) >> 3;
  const int tiles_x = P.W >> 4, tps = (Hv >> 4) * tiles_x;
  const int smp = tile / tps, trem = tile - smp * tps;
  const int tyi = trem / tiles_x, txi = trem - tyi * tiles_x;
  const int y0 = tyi * 16, x0 = txi * 16;
  const int n0 = tile_n * BN;
  const int m_org = (smp * Hv + y0) * P.W + x0;

  // ---- patch loader state: source pixel of each of this lane's (up to 6) patch rows, -1 = outside the image (or row >= 324)
  // (bit 30 of a valid entry's complement is free: the channel chunk of the lane's LDS slot, (lane & 7) ^ (column & 6), rides in bits 28-30)
  int pix[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int px = (w + 8 * k) * 8 + (lane >> 3);
    const int pl = (px * 3641) >> 16, pc = px - pl * HALO_PW;        // px / 18 for px < 368
    const int sg = seg ? (pl * 26) >> 8 : 0;                          // pl / 10 for pl < 20: the sample of the pair
    const int y = (seg ? pl - 10 * sg : y0 + pl) - 1, x = x0 - 1 + pc;
    const bool ok = px < npx && (unsigned)y < (unsigned)P.H && (unsigned)x < (unsigned)P.W;
    const int c = (lane & 7) ^ (pc & 6);
    pix[k] = ok ? (((((seg ? 2 * smp + sg : smp) * P.H + y) * P.W + x)) | (c << 28)) : -1;
  }
  const int Ctot = P.C1 + P.C2;
  const int cpt1 = P.C1 >> 6, ncm = Ctot >> 6, cpt3 = P.C3 >> 6;
  const int nch_all = ncm + ((P.C3 + P.C4) >> 6);                    // chunks: the 3x3 part, then the pointwise extension (one tap each)
  // split-K (blockIdx.y): whole chunks [c_begin, nch) per slice, fp32 partial tiles to P.ws, epilogue in splitk_reduce_kernel
  const int c_per = (nch_all + P.splits - 1) / P.splits;
  const int c_begin = min((int)blockIdx.y * c_per, nch_all);         // (a slice beyond the last chunk is empty: nsteps = 0)
  const int nch = min(nch_all, c_begin + c_per);
  const int nsteps = (min(nch, ncm) - min(c_begin, ncm)) * 9 + (max(nch, ncm) - max(c_begin, ncm));
  auto issue_patch = [&](const int ci) __attribute__((always_inline)) {
    const f16* src; int cs, ch;
    if (ci < cpt1) { src = P.p1; cs = P.C1; ch = ci; }
    else if (ci < ncm) { src = P.p2; cs = P.C2; ch = ci - cpt1; }
    else if (ci < ncm + cpt3) { src = P.p3; cs = P.C3; ch = ci - ncm; }
    else { src = P.p4; cs = P.C4; ch = ci - ncm - cpt3; }
    const __amdgpu_buffer_rsrc_t rsA = uniform_rsrc((const void*)src, (size_t)P.M * cs * 2);
    const unsigned coff = (unsigned)ch * 128u;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int q = w + 8 * k;
      if (q < nq) {      // wave-uniform (only the first waves have a sixth piece)
        const unsigned vo = pix[k] >= 0 ? (unsigned)((pix[k] & 0x0FFFFFFF) * cs + (pix[k] >> 28) * 8) * 2u : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(patch + q * 1024), 16, vo, coff, 0, 0);
      }
    }
  };

  // ---- weight loader (as gemm_conv_pipe_kernel): instruction i covers rows (i * NW + w) * 8 + lane / 8 of the tile
  const __amdgpu_buffer_rsrc_t rsB = uniform_rsrc((const void*)P.wt, (size_t)P.N * P.K * 2);
  const __amdgpu_buffer_rsrc_t rsZ = uniform_rsrc((const void*)P.wt, 0);
  __amdgpu_buffer_rsrc_t rsW = rsB;
  unsigned wvo[NB_FULL + 1];
#pragma unroll
  for (int i = 0; i < NB_FULL + 1; ++i) {
    const int row = (i * NW + w) * 8 + (lane >> 3);
    const int n = n0 + row;
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    wvo[i] = (row < BN && n < P.N) ? (unsigned)(((size_t)n * (P.wt_pm ? 64 : P.K) + chunk * 8) * 2) : OOB;
  }
  // byte offset of the K-step (chunk ci, tap) inside a weight row
  auto wkoff = [&](const int ci, const int tap) -> unsigned {
    // piece-major weights [K / 64][N][64] (lr_gemm_args.wt_pm): the K-step's piece index is its position in the packed K order
    if (P.wt_pm) return (unsigned)(ci < ncm ? tap * ncm + ci : 9 * ncm + (ci - ncm)) * (unsigned)P.N * 128u;
    return (unsigned)((ci < ncm ? tap * Ctot + ci * 64 : 9 * Ctot + (ci - ncm) * 64) * 2);
  };
  auto issue_weights = [&](const int slot, const unsigned koff) __attribute__((always_inline)) {
    char* Bs = wring + slot * B_BYTES;
    if (B_TAIL && w < TAIL_WAVES)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lptr_t)(Bs + ((NB_FULL * NW + w) * 8) * 128), 16, wvo[NB_FULL], koff, 0, 0);
#pragma unroll
    for (int i = 0; i < NB_FULL; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lptr_t)(Bs + ((i * NW + w) * 8) * 128), 16, wvo[i], koff, 0, 0);
  };
  auto advance = [&](int& ci, int& tap) __attribute__((always_inline)) {
    const int nt = ci < ncm ? 9 : 1;
    if (++tap == nt) { tap = 0; ++ci; }
  };

  // ---- fragments: activation = 16 consecutive patch pixels of line (wave line i + ky) starting at column kx; weights as the GEMM
  const int pb0 = (wm * TM + (seg ? 2 * ((wm * TM) >> 3) : 0)) * HALO_PW + fr;      // (second segment: two halo lines further down)
  auto read_frags = [&](vec8<T> (&xf)[TM], vec8<T> (&wf)[TN], const int slot, const int ks, const int ky, const int kx) {
    const char* Bs = wring + slot * B_BYTES;
    const int kc = ks * 4 + fq;
    const int col = fr + kx;
    const char* xs = patch + (pb0 + kx) * 128 + ((kc ^ (col & 6)) << 4) + ky * (HALO_PW * 128);
#pragma unroll
    for (int i = 0; i < TM; ++i) xf[i] = *reinterpret_cast<const vec8<T>*>(xs + i * (HALO_PW * 128));
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row = weight_tile<TN, WNW, false>(wn, j) * 16 + fr;
      wf[j] = *reinterpret_cast<const vec8<T>*>(Bs + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
    }
  };
  f32x4 acc[TN][TM];
  auto mma = [&](const vec8<T> (&xf)[TM], const vec8<T> (&wf)[TN]) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i)
        acc[j][i] = lr_mfma16(wf[j], xf[i], acc[j][i]);
  };
  // spread NDMA just-issued LDS-DMA instructions between the MFMAs that follow them in program order
  auto spread = [&](auto ndma) __attribute__((always_inline)) {
    constexpr int ND = decltype(ndma)::value;
    constexpr int PER = (TM * TN) / ND > 0 ? (TM * TN) / ND : 1;
#pragma unroll
    for (int g = 0; g < ND; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x8, PER, 0);
      __builtin_amdgcn_sched_group_barrier(0x10, 1, 0);
    }
  };
  using std::integral_constant;

  LR_STAMP(0);
  stage_params<BN, PAR_LD>(P, par, n0, w, lane, 0);
  if (nsteps > 0) issue_patch(c_begin);  // (block-uniform)
  int wci = c_begin, wtap = 0;           // K-step whose weights are issued next
#pragma unroll
  for (int sidx = 0; sidx < NSTAGE - 1; ++sidx) {
    if (sidx < nsteps) { issue_weights(sidx, wkoff(wci, wtap)); advance(wci, wtap); }
  }
  LR_STAMP(1);
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (NSTAGE == 3 && nsteps > 1) {       // patch + step 0 landed; step 1's weights may still be in flight
    if (B_TAIL && w < TAIL_WAVES) wait_vmcnt<NB_FULL + 1>(); else wait_vmcnt<NB_FULL>();
  } else {
    wait_vmcnt<0>();
  }
  __builtin_amdgcn_s_barrier();
  LR_STAMP(2);

  int ci = c_begin, tap = 0, cur = 0;
  vec8<T> xa[TM], wa[TN];
  for (int s = 0; s < nsteps; ++s) {
    const bool main_part = ci < ncm;
    const int ky = main_part ? (tap * 11) >> 5 : 1, kx = main_part ? tap - 3 * ky : 1;      // (tap * 11) >> 5 = tap / 3 for tap < 12
    const bool last_tap = !main_part || tap == 8;
    const bool pf_w = s + NSTAGE - 1 < nsteps;          // weights to prefetch in this step
    const bool pf_p = last_tap && ci + 1 < nch;         // patch to reload in this step
    int wslot = cur + NSTAGE - 1; if (wslot >= NSTAGE) wslot -= NSTAGE;
    read_frags(xa, wa, cur, 0, ky, kx);
    __builtin_amdgcn_sched_barrier(0);
    // (past the end of K the weights go through a zero-length descriptor: same instruction count in every step, constant waits)
    rsW = (pf_w && HALO_EXP != 3) ? rsB : rsZ;
    issue_weights(wslot, wkoff(wci, wtap));
    advance(wci, wtap);
    mma(xa, wa);
    spread(integral_constant<int, NB_FULL>{});
    __builtin_amdgcn_sched_barrier(0);
    read_frags(xa, wa, cur, 1, ky, kx);
    if (pf_p && HALO_EXP != 1) {
      // every wave's last reads of this patch are in registers -> the buffer is free for the next chunk
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      issue_patch(ci + 1);
    }
    mma(xa, wa);
    if (NSTAGE == 3 && !pf_p) {                            // the weights issued in this step stay in flight
      if (B_TAIL && w < TAIL_WAVES) wait_vmcnt<NB_FULL + 1>(); else wait_vmcnt<NB_FULL>();
    } else {
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    cur = cur + 1 == NSTAGE ? 0 : cur + 1;
    advance(ci, tap);
  }
  LR_STAMP(3);
  LR_STAMP(4);

  // ---- epilogue straight from the accumulators (the loop ended with vmcnt(0) + barrier: the patch buffer is free for the sums)
  float* gsl = reinterpret_cast<float*>(smem);
  const bool gp = P.gp_out != nullptr && P.splits == 1;
  epilogue_units<TM, TN, 0, PAR_LD, WNW, T, true>(P, acc, m_org + wm * TM * P.W, n0, wn, lane, par, par, tile_n * WNW + wn,
                                                  gsl + wm * (BN * 2), tile * WMW + wm);
  if (gp) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    gn_group_reduce<BN, WMW, 256>(P, gsl, tile * 256, n0, t);
  }
#ifdef LR_DEV_VARIANTS
  if (P.sk_cnt != nullptr && P.splits > 1)      // in-launch split-K reduce: sub-block j = two 16-pixel line segments of the tile
    sk_fused_tail<T, BN>(P, tile * P.ntiles_n + tile_n, 8,
                         [&](const int j, const int r) { return m_org + (2 * j + (r >> 4)) * P.W + (r & 15); },
                         [&](const int j) { return tile * 8 + j; }, [&](const int j) { return tile * 256 + 32 * j; }, n0,
                         reinterpret_cast<float*>(smem), t);
#endif
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
  // (the grouped tile order of gemm_conv_pipe_kernel was measured here too: no effect -- 4 column tiles at most, one round)
  // developer build (LR_DEV), default off (LR_HALO_MFASTEST=1): split-K launches walk m-fastest, so an XCD's consecutive tiles share ONE column tile's
  // weight slice instead of spanning all of them.  Measured (profiles/r05_conv_halo_proto.txt section 15): the launches' L2-miss bytes fall as
  // that predicts (4096 x 1280 x 23040: 578 -> 287 MB, 1024 x 1280 x 23040: 289 -> 143; family -1.34 GB per step), the UNet step does not move
  // (seven same-box pairs: +0.006 ... +0.031 ms, one -0.10) -- not enabled: it would only improve the traffic figure.
  if (LR_DEV("LR_HALO_MFASTEST", 0) && P.splits > 1) P.m_fastest = 1;
  P.nblocks = P.ntiles_n * P.ntiles_m;
  const size_t smem = (size_t)HALO_NQ * 1024 + NSTAGE * (size_t)BN * 128 + 2 * (((BN + 63) / 64) * 64) * sizeof(float);
  static unsigned long long attr_done = 0;
  if (lr_attr_needed(&attr_done)) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo_kernel<BN, WMW, NSTAGE, T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)smem);
  }
  hipLaunchKernelGGL((conv_halo_kernel<BN, WMW, NSTAGE, T>), dim3(P.nblocks, P.splits), dim3(512), smem, st, P);
  return lr_launch_status();
}

// the LR_PIPE_HALO instances of lr_gemm_conv_f16 (called from gemm_conv.hip after its argument checks)
int lr_launch_conv_halo(const GemmParams& P, int tile_n, hipStream_t st) {
  if (P.taps != 9 || P.stride != 1 || P.up || P.zins || P.pad != 1 || P.c16 || P.splits < 1 || P.geglu || P.gelu || P.ln_part ||
      P.wt_bstride || P.st_out || ((P.H & 15) && !(P.H == 8 && P.M % 256 == 0)) || (P.W & 15) || P.Hs != P.H || P.Ws != P.W)
    return LR_E_UNSUPPORTED;
  if ((long long)P.M >= (1ll << 28)) return LR_E_UNSUPPORTED;      // the patch loader packs (pixel index | chunk << 28) into one register
  // (splits above the number of 64-channel chunks are legal: the surplus slices own no chunk and write zero partials -- made explicit in the
  // kernel, ADVICE r5: c_begin is clamped, an empty slice issues no patch load and runs no K-step)
  if (tile_n == 320) return P.bf16 ? launch_halo_t<320, 2, 2, bf16>(P, st) : launch_halo_t<320, 2, 2, f16>(P, st);
  if (tile_n == 160) return P.bf16 ? launch_halo_t<160, 4, 3, bf16>(P, st) : launch_halo_t<160, 4, 3, f16>(P, st);
  return LR_E_UNSUPPORTED;
}

unsigned lr_halo_sk_timeouts() {
  unsigned v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(lr_sk_error), sizeof(v)) != hipSuccess) return 0xFFFFFFFFu;
  return v;
}
