// GroupNorm(32)[+SiLU] and LayerNorm for NHWC fp16 activations -- HBM-bound streaming kernels.
//
// Layout: x [N][HW][C] fp16, channels contiguous (one pixel = one row).  Every global access is a 16-byte
// (8 x fp16) vector, consecutive lanes read consecutive 16-byte pieces of a row => full-line coalescing.
#include "common.h"

#include <stdlib.h>

// ---------------------------------------------------------------------------------------------------------------
// GroupNorm statistics.  grid = (LR_GN_CHUNKS, N); block = nOct * R threads where nOct = C/8 octets per pixel and
// R = pixel rows per sweep.  Thread (o, r) always owns octet o and keeps per-channel fp32 (sum, sumsq); the block
// combines them per group in a FIXED order through LDS (no atomics => bitwise reproducible reruns) and writes
// partials[n][chunk][32][2].
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void gn_stats_kernel(const T* __restrict__ x1, int C1, const T* __restrict__ x2, int C2, int HW,
                                float* __restrict__ partials, int nOct, int R, int nchunks) {
  extern __shared__ float s_part[];  // [R][C][2]
  const int C = C1 + C2;
  const int Cg = C / 32;
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int t = threadIdx.x;
  const int o = t % nOct, r = t / nOct;
  const int per = (HW + nchunks - 1) / nchunks;
  const int p0 = chunk * per, p1 = min(HW, p0 + per);
  const int c0 = o * 8;
  const T* src;
  int cs, coff;
  if (c0 < C1) { src = x1; cs = C1; coff = c0; } else { src = x2; cs = C2; coff = c0 - C1; }
  float sm[8], sq[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { sm[i] = 0.f; sq[i] = 0.f; }
  const T* base = src + ((size_t)n * HW) * cs + coff;
  int p = p0 + r;
  // four pixels in flight per thread
  for (; p + 3 * R < p1; p += 4 * R) {
    uint4 u[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) u[k] = *reinterpret_cast<const uint4*>(base + (size_t)(p + k * R) * cs);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float f[8];
      lr_unpack8<T>(u[k], f);
#pragma unroll
      for (int i = 0; i < 8; ++i) { sm[i] += f[i]; sq[i] = fmaf(f[i], f[i], sq[i]); }
    }
  }
  for (; p < p1; p += R) {
    const uint4 u0 = *reinterpret_cast<const uint4*>(base + (size_t)p * cs);
    float f[8];
    lr_unpack8<T>(u0, f);
#pragma unroll
    for (int i = 0; i < 8; ++i) { sm[i] += f[i]; sq[i] = fmaf(f[i], f[i], sq[i]); }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    s_part[((size_t)r * C + c0 + i) * 2 + 0] = sm[i];
    s_part[((size_t)r * C + c0 + i) * 2 + 1] = sq[i];
  }
  __syncthreads();
  if (t < 32) {
    float s = 0.f, q = 0.f;
    for (int c = t * Cg; c < (t + 1) * Cg; ++c)
      for (int rr = 0; rr < R; ++rr) { s += s_part[((size_t)rr * C + c) * 2]; q += s_part[((size_t)rr * C + c) * 2 + 1]; }
    float* dst = partials + (((size_t)n * nchunks + chunk) * 32 + t) * 2;
    dst[0] = s;
    dst[1] = q;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// GroupNorm apply (+ optional SiLU).  grid = (pixel blocks, N); block = nOct * R threads like the stats kernel, so
// thread (o, r) owns channel octet o for its whole life: the 8 scale / shift coefficients sit in registers and the
// streaming loop is pure load -> 8 fma (+SiLU) -> store with four 16-byte loads in flight (no LDS table, no
// per-element index division).  Each block first finalises mean/rstd of its sample from the LR_GN_CHUNKS partials
// (fp64, fixed order => deterministic).
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void gn_apply_kernel(const T* __restrict__ x1, int C1, const T* __restrict__ x2, int C2, int HW,
                                const float* __restrict__ partials, const float* __restrict__ gamma,
                                const float* __restrict__ beta, float eps, int silu, T* __restrict__ y,
                                int pix_per_block, int nOct, int R, int nchunks) {
  __shared__ float s_mean[32], s_rstd[32];
  const int C = C1 + C2;
  const int Cg = C / 32;
  const int n = blockIdx.y;
  const int t = threadIdx.x;
  const int o = t % nOct, r = t / nOct;
  const int c0 = o * 8;
  const int p0 = blockIdx.x * pix_per_block;
  const int p1 = min(HW, p0 + pix_per_block);
  const T* src;
  int cs;
  if (c0 < C1) { src = x1 + ((size_t)n * HW) * C1 + c0; cs = C1; }
  else { src = x2 + ((size_t)n * HW) * C2 + (c0 - C1); cs = C2; }
  T* dst = y + ((size_t)n * HW) * C + c0;

  // request the first pixels before the statistics prologue (their latency hides under it)
  int p = p0 + r;
  uint4 u[4];
  const bool first = p + 3 * R < p1;
  if (first) {
#pragma unroll
    for (int k = 0; k < 4; ++k) u[k] = *reinterpret_cast<const uint4*>(src + (size_t)(p + k * R) * cs);
  }
  float ga[8], be[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { ga[i] = gamma[c0 + i]; be[i] = beta[c0 + i]; }
  if (t < 128) {   // 4 lanes per group split the partials (blockDim >= 128 always)
    const int g = t >> 2, sub = t & 3;
    double s = 0.0, q = 0.0;
    const float* ps = partials + ((size_t)n * nchunks * 32 + g) * 2;
#pragma unroll 4
    for (int c = sub; c < nchunks; c += 4) { s += (double)ps[c * 64]; q += (double)ps[c * 64 + 1]; }
#pragma unroll
    for (int sh = 2; sh > 0; sh >>= 1) { s += __shfl_xor(s, sh, 64); q += __shfl_xor(q, sh, 64); }
    if (sub == 0) {
      const double cnt = (double)HW * (double)Cg;
      const double mean = s / cnt;
      double var = q / cnt - mean * mean;
      if (var < 0.0) var = 0.0;
      s_mean[g] = (float)mean;
      s_rstd[g] = (float)(1.0 / sqrt(var + (double)eps));
    }
  }
  __syncthreads();
  float a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int g = (c0 + i) / Cg;
    a[i] = s_rstd[g] * ga[i];
    b[i] = be[i] - s_mean[g] * a[i];
  }
  auto emit = [&](int pp, const uint4& v) {
    float f[8];
    lr_unpack8<T>(v, f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float z = fmaf(f[i], a[i], b[i]);
      f[i] = silu ? z * __builtin_amdgcn_rcpf(1.0f + __expf(-z)) : z;
    }
    *reinterpret_cast<uint4*>(dst + (size_t)pp * C) = lr_pack8<T>(f);
  };
  if (r >= R) return;   // spare threads when blockDim was rounded up (never with nOct * R sizing)
  bool have = first;
  while (have) {
    const int np = p + 4 * R;
    const bool more = np + 3 * R < p1;
    uint4 w[4];
    if (more) {
#pragma unroll
      for (int k = 0; k < 4; ++k) w[k] = *reinterpret_cast<const uint4*>(src + (size_t)(np + k * R) * cs);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) emit(p + k * R, u[k]);
    if (more) {
#pragma unroll
      for (int k = 0; k < 4; ++k) u[k] = w[k];
    }
    p = np;
    have = more;
  }
  for (; p < p1; p += R) emit(p, *reinterpret_cast<const uint4*>(src + (size_t)p * cs));
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row held in registers (C <= 2048), two-pass mean / variance like ATen.
// ---------------------------------------------------------------------------------------------------------------
template <int NV, typename T>  // 16-byte vectors per lane
__global__ void layernorm_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, float eps, T* __restrict__ y, int M, int C) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nOct = C >> 3;
  float v[NV][8];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int o = lane + j * 64;
    if (o < nOct) {
      const uint4 u = *reinterpret_cast<const uint4*>(x + (size_t)row * C + o * 8);
      lr_unpack8<T>(u, v[j]);
#pragma unroll
      for (int i = 0; i < 8; ++i) sum += v[j][i];
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[j][i] = 0.f;
    }
  }
  const float mean = lr_wave_sum(sum) / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int o = lane + j * 64;
    if (o < nOct) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float d = v[j][i] - mean; sq = fmaf(d, d, sq); }
    }
  }
  const float rstd = rsqrtf(lr_wave_sum(sq) / (float)C + eps);
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int o = lane + j * 64;
    if (o < nOct) {
      const float4 g0 = *reinterpret_cast<const float4*>(gamma + o * 8);
      const float4 g1 = *reinterpret_cast<const float4*>(gamma + o * 8 + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(beta + o * 8);
      const float4 b1 = *reinterpret_cast<const float4*>(beta + o * 8 + 4);
      const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const float b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      float f[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] = fmaf((v[j][i] - mean) * rstd, g[i], b[i]);
      *reinterpret_cast<uint4*>(y + (size_t)row * C + o * 8) = lr_pack8<T>(f);
    }
  }
}

// Number of pixel chunks actually used for a tensor of HW pixels per sample (both kernels derive it the same way).
// Measured on MI355X inside a hipGraph (tools/bench_gn.py): ~256 stats blocks / ~512 apply blocks is the sweet spot --
// more blocks lose to the per-block prologue, fewer starve the 256 CUs.  Env overrides are for experiments only.

// ---------------------------------------------------------------------------------------------------------------
// Row softmax of materialised fp16 logits (single-head d = C attention of the VAE AttnBlock, where the logits go
// through the GEMM kernel):  p[m][:] = softmax(scale * s[m][:]) in fp32, one 256-thread block per row, the row stays
// in registers (NV 16-byte vectors per thread).  In place is allowed.
// ---------------------------------------------------------------------------------------------------------------
template <int NV, typename T>
__global__ void softmax_rows_kernel(const T* __restrict__ s, T* __restrict__ p, int N, float scale) {
  __shared__ float red[8];
  const size_t row = blockIdx.x;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const T* src = s + row * (size_t)N;
  T* dst = p + row * (size_t)N;
  float f[NV][8];
  float mx = -INFINITY;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c = (v * 256 + t) * 8;
    if (c < N) {
      const uint4 u = *reinterpret_cast<const uint4*>(src + c);
      lr_unpack8<T>(u, f[v]);
#pragma unroll
      for (int i = 0; i < 8; ++i) { f[v][i] *= scale; mx = fmaxf(mx, f[v][i]); }
    }
  }
  mx = lr_wave_max(mx);
  if (lane == 0) red[w] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c = (v * 256 + t) * 8;
    if (c < N) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { f[v][i] = __expf(f[v][i] - mx); sum += f[v][i]; }
    }
  }
  sum = lr_wave_sum(sum);
  if (lane == 0) red[4 + w] = sum;
  __syncthreads();
  const float inv = 1.f / ((red[4] + red[5]) + (red[6] + red[7]));
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c = (v * 256 + t) * 8;
    if (c < N) {
#pragma unroll
      for (int i = 0; i < 8; ++i) f[v][i] *= inv;
      *reinterpret_cast<uint4*>(dst + c) = lr_pack8<T>(f[v]);
    }
  }
}

static int gn_nchunks(int N, int HW, int C) {
  const int base_blocks = LR_DEV("LR_GN_STAT_BLOCKS", 256);
  // ~1 block per CU for the UNet's tensors (<= 126 MB); tensors of the VAE's size (0.5 GB) get one block per 512 KB so
  // enough loads are in flight to stream at HBM rate
  const long long want = ((long long)N * HW * C * 2) >> 19;
  const int target_blocks = want > base_blocks ? (int)want : base_blocks;
  int c = (target_blocks + N - 1) / N;
  if (c > LR_GN_CHUNKS) c = LR_GN_CHUNKS;
  if (c > HW / 8) c = HW / 8;
  if (c < 1) c = 1;
  return c;
}

template <typename T>
static int lr_groupnorm_stats_t(const lr_half* x1, int C1, const lr_half* x2, int C2, int N, int HW, float* partials,
                                  lr_stream_t s) {
  if (!x1 || !partials || N <= 0 || HW <= 0) return LR_E_ARG;
  if (!x2) C2 = 0;
  const int C = C1 + C2;
  if (C % 32 || C1 % 8 || C2 % 8) return LR_E_ALIGN;
  const int nOct = C / 8;
  int R = 256 / nOct;
  if (R < 1) R = 1;
  const int threads = nOct * R;
  if (threads > 1024) return LR_E_UNSUPPORTED;
  const int nchunks = gn_nchunks(N, HW, C);
  dim3 grid(nchunks, N);
  hipLaunchKernelGGL(gn_stats_kernel<T>, grid, dim3(threads), (size_t)R * C * 2 * sizeof(float), (hipStream_t)s,
                     (const T*)x1, C1, (const T*)x2, C2, HW, partials, nOct, R, nchunks);
  return lr_launch_status();
}

// GroupNorm statistics from the producing GEMMs' epilogues (lr_gemm_args.gn_stats_out): per-channel partial sums over
// row blocks -> per-group (sum, sumsq) of each sample, fixed order.  grid = (32 groups, N), block = 256: the (channel, row
// block) pairs of the group are strided over the threads (independent loads in flight), then a fixed tree combines them.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ p1, int C1, int R1,
                                                          const float* __restrict__ p2, int C2, int R2, int HW,
                                                          float* __restrict__ partials) {
  __shared__ double red[2][4];
  const int g = blockIdx.x, n = blockIdx.y, t = threadIdx.x;
  const int C = C1 + C2, Cg = C / 32;
  double s = 0.0, q = 0.0;
  // the group's channels split into the part that lives in source 1 and the part in source 2 (a group may straddle the
  // concat boundary); each part is a [row blocks][channels] patch flattened over the threads
  const int c_lo = g * Cg, c_hi = c_lo + Cg;
#pragma unroll
  for (int part = 0; part < 2; ++part) {
    const float* src = part ? p2 : p1;
    const int cs = part ? C2 : C1, R = part ? R2 : R1;
    const int lo = part ? max(c_lo - C1, 0) : min(c_lo, C1), hi = part ? max(c_hi - C1, 0) : min(c_hi, C1);
    const int w = hi - lo;
    if (w <= 0 || !src) continue;
    const int nb = HW / R;
    const float* base = src + ((size_t)n * nb * cs + lo) * 2;
    const int total = nb * w;
#pragma unroll 4
    for (int e = t; e < total; e += 256) {
      const int b = e / w, cc = e - b * w;
      const float2 v = *reinterpret_cast<const float2*>(base + ((size_t)b * cs + cc) * 2);
      s += (double)v.x; q += (double)v.y;
    }
  }
#pragma unroll
  for (int sh = 32; sh > 0; sh >>= 1) { s += __shfl_xor(s, sh, 64); q += __shfl_xor(q, sh, 64); }
  if ((t & 63) == 0) { red[0][t >> 6] = s; red[1][t >> 6] = q; }
  __syncthreads();
  if (t == 0) {
    partials[((size_t)n * 32 + g) * 2] = (float)((red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));   // fp32 like the statistics kernel
    partials[((size_t)n * 32 + g) * 2 + 1] = (float)((red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
  }
}

extern "C" int lr_groupnorm_finalize(const float* p1, int C1, int R1, const float* p2, int C2, int R2, int N, int HW,
                                     float* partials, lr_stream_t s) {
  if (!p1 || !partials || N <= 0 || HW <= 0 || R1 <= 0 || HW % R1) return LR_E_ARG;
  if (!p2) C2 = 0;
  if (p2 && (R2 <= 0 || HW % R2)) return LR_E_ARG;
  if ((C1 + C2) % 32) return LR_E_ALIGN;
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(32, N), dim3(256), 0, (hipStream_t)s, p1, C1, R1, p2, C2, p2 ? R2 : 1, HW, partials);
  return lr_launch_status();
}

#ifdef LR_DEV_VARIANTS      // (measured: the fold loses to gn_apply + proj_in in the step -- developer builds only, like the engine switch)
// GroupNorm folded into the pointwise GEMM that consumes it (lr_gn_fold_weights_f16): per sample the normalisation is a per-channel
// scale / shift, which goes into a per-sample copy of the weights.  grid = (N / FOLD_ROWS, B), block = 256 = 4 waves; prologue as in
// gn_apply_kernel (mean / rstd of the sample's 32 groups from the chunk partials, fp64, fixed order), then one wave per weight row.
#define FOLD_ROWS 4      // one weight row per wave: the row work is a dependent load -> scale -> store -> wave-reduce chain
template <typename T>
__global__ __launch_bounds__(256) void gn_fold_weights_kernel(const float* __restrict__ gpart, int nchunks, int HW, int C,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                              const T* __restrict__ w, const float* __restrict__ bias, int N,
                                                              T* __restrict__ w_out, float* __restrict__ bias_out) {
  __shared__ float s_mean[32], s_rstd[32];
  __shared__ float s_a[2048], s_m[2048], s_b[2048];      // per channel: scale a = gamma rstd, the group mean, beta
  const int b = blockIdx.y, t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int Cg = C / 32;
  if (t < 128) {
    const int g = t >> 2, sub = t & 3;
    double s = 0.0, q = 0.0;
    const float* ps = gpart + ((size_t)b * nchunks * 32 + g) * 2;
#pragma unroll 8
    for (int c = sub; c < nchunks; c += 4) { const float2 v = *reinterpret_cast<const float2*>(ps + c * 64); s += (double)v.x; q += (double)v.y; }
#pragma unroll
    for (int sh = 2; sh > 0; sh >>= 1) { s += __shfl_xor(s, sh, 64); q += __shfl_xor(q, sh, 64); }
    if (sub == 0) {
      const double cnt = (double)HW * (double)Cg;
      const double mean = s / cnt;
      double var = q / cnt - mean * mean;
      if (var < 0.0) var = 0.0;
      s_mean[g] = (float)mean;
      s_rstd[g] = (float)(1.0 / sqrt(var + (double)eps));
    }
  }
  // this wave's weight row is requested before the statistics prologue (its latency hides under it)
  const int n = blockIdx.x * FOLD_ROWS + wv;
  const bool have = n < N && lane * 8 < C;
  uint4 w0 = make_uint4(0, 0, 0, 0);
  if (have) w0 = *reinterpret_cast<const uint4*>(w + (size_t)n * C + lane * 8);
  __syncthreads();
  for (int c = t; c < C; c += 256) { const int g = c / Cg; s_a[c] = gamma[c] * s_rstd[g]; s_m[c] = s_mean[g]; s_b[c] = beta[c]; }
  __syncthreads();
  for (int r = wv; r < FOLD_ROWS; r += 4) {
    if (n >= N) break;
    const T* src = w + (size_t)n * C;
    T* dst = w_out + ((size_t)b * N + n) * C;
    float acc_b = 0.f, acc_m = 0.f;      // sum_c W beta ; sum_c rounded(W a) mean
    for (int c0 = lane * 8; c0 < C; c0 += 512) {
      float f[8], o[8];
      lr_unpack8<T>(c0 == lane * 8 ? w0 : *reinterpret_cast<const uint4*>(src + c0), f);
#pragma unroll
      for (int i = 0; i < 8; ++i) { o[i] = f[i] * s_a[c0 + i]; acc_b = fmaf(f[i], s_b[c0 + i], acc_b); }
      const uint4 pk = lr_pack8<T>(o);
      *reinterpret_cast<uint4*>(dst + c0) = pk;
      lr_unpack8<T>(pk, o);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc_m = fmaf(o[i], s_m[c0 + i], acc_m);
    }
    acc_b = lr_wave_sum(acc_b);
    acc_m = lr_wave_sum(acc_m);
    if (lane == 0) bias_out[(size_t)b * N + n] = (bias ? bias[n] : 0.f) + acc_b - acc_m;
  }
}

template <typename T>
static int lr_gn_fold_weights_t(const float* gpart, int chunks, int B, int HW, int C, const float* gamma, const float* beta, float eps,
                                const lr_half* w, const float* bias, int N, lr_half* w_out, float* bias_out, lr_stream_t s) {
  if (!gpart || !gamma || !beta || !w || !w_out || !bias_out || chunks <= 0 || B <= 0 || HW <= 0 || N <= 0) return LR_E_ARG;
  if (C % 32 || C % 8 || C > 2048) return LR_E_ALIGN;
  if (((uintptr_t)w | (uintptr_t)w_out) & 15) return LR_E_ALIGN;
  hipLaunchKernelGGL(gn_fold_weights_kernel<T>, dim3((N + FOLD_ROWS - 1) / FOLD_ROWS, B), dim3(256), 0, (hipStream_t)s, gpart, chunks, HW, C,
                     gamma, beta, eps, (const T*)w, bias, N, (T*)w_out, bias_out);
  return lr_launch_status();
}
extern "C" int lr_gn_fold_weights_f16(const float* gpart, int chunks, int B, int HW, int C, const float* gamma, const float* beta, float eps, const lr_half* w, const float* bias, int N, lr_half* w_out, float* bias_out, lr_stream_t s) { return lr_gn_fold_weights_t<f16>(gpart, chunks, B, HW, C, gamma, beta, eps, w, bias, N, w_out, bias_out, s); }
extern "C" int lr_gn_fold_weights_bf16(const float* gpart, int chunks, int B, int HW, int C, const float* gamma, const float* beta, float eps, const lr_half* w, const float* bias, int N, lr_half* w_out, float* bias_out, lr_stream_t s) { return lr_gn_fold_weights_t<bf16>(gpart, chunks, B, HW, C, gamma, beta, eps, w, bias, N, w_out, bias_out, s); }
#endif  // LR_DEV_VARIANTS

template <typename T>
static int groupnorm_apply_impl(const lr_half* x1, int C1, const lr_half* x2, int C2, int N, int HW, const float* partials,
                                int nchunks_in, const float* gamma, const float* beta, float eps, int silu, lr_half* y,
                                lr_stream_t s);

template <typename T>
static int lr_groupnorm_apply_t(const lr_half* x1, int C1, const lr_half* x2, int C2, int N, int HW,
                                  const float* partials, const float* gamma, const float* beta, float eps, int silu,
                                  lr_half* y, lr_stream_t s) {
  return groupnorm_apply_impl<T>(x1, C1, x2, C2, N, HW, partials, 0, gamma, beta, eps, silu, y, s);
}

template <typename T>
static int lr_groupnorm_apply_n_t(const lr_half* x1, int C1, const lr_half* x2, int C2, int N, int HW,
                                    const float* partials, int nchunks, const float* gamma, const float* beta, float eps,
                                    int silu, lr_half* y, lr_stream_t s) {
  if (nchunks <= 0) return LR_E_ARG;
  return groupnorm_apply_impl<T>(x1, C1, x2, C2, N, HW, partials, nchunks, gamma, beta, eps, silu, y, s);
}

template <typename T>
static int groupnorm_apply_impl(const lr_half* x1, int C1, const lr_half* x2, int C2, int N, int HW, const float* partials,
                                int nchunks_in, const float* gamma, const float* beta, float eps, int silu, lr_half* y,
                                lr_stream_t s) {
  if (!x1 || !partials || !gamma || !beta || !y || N <= 0 || HW <= 0) return LR_E_ARG;
  if (!x2) C2 = 0;
  const int C = C1 + C2;
  if (C % 32 || C1 % 8 || C2 % 8) return LR_E_ALIGN;
  // ~1024 blocks over the whole tensor (4 per CU), at least 16 pixels per block so the per-block finalisation of
  // mean/rstd and the scale/shift table amortise
  int ppb = (int)(((long long)N * HW + 1023) / 1024);
  if (ppb < 16) ppb = 16;
  if (ppb > HW) ppb = HW;
  const int nOct = C / 8;
  int R = 256 / nOct;
  if (R < 1) R = 1;
  const int threads = nOct * R;
  const int base_apply_blocks = LR_DEV("LR_GN_APPLY_BLOCKS", 2048);      // round 5: 512 -> 2048 (tools/bench_gn_apply.py: 654 -> 626 us per step, same bits)
  const long long want_blocks = ((long long)N * HW * C * 2) >> 18;     // one block per 256 KB for very large tensors
  const long long apply_blocks = want_blocks > base_apply_blocks ? want_blocks : base_apply_blocks;
  ppb = (int)(((long long)N * HW + apply_blocks - 1) / apply_blocks);
  if (ppb < 16) ppb = 16;
  if (ppb > HW) ppb = HW;
  ppb = ((ppb + 4 * R - 1) / (4 * R)) * (4 * R);   // whole 4-deep load batches per thread (no serial tail)
  dim3 grid((HW + ppb - 1) / ppb, N);
  if (threads > 1024 || threads < 128) return LR_E_UNSUPPORTED;
  hipLaunchKernelGGL(gn_apply_kernel<T>, grid, dim3(threads), 0, (hipStream_t)s, (const T*)x1, C1, (const T*)x2, C2,
                     HW, partials, gamma, beta, eps, silu, (T*)y, ppb, nOct, R, nchunks_in > 0 ? nchunks_in : gn_nchunks(N, HW, C));
  return lr_launch_status();
}

template <typename T>
static int lr_layernorm_t(const lr_half* x, const float* gamma, const float* beta, float eps, lr_half* y, int M, int C,
                            lr_stream_t s) {
  if (!x || !gamma || !beta || !y || M <= 0) return LR_E_ARG;
  if (C % 8 || C > 2048) return LR_E_ALIGN;
  const int nv = (C / 8 + 63) / 64;
  dim3 grid((M + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)s;
  switch (nv) {
    case 1: hipLaunchKernelGGL((layernorm_kernel<1, T>), grid, block, 0, st, (const T*)x, gamma, beta, eps, (T*)y, M, C); break;
    case 2: hipLaunchKernelGGL((layernorm_kernel<2, T>), grid, block, 0, st, (const T*)x, gamma, beta, eps, (T*)y, M, C); break;
    case 3: hipLaunchKernelGGL((layernorm_kernel<3, T>), grid, block, 0, st, (const T*)x, gamma, beta, eps, (T*)y, M, C); break;
    default: hipLaunchKernelGGL((layernorm_kernel<4, T>), grid, block, 0, st, (const T*)x, gamma, beta, eps, (T*)y, M, C); break;
  }
  return lr_launch_status();
}

extern "C" int lr_softmax_rows_f16(const lr_half* s, lr_half* p, int M, int N, float scale, lr_stream_t st) {
  if (!s || !p || M <= 0 || N <= 0) return LR_E_ARG;
  if (N % 8) return LR_E_ALIGN;
  if (N > 8 * 2048) return LR_E_UNSUPPORTED;
  const int nv = (N + 2047) / 2048;
  dim3 grid(M), block(256);
  hipStream_t hs = (hipStream_t)st;
  if (nv <= 1) hipLaunchKernelGGL((softmax_rows_kernel<1, f16>), grid, block, 0, hs, (const f16*)s, (f16*)p, N, scale);
  else if (nv <= 2) hipLaunchKernelGGL((softmax_rows_kernel<2, f16>), grid, block, 0, hs, (const f16*)s, (f16*)p, N, scale);
  else if (nv <= 4) hipLaunchKernelGGL((softmax_rows_kernel<4, f16>), grid, block, 0, hs, (const f16*)s, (f16*)p, N, scale);
  else hipLaunchKernelGGL((softmax_rows_kernel<8, f16>), grid, block, 0, hs, (const f16*)s, (f16*)p, N, scale);
  return lr_launch_status();
}

// =====================================================================================================================
// Backward (training, frozen weights: only the input gradient is produced).  fp16 activations / gradients, fp32 math.
// =====================================================================================================================

// ---- LayerNorm backward: dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma.  One wave per row.
//      dres (or NULL): the gradient that reaches x along the residual branch around the LayerNorm (x + f(LayerNorm(x))); it is added
//      in fp32 before the one rounding, so the fan-in sum of the two branches is not a separate pass (and not a second rounding).
template <int NV, typename T>
__global__ void layernorm_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ gamma,
                                     float eps, T* __restrict__ dx, int M, int C, const T* __restrict__ dres) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nOct = C >> 3;
  float v[NV][8], g[NV][8];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int o = lane + j * 64;
    if (o < nOct) {
      lr_unpack8<T>(*reinterpret_cast<const uint4*>(x + (size_t)row * C + o * 8), v[j]);
      lr_unpack8<T>(*reinterpret_cast<const uint4*>(dy + (size_t)row * C + o * 8), g[j]);
      const float4 g0 = *reinterpret_cast<const float4*>(gamma + o * 8);
      const float4 g1 = *reinterpret_cast<const float4*>(gamma + o * 8 + 4);
      const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
      for (int i = 0; i < 8; ++i) { sum += v[j][i]; g[j][i] *= gm[i]; }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) { v[j][i] = 0.f; g[j][i] = 0.f; }
    }
  }
  const float mean = lr_wave_sum(sum) / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int o = lane + j * 64;
    if (o < nOct) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float d = v[j][i] - mean; sq = fmaf(d, d, sq); }
    }
  }
  const float rstd = rsqrtf(lr_wave_sum(sq) / (float)C + eps);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int o = lane + j * 64;
    if (o < nOct) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        v[j][i] = (v[j][i] - mean) * rstd;     // xhat
        s1 += g[j][i];
        s2 = fmaf(g[j][i], v[j][i], s2);
      }
    }
  }
  s1 = lr_wave_sum(s1) / (float)C;
  s2 = lr_wave_sum(s2) / (float)C;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int o = lane + j * 64;
    if (o < nOct) {
      float f[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] = rstd * (g[j][i] - s1 - v[j][i] * s2);
      if (dres) {
        float e[8];
        lr_unpack8<T>(*reinterpret_cast<const uint4*>(dres + (size_t)row * C + o * 8), e);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] += e[i];
      }
      *reinterpret_cast<uint4*>(dx + (size_t)row * C + o * 8) = lr_pack8<T>(f);
    }
  }
}

template <typename T>
static int lr_layernorm_bwd_t(const lr_half* x, const lr_half* dy, const float* gamma, float eps, lr_half* dx, int M,
                                int C, lr_stream_t s, const lr_half* dres = nullptr) {
  if (!x || !dy || !gamma || !dx || M <= 0) return LR_E_ARG;
  if (C % 8 || C > 2048) return LR_E_ALIGN;
  const int nv = (C / 8 + 63) / 64;
  dim3 grid((M + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)s;
  switch (nv) {
    case 1: hipLaunchKernelGGL((layernorm_bwd_kernel<1, T>), grid, block, 0, st, (const T*)x, (const T*)dy, gamma, eps, (T*)dx, M, C, (const T*)dres); break;
    case 2: hipLaunchKernelGGL((layernorm_bwd_kernel<2, T>), grid, block, 0, st, (const T*)x, (const T*)dy, gamma, eps, (T*)dx, M, C, (const T*)dres); break;
    case 3: hipLaunchKernelGGL((layernorm_bwd_kernel<3, T>), grid, block, 0, st, (const T*)x, (const T*)dy, gamma, eps, (T*)dx, M, C, (const T*)dres); break;
    default: hipLaunchKernelGGL((layernorm_bwd_kernel<4, T>), grid, block, 0, st, (const T*)x, (const T*)dy, gamma, eps, (T*)dx, M, C, (const T*)dres); break;
  }
  return lr_launch_status();
}

// ---- GroupNorm(32)[+SiLU] backward.  z = xhat * gamma + beta, y = act(z); dz = dy * act'(z), g = dz * gamma;
//      per (sample, group): S1 = sum g, S2 = sum g * xhat;  dx = rstd * (g - (S1 + xhat * S2) / m),  m = HW * C/32.
//      Two launches like the forward: bwd_stats -> partials [N][chunks][32][2] (S1, S2), bwd_apply -> dx1 | dx2.
//      Both re-derive mean / rstd from the FORWARD partials (lr_groupnorm_stats of the same input).
__device__ __forceinline__ void gn_finalize_stats(const float* __restrict__ partials, int n, int nchunks, int HW, int Cg,
                                                  float eps, float* s_mean, float* s_rstd) {
  const int t = threadIdx.x;
  if (t < 128) {
    const int g = t >> 2, sub = t & 3;
    double s = 0.0, q = 0.0;
    const float* ps = partials + ((size_t)n * nchunks * 32 + g) * 2;
    for (int c = sub; c < nchunks; c += 4) { s += (double)ps[c * 64]; q += (double)ps[c * 64 + 1]; }
#pragma unroll
    for (int sh = 2; sh > 0; sh >>= 1) { s += __shfl_xor(s, sh, 64); q += __shfl_xor(q, sh, 64); }
    if (sub == 0) {
      const double cnt = (double)HW * (double)Cg;
      const double mean = s / cnt;
      double var = q / cnt - mean * mean;
      if (var < 0.0) var = 0.0;
      s_mean[g] = (float)mean;
      s_rstd[g] = (float)(1.0 / sqrt(var + (double)eps));
    }
  }
}

__device__ __forceinline__ float gn_act_grad(float z, int silu) {
  if (!silu) return 1.0f;
  const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
  return sg * fmaf(z, 1.0f - sg, 1.0f);
}

template <typename T>
__global__ void gn_bwd_stats_kernel(const T* __restrict__ x1, int C1, const T* __restrict__ x2, int C2, int HW,
                                    const T* __restrict__ dy, const float* __restrict__ fwd, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, float eps, int silu, float* __restrict__ out, int nOct,
                                    int R, int nchunks, int fwd_chunks) {
  extern __shared__ float s_part[];  // [R][C][2]
  __shared__ float s_mean[32], s_rstd[32];
  const int C = C1 + C2, Cg = C / 32;
  const int n = blockIdx.y, chunk = blockIdx.x, t = threadIdx.x;
  const int o = t % nOct, r = t / nOct;
  const int per = (HW + nchunks - 1) / nchunks;
  const int p0 = chunk * per, p1 = min(HW, p0 + per);
  const int c0 = o * 8;
  gn_finalize_stats(fwd, n, fwd_chunks, HW, Cg, eps, s_mean, s_rstd);
  __syncthreads();
  const T* src;
  int cs;
  if (c0 < C1) { src = x1 + ((size_t)n * HW) * C1 + c0; cs = C1; } else { src = x2 + ((size_t)n * HW) * C2 + (c0 - C1); cs = C2; }
  const T* gsrc = dy + ((size_t)n * HW) * C + c0;
  float mu[8], rs[8], ga[8], be[8], s1[8], s2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int g = (c0 + i) / Cg;
    mu[i] = s_mean[g]; rs[i] = s_rstd[g]; ga[i] = gamma[c0 + i]; be[i] = beta[c0 + i];
    s1[i] = 0.f; s2[i] = 0.f;
  }
  auto accum = [&](const uint4& ux, const uint4& ug) {
    float xv[8], gv[8];
    lr_unpack8<T>(ux, xv);
    lr_unpack8<T>(ug, gv);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float xh = (xv[i] - mu[i]) * rs[i];
      const float z = fmaf(xh, ga[i], be[i]);
      const float g = gv[i] * gn_act_grad(z, silu) * ga[i];
      s1[i] += g;
      s2[i] = fmaf(g, xh, s2[i]);
    }
  };
  int p = p0 + r;
  for (; p + 3 * R < p1; p += 4 * R) {      // eight 16-byte loads in flight per thread
    uint4 ux[4], ug[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      ux[k] = *reinterpret_cast<const uint4*>(src + (size_t)(p + k * R) * cs);
      ug[k] = *reinterpret_cast<const uint4*>(gsrc + (size_t)(p + k * R) * C);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) accum(ux[k], ug[k]);
  }
  for (; p < p1; p += R)
    accum(*reinterpret_cast<const uint4*>(src + (size_t)p * cs), *reinterpret_cast<const uint4*>(gsrc + (size_t)p * C));
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    s_part[((size_t)r * C + c0 + i) * 2 + 0] = s1[i];
    s_part[((size_t)r * C + c0 + i) * 2 + 1] = s2[i];
  }
  __syncthreads();
  if (t < 32) {
    float a = 0.f, b = 0.f;
    for (int c = t * Cg; c < (t + 1) * Cg; ++c)
      for (int rr = 0; rr < R; ++rr) { a += s_part[((size_t)rr * C + c) * 2]; b += s_part[((size_t)rr * C + c) * 2 + 1]; }
    float* dst = out + (((size_t)n * nchunks + chunk) * 32 + t) * 2;
    dst[0] = a;
    dst[1] = b;
  }
}

template <typename T>
__global__ void gn_bwd_apply_kernel(const T* __restrict__ x1, int C1, const T* __restrict__ x2, int C2, int HW,
                                    const T* __restrict__ dy, const float* __restrict__ fwd, const float* __restrict__ bwd,
                                    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int silu,
                                    T* __restrict__ dx1, T* __restrict__ dx2, int pix_per_block, int nOct, int R,
                                    int nchunks, int fwd_chunks, const T* __restrict__ dres1, const T* __restrict__ dres2) {
  __shared__ float s_mean[32], s_rstd[32], s_c1[32], s_c2[32];
  const int C = C1 + C2, Cg = C / 32;
  const int n = blockIdx.y, t = threadIdx.x;
  const int o = t % nOct, r = t / nOct;
  const int c0 = o * 8;
  const int p0 = blockIdx.x * pix_per_block, p1 = min(HW, p0 + pix_per_block);
  gn_finalize_stats(fwd, n, fwd_chunks, HW, Cg, eps, s_mean, s_rstd);
  __syncthreads();
  if (t < 128) {
    const int g = t >> 2, sub = t & 3;
    double a = 0.0, b = 0.0;
    const float* ps = bwd + ((size_t)n * nchunks * 32 + g) * 2;
    for (int c = sub; c < nchunks; c += 4) { a += (double)ps[c * 64]; b += (double)ps[c * 64 + 1]; }
#pragma unroll
    for (int sh = 2; sh > 0; sh >>= 1) { a += __shfl_xor(a, sh, 64); b += __shfl_xor(b, sh, 64); }
    if (sub == 0) {
      const double m = (double)HW * (double)Cg;
      s_c1[g] = (float)(a / m);
      s_c2[g] = (float)(b / m);
    }
  }
  __syncthreads();
  if (r >= R) return;
  const T* src;
  const T* rsrc;      // residual-branch gradient of this source (or NULL), same layout as the source
  T* dst;
  int cs;
  if (c0 < C1) { src = x1 + ((size_t)n * HW) * C1 + c0; dst = dx1 + ((size_t)n * HW) * C1 + c0; cs = C1; rsrc = dres1 ? dres1 + ((size_t)n * HW) * C1 + c0 : nullptr; }
  else { src = x2 + ((size_t)n * HW) * C2 + (c0 - C1); dst = dx2 + ((size_t)n * HW) * C2 + (c0 - C1); cs = C2; rsrc = dres2 ? dres2 + ((size_t)n * HW) * C2 + (c0 - C1) : nullptr; }
  const T* gsrc = dy + ((size_t)n * HW) * C + c0;
  float mu[8], rs[8], ga[8], be[8], k1[8], k2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int g = (c0 + i) / Cg;
    mu[i] = s_mean[g]; rs[i] = s_rstd[g]; ga[i] = gamma[c0 + i]; be[i] = beta[c0 + i];
    k1[i] = s_c1[g]; k2[i] = s_c2[g];
  }
  auto emit = [&](int pp, const uint4& ux, const uint4& ug) {
    float xv[8], gv[8], f[8];
    lr_unpack8<T>(ux, xv);
    lr_unpack8<T>(ug, gv);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float xh = (xv[i] - mu[i]) * rs[i];
      const float z = fmaf(xh, ga[i], be[i]);
      const float g = gv[i] * gn_act_grad(z, silu) * ga[i];
      f[i] = rs[i] * (g - k1[i] - xh * k2[i]);
    }
    if (rsrc) {      // fan-in of the residual branch (x + f(GroupNorm(x))): fp32 add before the one rounding
      float e[8];
      lr_unpack8<T>(*reinterpret_cast<const uint4*>(rsrc + (size_t)pp * cs), e);
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] += e[i];
    }
    *reinterpret_cast<uint4*>(dst + (size_t)pp * cs) = lr_pack8<T>(f);
  };
  int p = p0 + r;
  for (; p + 3 * R < p1; p += 4 * R) {
    uint4 ux[4], ug[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      ux[k] = *reinterpret_cast<const uint4*>(src + (size_t)(p + k * R) * cs);
      ug[k] = *reinterpret_cast<const uint4*>(gsrc + (size_t)(p + k * R) * C);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) emit(p + k * R, ux[k], ug[k]);
  }
  for (; p < p1; p += R)
    emit(p, *reinterpret_cast<const uint4*>(src + (size_t)p * cs), *reinterpret_cast<const uint4*>(gsrc + (size_t)p * C));
}

template <typename T>
static int lr_groupnorm_bwd_t(const lr_half* x1, int C1, const lr_half* x2, int C2, const lr_half* dy, int N, int HW,
                                const float* fwd_partials, const float* gamma, const float* beta, float eps, int silu,
                                float* bwd_partials, lr_half* dx1, lr_half* dx2, lr_stream_t s, int fwd_chunks = 0,
                                const lr_half* dres1 = nullptr, const lr_half* dres2 = nullptr) {
  if (!x1 || !dy || !fwd_partials || !bwd_partials || !gamma || !beta || !dx1 || N <= 0 || HW <= 0 || fwd_chunks < 0) return LR_E_ARG;
  if (!x2) C2 = 0;
  if (C2 && !dx2) return LR_E_ARG;
  const int C = C1 + C2;
  if (C % 32 || C1 % 8 || C2 % 8) return LR_E_ALIGN;
  const int nOct = C / 8;
  int R = 256 / nOct;
  if (R < 1) R = 1;
  const int threads = nOct * R;
  if (threads > 1024 || threads < 128) return LR_E_UNSUPPORTED;
  if (fwd_chunks == 0) fwd_chunks = gn_nchunks(N, HW, C);      // fwd_partials from lr_groupnorm_stats of the same input
  // chunks of the backward's own statistics pass: ~1024 blocks (round 6; the forward pass's ~256 left one 4-wave block per CU, 32 KB of
  // loads in flight per CU: 2.6 TB/s on x + dy) -- a pure function of the shape like the forward's count
  int nchunks = (LR_DEV("LR_GN_BWD_BLOCKS", 1024) + N - 1) / N;
  {
    const int lo = gn_nchunks(N, HW, C);
    if (nchunks < lo) nchunks = lo;
    if (nchunks > LR_GN_CHUNKS) nchunks = LR_GN_CHUNKS;
    if (nchunks > HW / 8) nchunks = HW / 8;
    if (nchunks < 1) nchunks = 1;
  }
  hipStream_t st = (hipStream_t)s;
  hipLaunchKernelGGL(gn_bwd_stats_kernel<T>, dim3(nchunks, N), dim3(threads), (size_t)R * C * 2 * sizeof(float), st,
                     (const T*)x1, C1, (const T*)x2, C2, HW, (const T*)dy, fwd_partials, gamma, beta, eps, silu,
                     bwd_partials, nOct, R, nchunks, fwd_chunks);
  int rc = lr_launch_status();
  if (rc) return rc;
  long long blocks = ((long long)N * HW * C * 2) >> 18;
  if (blocks < 512) blocks = 512;
  int ppb = (int)(((long long)N * HW + blocks - 1) / blocks);
  if (ppb < 16) ppb = 16;
  if (ppb > HW) ppb = HW;
  hipLaunchKernelGGL(gn_bwd_apply_kernel<T>, dim3((HW + ppb - 1) / ppb, N), dim3(threads), 0, st, (const T*)x1, C1,
                     (const T*)x2, C2, HW, (const T*)dy, fwd_partials, bwd_partials, gamma, beta, eps, silu, (T*)dx1,
                     (T*)dx2, ppb, nOct, R, nchunks, fwd_chunks, (const T*)dres1, (const T*)dres2);
  return lr_launch_status();
}

// ---- C ABI: every entry point in its fp16 and bf16 form -------------------------------------------------------------
extern "C" int lr_groupnorm_stats(const lr_half* x1, int C1, const lr_half* x2, int C2, int N, int HW, float* partials, lr_stream_t s) { return lr_groupnorm_stats_t<f16>(x1, C1, x2, C2, N, HW, partials, s); }
extern "C" int lr_groupnorm_stats_bf16(const lr_half* x1, int C1, const lr_half* x2, int C2, int N, int HW, float* partials, lr_stream_t s) { return lr_groupnorm_stats_t<bf16>(x1, C1, x2, C2, N, HW, partials, s); }
extern "C" int lr_groupnorm_apply(const lr_half* x1, int C1, const lr_half* x2, int C2, int N, int HW, const float* partials, const float* gamma, const float* beta, float eps, int silu, lr_half* y, lr_stream_t s) { return lr_groupnorm_apply_t<f16>(x1, C1, x2, C2, N, HW, partials, gamma, beta, eps, silu, y, s); }
extern "C" int lr_groupnorm_apply_bf16(const lr_half* x1, int C1, const lr_half* x2, int C2, int N, int HW, const float* partials, const float* gamma, const float* beta, float eps, int silu, lr_half* y, lr_stream_t s) { return lr_groupnorm_apply_t<bf16>(x1, C1, x2, C2, N, HW, partials, gamma, beta, eps, silu, y, s); }
extern "C" int lr_groupnorm_apply_n(const lr_half* x1, int C1, const lr_half* x2, int C2, int N, int HW, const float* partials, int nchunks, const float* gamma, const float* beta, float eps, int silu, lr_half* y, lr_stream_t s) { return lr_groupnorm_apply_n_t<f16>(x1, C1, x2, C2, N, HW, partials, nchunks, gamma, beta, eps, silu, y, s); }
extern "C" int lr_groupnorm_apply_n_bf16(const lr_half* x1, int C1, const lr_half* x2, int C2, int N, int HW, const float* partials, int nchunks, const float* gamma, const float* beta, float eps, int silu, lr_half* y, lr_stream_t s) { return lr_groupnorm_apply_n_t<bf16>(x1, C1, x2, C2, N, HW, partials, nchunks, gamma, beta, eps, silu, y, s); }
extern "C" int lr_layernorm(const lr_half* x, const float* gamma, const float* beta, float eps, lr_half* y, int M, int C, lr_stream_t s) { return lr_layernorm_t<f16>(x, gamma, beta, eps, y, M, C, s); }
extern "C" int lr_layernorm_bf16(const lr_half* x, const float* gamma, const float* beta, float eps, lr_half* y, int M, int C, lr_stream_t s) { return lr_layernorm_t<bf16>(x, gamma, beta, eps, y, M, C, s); }
extern "C" int lr_layernorm_bwd(const lr_half* x, const lr_half* dy, const float* gamma, float eps, lr_half* dx, int M, int C, lr_stream_t s) { return lr_layernorm_bwd_t<f16>(x, dy, gamma, eps, dx, M, C, s); }
extern "C" int lr_layernorm_bwd_bf16(const lr_half* x, const lr_half* dy, const float* gamma, float eps, lr_half* dx, int M, int C, lr_stream_t s) { return lr_layernorm_bwd_t<bf16>(x, dy, gamma, eps, dx, M, C, s); }
extern "C" int lr_groupnorm_bwd(const lr_half* x1, int C1, const lr_half* x2, int C2, const lr_half* dy, int N, int HW, const float* fwd_partials, const float* gamma, const float* beta, float eps, int silu, float* bwd_partials, lr_half* dx1, lr_half* dx2, lr_stream_t s) { return lr_groupnorm_bwd_t<f16>(x1, C1, x2, C2, dy, N, HW, fwd_partials, gamma, beta, eps, silu, bwd_partials, dx1, dx2, s); }
extern "C" int lr_groupnorm_bwd_bf16(const lr_half* x1, int C1, const lr_half* x2, int C2, const lr_half* dy, int N, int HW, const float* fwd_partials, const float* gamma, const float* beta, float eps, int silu, float* bwd_partials, lr_half* dx1, lr_half* dx2, lr_stream_t s) { return lr_groupnorm_bwd_t<bf16>(x1, C1, x2, C2, dy, N, HW, fwd_partials, gamma, beta, eps, silu, bwd_partials, dx1, dx2, s); }
// ABI 25: the same backward passes with the residual-branch gradient(s) of the input added before the rounding (dres* may be NULL), and
// GroupNorm forward partials with their own chunk count (the producer-epilogue [N][fwd_chunks][32][2] sums of lr_gemm_args.gn_group_out)
extern "C" int lr_layernorm_bwd_res(const lr_half* x, const lr_half* dy, const lr_half* dres, const float* gamma, float eps, lr_half* dx, int M, int C, lr_stream_t s) { return lr_layernorm_bwd_t<f16>(x, dy, gamma, eps, dx, M, C, s, dres); }
extern "C" int lr_layernorm_bwd_res_bf16(const lr_half* x, const lr_half* dy, const lr_half* dres, const float* gamma, float eps, lr_half* dx, int M, int C, lr_stream_t s) { return lr_layernorm_bwd_t<bf16>(x, dy, gamma, eps, dx, M, C, s, dres); }
extern "C" int lr_groupnorm_bwd_res(const lr_half* x1, int C1, const lr_half* x2, int C2, const lr_half* dy, const lr_half* dres1, const lr_half* dres2, int N, int HW, const float* fwd_partials, int fwd_chunks, const float* gamma, const float* beta, float eps, int silu, float* bwd_partials, lr_half* dx1, lr_half* dx2, lr_stream_t s) { return lr_groupnorm_bwd_t<f16>(x1, C1, x2, C2, dy, N, HW, fwd_partials, gamma, beta, eps, silu, bwd_partials, dx1, dx2, s, fwd_chunks, dres1, dres2); }
extern "C" int lr_groupnorm_bwd_res_bf16(const lr_half* x1, int C1, const lr_half* x2, int C2, const lr_half* dy, const lr_half* dres1, const lr_half* dres2, int N, int HW, const float* fwd_partials, int fwd_chunks, const float* gamma, const float* beta, float eps, int silu, float* bwd_partials, lr_half* dx1, lr_half* dx2, lr_stream_t s) { return lr_groupnorm_bwd_t<bf16>(x1, C1, x2, C2, dy, N, HW, fwd_partials, gamma, beta, eps, silu, bwd_partials, dx1, dx2, s, fwd_chunks, dres1, dres2); }
