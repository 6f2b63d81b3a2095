// Streaming kernels at the edges of the UNet step: layout converters, timestep embedding, the time-MLP
// (small-M linear), the re-arranged multi-view token gather/scatter and the fused CFG + DDIM update.
// All are HBM/launch bound; every global access is vectorised and coalesced.
#include "common.h"

// ---------------------------------------------------------------------------------------------------------------
// NCHW fp32 -> NHWC fp16 (channel-padded).  One thread per (pixel, octet of output channels): the reads of one
// channel plane are coalesced across the 64 lanes of a wave (consecutive pixels), the 16-byte writes land in the
// pixel's row.  Cpad is a multiple of 8.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x1, int C1, const float* __restrict__ x2, int C2,
                                    T* __restrict__ y, int Cpad, int HW, long long total) {
  const int nOct = Cpad >> 3;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    // idx = (n * nOct + o) * HW + p   (pixel fastest => coalesced plane reads)
    const int p = (int)(idx % HW);
    const long long q = idx / HW;
    const int o = (int)(q % nOct);
    const long long n = q / nOct;
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = o * 8 + i;
      float v = 0.f;
      if (c < C1) v = x1[((size_t)n * C1 + c) * HW + p];
      else if (c < C1 + C2) v = x2[((size_t)n * C2 + (c - C1)) * HW + p];
      f[i] = v;
    }
    *reinterpret_cast<uint4*>(y + ((size_t)n * HW + p) * Cpad + o * 8) = lr_pack8<T>(f);
  }
}

template <typename OutT, typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ y, int Cstride, int C, OutT* __restrict__ out, int HW,
                                    long long total) {
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int p = (int)(idx % HW);
    const long long q = idx / HW;
    const int c = (int)(q % C);
    const long long n = q / C;
    out[idx] = (OutT)(float)y[((size_t)n * HW + p) * Cstride + c];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// timestep_embedding (util.py:154-174): out[n][0:half] = cos(t * f_j), out[n][half:] = sin(t * f_j),
// f_j = exp(-ln(10000) * j / half), all in fp32 with the exact libm-class functions (t up to 981 rad: no fast-math).
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void timestep_embedding_kernel(const int64_t* __restrict__ t, int N, int dim, T* __restrict__ out) {
  const int half = dim >> 1;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * half) return;
  const int n = idx / half, j = idx % half;
  const float freq = expf(-logf(10000.0f) * (float)j / (float)half);
  const float a = (float)t[n] * freq;
  out[(size_t)n * dim + j] = (T)cosf(a);
  out[(size_t)n * dim + half + j] = (T)sinf(a);
  if ((dim & 1) && j == 0) out[(size_t)n * dim + dim - 1] = (T)0.f;
}

// ---------------------------------------------------------------------------------------------------------------
// Small-M linear: out[m][n] = act_out(sum_k act_in(a[m][k]) w[n][k] + b[n]), M <= 16.  Weight-streaming bound:
// each wave owns output columns n, lanes split K in 16-byte pieces (coalesced 1 KiB per wave load), the (tiny)
// activation matrix is staged once per block in LDS with act_in applied.
// ---------------------------------------------------------------------------------------------------------------
template <int MMAX, typename T>
__global__ void linear_small_m_kernel(const T* __restrict__ a, int lda, const T* __restrict__ w,
                                      const float* __restrict__ bias, T* __restrict__ out, int ldo, int M, int N,
                                      int K, int act_in, int act_out, int cols_per_wave) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* s_a = reinterpret_cast<T*>(smem_raw);  // [MMAX][K]
  const int t = threadIdx.x;
  const int K8 = K >> 3;
  if ((lda & 7) == 0 && (((uintptr_t)a) & 15) == 0) {      // 16-byte loads (round 6: the scalar staging loop was most of a batch-16 launch)
    for (int idx = t; idx < MMAX * K8; idx += blockDim.x) {
      const int m = idx / K8, k = (idx - m * K8) * 8;
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (m < M) {
        lr_unpack8<T>(*reinterpret_cast<const uint4*>(a + (size_t)m * lda + k), v);
        if (act_in) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = lr_silu(v[i]);
        }
      }
      *reinterpret_cast<uint4*>(s_a + m * K + k) = lr_pack8<T>(v);
    }
  } else {
    for (int idx = t; idx < MMAX * K; idx += blockDim.x) {
      const int m = idx / K, k = idx % K;
      float v = 0.f;
      if (m < M) {
        v = (float)a[(size_t)m * lda + k];
        if (act_in) v = lr_silu(v);
      }
      s_a[idx] = (T)v;
    }
  }
  __syncthreads();
  const int lane = t & 63, wave = t >> 6;
  const int nwaves = blockDim.x >> 6;
  const int n_begin = (blockIdx.x * nwaves + wave) * cols_per_wave;
  for (int n = n_begin; n < min(N, n_begin + cols_per_wave); ++n) {
    float acc[MMAX];
#pragma unroll
    for (int m = 0; m < MMAX; ++m) acc[m] = 0.f;
    for (int k = lane * 8; k < K; k += 64 * 8) {
      float wf[8];
      lr_unpack8<T>(*reinterpret_cast<const uint4*>(w + (size_t)n * K + k), wf);
#pragma unroll
      for (int m = 0; m < MMAX; ++m) {
        float af[8];
        lr_unpack8<T>(*reinterpret_cast<const uint4*>(s_a + m * K + k), af);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[m] = fmaf(af[i], wf[i], acc[m]);
      }
    }
#pragma unroll
    for (int m = 0; m < MMAX; ++m) acc[m] = lr_wave_sum(acc[m]);
    if (lane < M) {
      float v = 0.f;
#pragma unroll
      for (int m = 0; m < MMAX; ++m) if (m == lane) v = acc[m];
      if (bias) v += bias[n];
      if (act_out) v = lr_silu(v);
      out[(size_t)lane * ldo + n] = (T)v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Multi-view token re-arrangement, concat_target=True (multiview_attention.py:440-446 / 456-460).
// canvases x [b*v][s rows][2s cols][C]; sequence seq [b][(v+1)][s][s][C] = [target(from canvas 0), ref_0..ref_{v-1}].
// ---------------------------------------------------------------------------------------------------------------
// SUM = true is the backward of mv_scatter: the target slot collects the right halves of ALL canvases (fp32 sum).
template <bool SUM, typename T>
__global__ void mv_gather_kernel(const uint4* __restrict__ x, uint4* __restrict__ seq, int b, int v, int s, int C8,
                                 long long total) {
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C8);
    long long q = idx / C8;
    const int col = (int)(q % s); q /= s;
    const int row = (int)(q % s); q /= s;
    const int j = (int)(q % (v + 1));
    const long long bi = q / (v + 1);
    const int canvas = j == 0 ? 0 : j - 1;
    const int scol = j == 0 ? s + col : col;
    if (SUM && j == 0) {
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int cv = 0; cv < v; ++cv) {
        float f[8];
        lr_unpack8<T>(x[((((size_t)bi * v + cv) * s + row) * (2 * s) + scol) * C8 + c], f);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += f[i];
      }
      seq[idx] = lr_pack8<T>(acc);
    } else {
      seq[idx] = x[((((size_t)bi * v + canvas) * s + row) * (2 * s) + scol) * C8 + c];
    }
  }
}

// ZERO = true is the backward of mv_gather: only canvas 0 contributed its right half, the others receive zero there.
template <bool ZERO>
__global__ void mv_scatter_kernel(const uint4* __restrict__ seq, uint4* __restrict__ x, int b, int v, int s, int C8,
                                  long long total) {
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C8);
    long long q = idx / C8;
    const int col = (int)(q % (2 * s)); q /= (2 * s);
    const int row = (int)(q % s); q /= s;
    const int canvas = (int)(q % v);
    const long long bi = q / v;
    const int j = col >= s ? 0 : canvas + 1;
    const int scol = col >= s ? col - s : col;
    if (ZERO && col >= s && canvas != 0) x[idx] = make_uint4(0, 0, 0, 0);
    else x[idx] = seq[((((size_t)bi * (v + 1) + j) * s + row) * s + scol) * C8 + c];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Row copies with optional index tables (lr_row_copy): the glue of the canvas-sharded multi-view block -- packing rows + their
// LayerNorm statistics into one message, unpacking a received message into sequence order / this rank's own rows, writing the
// canvas back -- as ONE launch of up to four jobs instead of a dozen torch slice / cat / copy kernels.  8-byte granules.
// ---------------------------------------------------------------------------------------------------------------
struct RowCopyJobs { lr_row_copy_job j[4]; };
__global__ void row_copy_kernel(const RowCopyJobs J) {
  const lr_row_copy_job& jb = J.j[blockIdx.y];
  const int g8 = jb.row_bytes >> 3;                       // 8-byte granules per row
  const long long total = (long long)jb.n_rows * g8;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(idx / g8), g = (int)(idx - (long long)r * g8);
    const long long sr = jb.src_idx ? jb.src_idx[r] : r, dr = jb.dst_idx ? jb.dst_idx[r] : r;
    const uint2 v = *reinterpret_cast<const uint2*>((const char*)jb.src + sr * jb.src_pitch + jb.src_off + (long long)g * 8);
    *reinterpret_cast<uint2*>((char*)jb.dst + dr * jb.dst_pitch + jb.dst_off + (long long)g * 8) = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// CFG combine + DDIM update (ddim.py:343-381), fp32 state, 4 elements per thread.
// ---------------------------------------------------------------------------------------------------------------
template <typename EpsT, typename T>
__global__ void ddim_cfg_step_kernel(const float* __restrict__ x, const EpsT* __restrict__ eps,
                                     const float* __restrict__ noise, float* __restrict__ x_prev,
                                     float* __restrict__ pred_x0, long long numel, float scale, float sqrt_at,
                                     float sqrt_1m_at, float sqrt_aprev, float dir_coef, float sigma) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < numel;
       i += (long long)gridDim.x * blockDim.x) {
    const float eu = (float)eps[i];
    const float ec = (float)eps[numel + i];
    float e;
    if (sizeof(EpsT) == 2) {
      // the reference's CFG combine runs in fp16 (model output dtype under autocast, ddim.py:343)
      const T d = (T)(ec - eu);
      const T sd = (T)(scale * (float)d);
      e = (float)(T)(eu + (float)sd);
    } else {
      e = eu + scale * (ec - eu);
    }
    const float xv = x[i];
    const float p0 = (xv - sqrt_1m_at * e) / sqrt_at;
    float xp = sqrt_aprev * p0 + dir_coef * e;
    if (noise) xp += sigma * noise[i];
    pred_x0[i] = p0;
    x_prev[i] = xp;
  }
}

static inline int grid_for(long long total, int block, int cap = 4096) {
  long long g = (total + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" int lr_abi_version(void) { return 26; }

#ifdef LR_DEV_VARIANTS
// developer build only: name -> value table behind LR_DEV (common.h); set through lr_dev_set by the Python front end
#include <map>
#include <mutex>
#include <string>
static std::map<std::string, int>& lr_dev_table() { static std::map<std::string, int> t; return t; }
static std::mutex lr_dev_mutex;
extern "C" int lr_dev_set(const char* name, int value) {
  if (!name) return LR_E_ARG;
  std::lock_guard<std::mutex> g(lr_dev_mutex);
  lr_dev_table()[name] = value;
  return 0;
}
extern "C" int lr_dev_unset(const char* name) {
  if (!name) return LR_E_ARG;
  std::lock_guard<std::mutex> g(lr_dev_mutex);
  lr_dev_table().erase(name);
  return 0;
}
int lr_dev_get(const char* name, int dflt) {
  std::lock_guard<std::mutex> g(lr_dev_mutex);
  const auto it = lr_dev_table().find(name);
  return it == lr_dev_table().end() ? dflt : it->second;
}
#endif

template <typename T>
static int lr_nchw_f32_to_nhwc_t(const float* x1, int C1, const float* x2, int C2, lr_half* y, int Cpad, int N,
                                       int H, int W, lr_stream_t s) {
  if (!x1 || !y || N <= 0 || H <= 0 || W <= 0 || C1 <= 0) return LR_E_ARG;
  if (!x2) C2 = 0;
  if (Cpad % 8 || Cpad < C1 + C2) return LR_E_ALIGN;
  const long long total = (long long)N * (Cpad / 8) * H * W;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel<T>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)s, x1, C1, x2, C2,
                     (T*)y, Cpad, H * W, total);
  return lr_launch_status();
}

template <typename T>
static int lr_nhwc_f16_to_nchw_t(const lr_half* y, int Cstride, int C, void* out, int out_is_f32, int N, int H, int W,
                                   lr_stream_t s) {
  if (!y || !out || N <= 0 || C <= 0 || C > Cstride) return LR_E_ARG;
  const long long total = (long long)N * C * H * W;
  if (out_is_f32)
    hipLaunchKernelGGL((nhwc_to_nchw_kernel<float, T>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)s,
                       (const T*)y, Cstride, C, (float*)out, H * W, total);
  else
    hipLaunchKernelGGL((nhwc_to_nchw_kernel<T, T>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)s,
                       (const T*)y, Cstride, C, (T*)out, H * W, total);
  return lr_launch_status();
}

template <typename T>
static int lr_timestep_embedding_t(const int64_t* t, int N, int dim, lr_half* out, lr_stream_t s) {
  if (!t || !out || N <= 0 || dim < 2) return LR_E_ARG;
  const int total = N * (dim / 2);
  hipLaunchKernelGGL(timestep_embedding_kernel<T>, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)s, t, N, dim,
                     (T*)out);
  return lr_launch_status();
}

template <typename T>
static int lr_linear_small_m_t(const lr_half* a, int lda, const lr_half* w, const float* bias, lr_half* out, int ldo,
                                 int M, int N, int K, int act_in, int act_out, lr_stream_t s) {
  if (!a || !w || !out || M <= 0 || N <= 0 || K <= 0) return LR_E_ARG;
  if (M > 16) return LR_E_UNSUPPORTED;
  if (K % 8) return LR_E_ALIGN;
  // columns per wave: every block stages the M activation rows once, so wide layers take more columns per block (about two blocks per
  // CU); a column's arithmetic does not depend on it (one lane-strided K loop + one wave sum per column): same bits for every value
  const int waves = 4;
  int cpw = (N + waves * 512 - 1) / (waves * 512);
  if (cpw < 4) cpw = 4;
  if (cpw > 32) cpw = 32;
  dim3 grid((N + cpw * waves - 1) / (cpw * waves)), block(64 * waves);
  hipStream_t st = (hipStream_t)s;
  if (M <= 4)
    hipLaunchKernelGGL((linear_small_m_kernel<4, T>), grid, block, 4 * K * sizeof(T), st, (const T*)a, lda,
                       (const T*)w, bias, (T*)out, ldo, M, N, K, act_in, act_out, cpw);
  else if (M <= 8)
    hipLaunchKernelGGL((linear_small_m_kernel<8, T>), grid, block, 8 * K * sizeof(T), st, (const T*)a, lda,
                       (const T*)w, bias, (T*)out, ldo, M, N, K, act_in, act_out, cpw);
  else
    hipLaunchKernelGGL((linear_small_m_kernel<16, T>), grid, block, 16 * K * sizeof(T), st, (const T*)a, lda,
                       (const T*)w, bias, (T*)out, ldo, M, N, K, act_in, act_out, cpw);
  return lr_launch_status();
}

template <typename T>
static int lr_mv_gather_t(const lr_half* x, lr_half* seq, int b, int v, int s, int C, lr_stream_t st) {
  if (!x || !seq || b <= 0 || v <= 0 || s <= 0) return LR_E_ARG;
  if (C % 8) return LR_E_ALIGN;
  const long long total = (long long)b * (v + 1) * s * s * (C / 8);
  hipLaunchKernelGGL((mv_gather_kernel<false, T>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)st, (const uint4*)x,
                     (uint4*)seq, b, v, s, C / 8, total);
  return lr_launch_status();
}

template <typename T>
static int lr_mv_scatter_t(const lr_half* seq, lr_half* x, int b, int v, int s, int C, lr_stream_t st) {
  if (!x || !seq || b <= 0 || v <= 0 || s <= 0) return LR_E_ARG;
  if (C % 8) return LR_E_ALIGN;
  const long long total = (long long)b * v * s * 2 * s * (C / 8);
  hipLaunchKernelGGL(mv_scatter_kernel<false>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)st, (const uint4*)seq,
                     (uint4*)x, b, v, s, C / 8, total);
  return lr_launch_status();
}

template <typename T>
static int lr_ddim_cfg_step_t(const float* x, const void* eps, int eps_is_f32, const float* noise, float* x_prev,
                                float* pred_x0, int64_t numel, float cfg_scale, float a_t, float a_prev, float sigma_t,
                                float sqrt_one_minus_at, lr_stream_t s) {
  if (!x || !eps || !x_prev || !pred_x0 || numel <= 0) return LR_E_ARG;
  // same fp32 scalar arithmetic as the reference's 0-dim fp32 tensors (ddim.py:359-381)
  const float sqrt_at = sqrtf(a_t);
  const float sqrt_aprev = sqrtf(a_prev);
  const float dir_coef = sqrtf(1.0f - a_prev - sigma_t * sigma_t);
  dim3 grid(grid_for(numel, 256)), block(256);
  if (eps_is_f32)
    hipLaunchKernelGGL((ddim_cfg_step_kernel<float, T>), grid, block, 0, (hipStream_t)s, x, (const float*)eps, noise, x_prev,
                       pred_x0, (long long)numel, cfg_scale, sqrt_at, sqrt_one_minus_at, sqrt_aprev, dir_coef,
                       sigma_t);
  else
    hipLaunchKernelGGL((ddim_cfg_step_kernel<T, T>), grid, block, 0, (hipStream_t)s, x, (const T*)eps, noise, x_prev,
                       pred_x0, (long long)numel, cfg_scale, sqrt_at, sqrt_one_minus_at, sqrt_aprev, dir_coef,
                       sigma_t);
  return lr_launch_status();
}


// =====================================================================================================================
// Backward helpers (training with frozen weights)
// =====================================================================================================================
// GEGLU backward.  pre [M][2H]: the projection (+bias) in the packed layout of lr_gemm_conv_f16 (16-column groups
// [u16 | g16 | u16 | g16 ...]); dy [M][H];  dpre (same layout as pre): du = dy * gelu(g), dg = dy * u * gelu'(g),
// gelu'(g) = Phi(g) + g * phi(g)  (erf form, attention.py:56-58).  One thread per 8 output columns.
template <typename T>
__global__ void geglu_bwd_kernel(const T* __restrict__ pre, const T* __restrict__ dy, T* __restrict__ dpre, long long total,
                                 int H) {
  const int cpr = H >> 3;      // 8-column chunks per row of dy
  for (long long id = blockIdx.x * (long long)blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
    const long long m = id / cpr;
    const int c = (int)(id - m * cpr) * 8;            // first output column of this chunk
    const int pc = (c >> 4) * 32 + (c & 15);          // packed column of u; g sits 16 columns later
    float u[8], g[8], d[8], du[8], dg[8];
    lr_unpack8<T>(*reinterpret_cast<const uint4*>(pre + m * 2 * H + pc), u);
    lr_unpack8<T>(*reinterpret_cast<const uint4*>(pre + m * 2 * H + pc + 16), g);
    lr_unpack8<T>(*reinterpret_cast<const uint4*>(dy + m * H + c), d);
#pragma unroll
    for (int i = 0; i < 8; i += 2) {      // two values per instruction stream (packed fp32), Phi from the forward's own formula
      const f32x2_t gg = {g[i], g[i + 1]}, dd = {d[i], d[i + 1]}, uu = {u[i], u[i + 1]};
      const f32x2_t ph = lr_phi_mhalf2(gg);                                                    // Phi(g) - 0.5
      const f32x2_t q = gg * gg * -0.72134752044448170368f;                                    // -g^2 / 2 in log2 units
      const f32x2_t pdf = (f32x2_t){__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])} * 0.3989422804014327f;
      const f32x2_t gelu = __builtin_elementwise_fma(gg, ph, gg * 0.5f);                       // == lr_gelu_erf2(g)
      const f32x2_t dgel = __builtin_elementwise_fma(gg, pdf, ph + 0.5f);                      // Phi + g phi
      const f32x2_t a = dd * gelu, b = dd * uu * dgel;
      du[i] = a[0]; du[i + 1] = a[1];
      dg[i] = b[0]; dg[i + 1] = b[1];
    }
    *reinterpret_cast<uint4*>(dpre + m * 2 * H + pc) = lr_pack8<T>(du);
    *reinterpret_cast<uint4*>(dpre + m * 2 * H + pc + 16) = lr_pack8<T>(dg);
  }
}

// GEGLU forward from the stored projection (training keeps `pre` for the backward instead of recomputing the GEMM):
// out[m][c] = u * gelu_erf(g), same packed layout of pre as above.
template <typename T>
__global__ void geglu_fwd_kernel(const T* __restrict__ pre, T* __restrict__ out, long long total, int H) {
  const int cpr = H >> 3;
  for (long long id = blockIdx.x * (long long)blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
    const long long m = id / cpr;
    const int c = (int)(id - m * cpr) * 8;
    const int pc = (c >> 4) * 32 + (c & 15);
    float u[8], g[8], o[8];
    lr_unpack8<T>(*reinterpret_cast<const uint4*>(pre + m * 2 * H + pc), u);
    lr_unpack8<T>(*reinterpret_cast<const uint4*>(pre + m * 2 * H + pc + 16), g);
#pragma unroll
    for (int i = 0; i < 8; i += 2) {      // (packed pairs: same formula, same bits as the scalar form)
      const f32x2_t ge = lr_gelu_erf2((f32x2_t){g[i], g[i + 1]});
      o[i] = u[i] * ge[0];
      o[i + 1] = u[i + 1] * ge[1];
    }
    *reinterpret_cast<uint4*>(out + m * H + c) = lr_pack8<T>(o);
  }
}

template <typename T>
static int lr_geglu_fwd_t(const lr_half* pre, lr_half* out, int M, int H, lr_stream_t s) {
  if (!pre || !out || M <= 0 || H <= 0) return LR_E_ARG;
  if (H % 16) return LR_E_ALIGN;
  const long long total = (long long)M * (H / 8);
  long long blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(geglu_fwd_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, (const T*)pre, (T*)out, total, H);
  return lr_launch_status();
}

template <typename T>
static int lr_geglu_bwd_t(const lr_half* pre, const lr_half* dy, lr_half* dpre, int M, int H, lr_stream_t s) {
  if (!pre || !dy || !dpre || M <= 0 || H <= 0) return LR_E_ARG;
  if (H % 16) return LR_E_ALIGN;
  const long long total = (long long)M * (H / 8);
  long long blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(geglu_bwd_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, (const T*)pre, (const T*)dy,
                     (T*)dpre, total, H);
  return lr_launch_status();
}

// Backward of the nearest-2x upsample in front of a conv (Upsample.forward, openaimodel.py:115): the four fine pixels
// of a coarse pixel add up.  x [N][2H][2W][C] -> y [N][H][W][C].
template <typename T>
__global__ void sumpool2x2_kernel(const T* __restrict__ x, T* __restrict__ y, long long total, int H, int W, int C) {
  const int cpr = C >> 3;
  for (long long id = blockIdx.x * (long long)blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(id % cpr) * 8;
    long long pix = id / cpr;
    const int xw = (int)(pix % W);
    pix /= W;
    const int yh = (int)(pix % H);
    const long long n = pix / H;
    const T* src = x + ((n * 2 * H + 2 * yh) * 2 * W + 2 * xw) * (long long)C + c;
    float a[8], acc[8];
    lr_unpack8<T>(*reinterpret_cast<const uint4*>(src), acc);
    lr_unpack8<T>(*reinterpret_cast<const uint4*>(src + C), a);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += a[i];
    lr_unpack8<T>(*reinterpret_cast<const uint4*>(src + 2LL * W * C), a);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += a[i];
    lr_unpack8<T>(*reinterpret_cast<const uint4*>(src + 2LL * W * C + C), a);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += a[i];
    *reinterpret_cast<uint4*>(y + ((n * H + yh) * W + xw) * (long long)C + c) = lr_pack8<T>(acc);
  }
}

template <typename T>
static int lr_sumpool2x2_t(const lr_half* x, lr_half* y, int N, int H, int W, int C, lr_stream_t s) {
  if (!x || !y || N <= 0 || H <= 0 || W <= 0) return LR_E_ARG;
  if (C % 8) return LR_E_ALIGN;
  const long long total = (long long)N * H * W * (C / 8);
  long long blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(sumpool2x2_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, (const T*)x, (T*)y, total, H, W, C);
  return lr_launch_status();
}

// Backward of the multi-view re-arrangement (same shapes as lr_mv_gather / lr_mv_scatter, roles of x and seq swapped)
template <typename T>
static int lr_mv_gather_bwd_t(const lr_half* dseq, lr_half* dx, int b, int v, int s, int C, lr_stream_t st) {
  if (!dx || !dseq || b <= 0 || v <= 0 || s <= 0) return LR_E_ARG;
  if (C % 8) return LR_E_ALIGN;
  const long long total = (long long)b * v * s * 2 * s * (C / 8);
  hipLaunchKernelGGL(mv_scatter_kernel<true>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)st, (const uint4*)dseq,
                     (uint4*)dx, b, v, s, C / 8, total);
  return lr_launch_status();
}

template <typename T>
static int lr_mv_scatter_bwd_t(const lr_half* dx, lr_half* dseq, int b, int v, int s, int C, lr_stream_t st) {
  if (!dx || !dseq || b <= 0 || v <= 0 || s <= 0) return LR_E_ARG;
  if (C % 8) return LR_E_ALIGN;
  const long long total = (long long)b * (v + 1) * s * s * (C / 8);
  hipLaunchKernelGGL((mv_gather_kernel<true, T>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)st, (const uint4*)dx,
                     (uint4*)dseq, b, v, s, C / 8, total);
  return lr_launch_status();
}

// ---- C ABI: every entry point in its fp16 and bf16 form -------------------------------------------------------------
extern "C" int lr_nchw_f32_to_nhwc_f16(const float* x1, int C1, const float* x2, int C2, lr_half* y, int Cpad, int N, int H, int W, lr_stream_t s) { return lr_nchw_f32_to_nhwc_t<f16>(x1, C1, x2, C2, y, Cpad, N, H, W, s); }
extern "C" int lr_nchw_f32_to_nhwc_bf16(const float* x1, int C1, const float* x2, int C2, lr_half* y, int Cpad, int N, int H, int W, lr_stream_t s) { return lr_nchw_f32_to_nhwc_t<bf16>(x1, C1, x2, C2, y, Cpad, N, H, W, s); }
extern "C" int lr_nhwc_f16_to_nchw(const lr_half* y, int Cstride, int C, void* out, int out_is_f32, int N, int H, int W, lr_stream_t s) { return lr_nhwc_f16_to_nchw_t<f16>(y, Cstride, C, out, out_is_f32, N, H, W, s); }
extern "C" int lr_nhwc_f16_to_nchw_bf16(const lr_half* y, int Cstride, int C, void* out, int out_is_f32, int N, int H, int W, lr_stream_t s) { return lr_nhwc_f16_to_nchw_t<bf16>(y, Cstride, C, out, out_is_f32, N, H, W, s); }
extern "C" int lr_timestep_embedding(const int64_t* t, int N, int dim, lr_half* out, lr_stream_t s) { return lr_timestep_embedding_t<f16>(t, N, dim, out, s); }
extern "C" int lr_timestep_embedding_bf16(const int64_t* t, int N, int dim, lr_half* out, lr_stream_t s) { return lr_timestep_embedding_t<bf16>(t, N, dim, out, s); }
extern "C" int lr_linear_small_m(const lr_half* a, int lda, const lr_half* w, const float* bias, lr_half* out, int ldo, int M, int N, int K, int act_in, int act_out, lr_stream_t s) { return lr_linear_small_m_t<f16>(a, lda, w, bias, out, ldo, M, N, K, act_in, act_out, s); }
extern "C" int lr_linear_small_m_bf16(const lr_half* a, int lda, const lr_half* w, const float* bias, lr_half* out, int ldo, int M, int N, int K, int act_in, int act_out, lr_stream_t s) { return lr_linear_small_m_t<bf16>(a, lda, w, bias, out, ldo, M, N, K, act_in, act_out, s); }
extern "C" int lr_mv_gather(const lr_half* x, lr_half* seq, int b, int v, int s, int C, lr_stream_t st) { return lr_mv_gather_t<f16>(x, seq, b, v, s, C, st); }
extern "C" int lr_mv_gather_bf16(const lr_half* x, lr_half* seq, int b, int v, int s, int C, lr_stream_t st) { return lr_mv_gather_t<bf16>(x, seq, b, v, s, C, st); }
extern "C" int lr_mv_scatter(const lr_half* seq, lr_half* x, int b, int v, int s, int C, lr_stream_t st) { return lr_mv_scatter_t<f16>(seq, x, b, v, s, C, st); }
extern "C" int lr_mv_scatter_bf16(const lr_half* seq, lr_half* x, int b, int v, int s, int C, lr_stream_t st) { return lr_mv_scatter_t<bf16>(seq, x, b, v, s, C, st); }
extern "C" int lr_ddim_cfg_step(const float* x, const void* eps, int eps_is_f32, const float* noise, float* x_prev, float* pred_x0, int64_t numel, float cfg_scale, float a_t, float a_prev, float sigma_t, float sqrt_one_minus_at, lr_stream_t s) { return lr_ddim_cfg_step_t<f16>(x, eps, eps_is_f32, noise, x_prev, pred_x0, numel, cfg_scale, a_t, a_prev, sigma_t, sqrt_one_minus_at, s); }
extern "C" int lr_ddim_cfg_step_bf16(const float* x, const void* eps, int eps_is_f32, const float* noise, float* x_prev, float* pred_x0, int64_t numel, float cfg_scale, float a_t, float a_prev, float sigma_t, float sqrt_one_minus_at, lr_stream_t s) { return lr_ddim_cfg_step_t<bf16>(x, eps, eps_is_f32, noise, x_prev, pred_x0, numel, cfg_scale, a_t, a_prev, sigma_t, sqrt_one_minus_at, s); }
extern "C" int lr_geglu_fwd(const lr_half* pre, lr_half* out, int M, int H, lr_stream_t s) { return lr_geglu_fwd_t<f16>(pre, out, M, H, s); }
extern "C" int lr_geglu_fwd_bf16(const lr_half* pre, lr_half* out, int M, int H, lr_stream_t s) { return lr_geglu_fwd_t<bf16>(pre, out, M, H, s); }
extern "C" int lr_geglu_bwd(const lr_half* pre, const lr_half* dy, lr_half* dpre, int M, int H, lr_stream_t s) { return lr_geglu_bwd_t<f16>(pre, dy, dpre, M, H, s); }
extern "C" int lr_geglu_bwd_bf16(const lr_half* pre, const lr_half* dy, lr_half* dpre, int M, int H, lr_stream_t s) { return lr_geglu_bwd_t<bf16>(pre, dy, dpre, M, H, s); }
extern "C" int lr_sumpool2x2(const lr_half* x, lr_half* y, int N, int H, int W, int C, lr_stream_t s) { return lr_sumpool2x2_t<f16>(x, y, N, H, W, C, s); }
extern "C" int lr_sumpool2x2_bf16(const lr_half* x, lr_half* y, int N, int H, int W, int C, lr_stream_t s) { return lr_sumpool2x2_t<bf16>(x, y, N, H, W, C, s); }
extern "C" int lr_mv_gather_bwd(const lr_half* dseq, lr_half* dx, int b, int v, int s, int C, lr_stream_t st) { return lr_mv_gather_bwd_t<f16>(dseq, dx, b, v, s, C, st); }
extern "C" int lr_mv_gather_bwd_bf16(const lr_half* dseq, lr_half* dx, int b, int v, int s, int C, lr_stream_t st) { return lr_mv_gather_bwd_t<bf16>(dseq, dx, b, v, s, C, st); }
extern "C" int lr_mv_scatter_bwd(const lr_half* dx, lr_half* dseq, int b, int v, int s, int C, lr_stream_t st) { return lr_mv_scatter_bwd_t<f16>(dx, dseq, b, v, s, C, st); }
extern "C" int lr_mv_scatter_bwd_bf16(const lr_half* dx, lr_half* dseq, int b, int v, int s, int C, lr_stream_t st) { return lr_mv_scatter_bwd_t<bf16>(dx, dseq, b, v, s, C, st); }
extern "C" int lr_row_copy(const lr_row_copy_job* jobs, int n_jobs, lr_stream_t st) {
  if (!jobs || n_jobs < 1 || n_jobs > 4) return LR_E_ARG;
  RowCopyJobs J;
  long long mx = 0;
  for (int i = 0; i < n_jobs; ++i) {
    const lr_row_copy_job& jb = jobs[i];
    if (!jb.src || !jb.dst || jb.n_rows < 0 || jb.row_bytes <= 0) return LR_E_ARG;
    if ((jb.row_bytes | jb.src_pitch | jb.dst_pitch | jb.src_off | jb.dst_off | (int64_t)(uintptr_t)jb.src | (int64_t)(uintptr_t)jb.dst) & 7) return LR_E_ALIGN;
    J.j[i] = jb;
    const long long t_ = (long long)jb.n_rows * (jb.row_bytes >> 3);
    if (t_ > mx) mx = t_;
  }
  if (mx == 0) return 0;
  hipLaunchKernelGGL(row_copy_kernel, dim3(grid_for(mx, 256), n_jobs), dim3(256), 0, (hipStream_t)st, J);
  return lr_launch_status();
}

