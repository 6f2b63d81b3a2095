// The UNet's `out` block in one launch: GroupNorm(32) -> SiLU -> 3x3 pad-1 conv to <= 4 output channels -> NCHW
// (reference openaimodel.py:714-718: nn.Sequential(normalization(ch), nn.SiLU(), zero_module(conv_nd(dims, model_channels, out_channels, 3, padding=1)))).
//
// Why its own kernel: as an implicit GEMM this conv has N = 4 of a 64-column tile and re-reads the 42 MB input nine times (once per
// tap) for 1.5 GFLOP, behind a GroupNorm pass that reads and writes the same 42 MB: 66 + 21 us at latent 64x128, batch 8.  Here the
// input is read ONCE: a block owns an 8 x 16 pixel tile of one sample, walks the channels in chunks of 64, and per chunk
//   * loads the 10 x 18 halo patch (16-byte pieces), normalises (a[c] x + b[c] from the producer's per-group partial sums, fp64
//     prologue like gn_apply_kernel), applies SiLU, rounds to the 16-bit type exactly where the unfused path stored the GroupNorm
//     output, and writes it to LDS (zero outside the image = the conv's padding);
//   * multiplies it with the chunk's weights on the matrix cores: D[out channel][pixel] += W[out channel][k] X[k][pixel] with
//     v_mfma_f32_16x16x32 (rows 4..15 of the A operand are a zero row: the matrix work is 0.4 us either way), B fragments = one
//     ds_read_b128 per lane (pixel = lane & 15 of a 16-pixel row segment shifted by the tap, 8 channels), conflict-free with the
//     GEMM's row swizzle (slot = chunk ^ ((pixel >> 1) & 7)).
// The patch is double buffered and the global loads run two chunks ahead in two register sets; weights ([Cout + 1 zero row][9 C]) and
// the per-channel (a, b) table sit in LDS for the whole block.  HBM-bound: 2 B per input element + the halo from L2.
#include "common.h"

#define CO_THREADS 256
#define CO_TH 8
#define CO_TW 16
#define CO_HALO ((CO_TH + 2) * (CO_TW + 2))      // 180 patch pixels
#define CO_PIECES (CO_HALO * 8)                  // 16-byte pieces of a 64-channel patch
#define CO_NLOAD ((CO_PIECES + CO_THREADS - 1) / CO_THREADS)
#define CO_PATCH_BYTES (CO_HALO * 128)

struct ConvOutParams {
  const void* x; const float* gpart; const float* gamma; const float* beta; const void* w; const float* bias; void* y;
  int B, H, W, C, chunks, Cout, ldw;
  float eps;
};

template <typename T>
__global__ __launch_bounds__(CO_THREADS) void gn_conv_out_kernel(const ConvOutParams P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float s_mean[32], s_rstd[32];
  const int C = P.C, K = 9 * C;
  // LDS carve: patch[2] | weights [5][K] | a[C] | b[C]
  char* patch = smem;
  T* s_w = reinterpret_cast<T*>(smem + 2 * CO_PATCH_BYTES);
  float* s_a = reinterpret_cast<float*>(smem + 2 * CO_PATCH_BYTES + 5 * K * 2);
  float* s_b = s_a + C;

  const int t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int tiles_x = P.W / CO_TW, tiles_y = P.H / CO_TH;
  const int n = blockIdx.x / (tiles_x * tiles_y);
  const int tr = blockIdx.x % (tiles_x * tiles_y);
  const int Y0 = (tr / tiles_x) * CO_TH, X0 = (tr % tiles_x) * CO_TW;
  const T* x = reinterpret_cast<const T*>(P.x) + (size_t)n * P.H * P.W * C;

  // ---- piece ownership of the patch loader (the same for every chunk): piece q = it * 256 + t -> (patch pixel, 8-channel group)
  int goff[CO_NLOAD];      // element offset of the piece in x (chunk 0), or -1 outside the image / past the patch
  int loff[CO_NLOAD];      // byte offset in the patch buffer
#pragma unroll
  for (int it = 0; it < CO_NLOAD; ++it) {
    const int q = it * CO_THREADS + t;
    const int pix = q >> 3, c8 = q & 7;
    const int hy = pix / (CO_TW + 2), hx = pix % (CO_TW + 2);
    const int gy = Y0 + hy - 1, gx = X0 + hx - 1;
    const bool in = q < CO_PIECES && gy >= 0 && gy < P.H && gx >= 0 && gx < P.W;
    goff[it] = in ? (gy * P.W + gx) * C + c8 * 8 : -1;
    loff[it] = q < CO_PIECES ? pix * 128 + ((c8 ^ ((pix >> 1) & 7)) << 4) : -1;
  }
  // two register sets: the loads of chunk c + 2 are issued before chunk c + 1 is converted, so a chunk's global latency hides behind a
  // whole iteration (MFMAs of chunk c + SiLU / LDS writes of chunk c + 1)
  uint4 rega[CO_NLOAD], regb[CO_NLOAD];
  auto load = [&](int chunk, uint4 (&regs)[CO_NLOAD]) {
#pragma unroll
    for (int it = 0; it < CO_NLOAD; ++it)
      regs[it] = goff[it] >= 0 ? *reinterpret_cast<const uint4*>(x + goff[it] + chunk * 64) : make_uint4(0, 0, 0, 0);
  };
  auto store = [&](int chunk, int buf, const uint4 (&regs)[CO_NLOAD]) {
    char* dst = patch + buf * CO_PATCH_BYTES;
#pragma unroll
    for (int it = 0; it < CO_NLOAD; ++it) {
      if (loff[it] < 0) continue;
      uint4 o = make_uint4(0, 0, 0, 0);
      if (goff[it] >= 0) {
        const int c0 = chunk * 64 + (((it * CO_THREADS + t) & 7) << 3);
        float f[8];
        lr_unpack8<T>(regs[it], f);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float z = fmaf(f[i], s_a[c0 + i], s_b[c0 + i]);
          f[i] = z * __builtin_amdgcn_rcpf(1.0f + __expf(-z));
        }
        o = lr_pack8<T>(f);
      }
      *reinterpret_cast<uint4*>(dst + loff[it]) = o;
    }
  };

  const int nchunk = C / 64;
  load(0, rega);      // in flight under the prologue
  if (nchunk > 1) load(1, regb);

  // ---- GroupNorm statistics of sample n from the chunk partials [B][chunks][32][2] (fp64, fixed order: as gn_apply_kernel)
  const int Cg = C / 32;
  if (t < 128) {
    const int g = t >> 2, sub = t & 3;
    double s = 0.0, q = 0.0;
    const float* ps = P.gpart + ((size_t)n * P.chunks * 32 + g) * 2;
    for (int c = sub; c < P.chunks; c += 4) { s += (double)ps[c * 64]; q += (double)ps[c * 64 + 1]; }
#pragma unroll
    for (int sh = 2; sh > 0; sh >>= 1) { s += __shfl_xor(s, sh, 64); q += __shfl_xor(q, sh, 64); }
    if (sub == 0) {
      const double cnt = (double)P.H * (double)P.W * (double)Cg;
      const double mean = s / cnt;
      double var = q / cnt - mean * mean;
      if (var < 0.0) var = 0.0;
      s_mean[g] = (float)mean;
      s_rstd[g] = (float)(1.0 / sqrt(var + (double)P.eps));
    }
  }
  // ---- weights: rows 0 .. Cout-1 of w ([*][ldw], K = tap * C + channel), rows Cout .. 4 zero
  {
    const T* w = reinterpret_cast<const T*>(P.w);
    const int per_row = K / 8;
    for (int q = t; q < 5 * per_row; q += CO_THREADS) {
      const int r = q / per_row, k8 = q % per_row;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (r < P.Cout) v = *reinterpret_cast<const uint4*>(w + (size_t)r * P.ldw + k8 * 8);
      *reinterpret_cast<uint4*>(s_w + (size_t)r * K + k8 * 8) = v;
    }
  }
  __syncthreads();
  for (int c = t; c < C; c += CO_THREADS) {
    const int g = c / Cg;
    const float a = s_rstd[g] * P.gamma[c];
    s_a[c] = a;
    s_b[c] = P.beta[c] - s_mean[g] * a;
  }
  __syncthreads();
  store(0, 0, rega);
  __syncthreads();

  // ---- main loop: wave wv owns tile rows 2 wv, 2 wv + 1 (two 16-pixel segments)
  const int fr = lane & 15, fq = lane >> 4;
  const T* wrow = s_w + (size_t)(fr < P.Cout ? fr : 4) * K + fq * 8;
  f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  auto mma = [&](int c) {
    const char* src = patch + (c & 1) * CO_PATCH_BYTES;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap % 3;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const vec8<T> wf = *reinterpret_cast<const vec8<T>*>(wrow + tap * C + c * 64 + ks * 32);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int pix = (2 * wv + g + ky) * (CO_TW + 2) + fr + kx;
          const int kc = ks * 4 + fq;
          const vec8<T> xf = *reinterpret_cast<const vec8<T>*>(src + pix * 128 + ((kc ^ ((pix >> 1) & 7)) << 4));
          acc[g] = lr_mfma16(wf, xf, acc[g]);
        }
      }
    }
  };
  // chunk c + 1 sits in regb when c is even, in rega when c is odd; the set that held chunk c is refilled with chunk c + 2
  for (int c = 0; c < nchunk; c += 2) {
    if (c + 2 < nchunk) load(c + 2, rega);
    mma(c);
    if (c + 1 < nchunk) store(c + 1, 1, regb);
    __syncthreads();
    if (c + 1 >= nchunk) break;
    if (c + 3 < nchunk) load(c + 3, regb);
    mma(c + 1);
    if (c + 2 < nchunk) store(c + 2, 0, rega);
    __syncthreads();
  }

  // ---- epilogue: lanes 0..15 hold out channels 0..3 of pixel (row 2 wv + g, column lane); NCHW store
  if (fq == 0) {
    T* y = reinterpret_cast<T*>(P.y) + (size_t)n * P.Cout * P.H * P.W;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < P.Cout)
          y[((size_t)i * P.H + Y0 + 2 * wv + g) * P.W + X0 + fr] = (T)(acc[g][i] + (P.bias ? P.bias[i] : 0.f));
  }
}

template <typename T>
static int lr_gn_conv_out_t(const lr_half* x, int B, int H, int W, int C, const float* gpart, int chunks, const float* gamma,
                            const float* beta, float eps, const lr_half* w, int ldw, const float* bias, int Cout, lr_half* y, lr_stream_t s) {
  if (!x || !gpart || !gamma || !beta || !w || !y || B <= 0 || H <= 0 || W <= 0 || chunks <= 0) return LR_E_ARG;
  if (Cout < 1 || Cout > 4 || C % 64 || C < 64 || H % CO_TH || W % CO_TW) return LR_E_UNSUPPORTED;
  if (ldw % 8 || ldw < 9 * C || (((uintptr_t)x | (uintptr_t)w) & 15)) return LR_E_ALIGN;
  if ((size_t)H * W * C >= (1ull << 31)) return LR_E_UNSUPPORTED;      // 32-bit element offsets inside a sample
  const size_t lds = 2 * CO_PATCH_BYTES + (size_t)5 * 9 * C * 2 + 2 * (size_t)C * 4;
  if (lds > 150 * 1024) return LR_E_UNSUPPORTED;
  static unsigned long long attr_done = 0;
  if (lr_attr_needed(&attr_done))
    hipFuncSetAttribute(reinterpret_cast<const void*>(gn_conv_out_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  ConvOutParams P;
  P.x = x; P.gpart = gpart; P.gamma = gamma; P.beta = beta; P.w = w; P.bias = bias; P.y = y;
  P.B = B; P.H = H; P.W = W; P.C = C; P.chunks = chunks; P.Cout = Cout; P.ldw = ldw; P.eps = eps;
  hipLaunchKernelGGL(gn_conv_out_kernel<T>, dim3(B * (H / CO_TH) * (W / CO_TW)), dim3(CO_THREADS), lds, (hipStream_t)s, P);
  return lr_launch_status();
}

extern "C" int lr_gn_conv_out_f16(const lr_half* x, int B, int H, int W, int C, const float* gpart, int chunks, const float* gamma, const float* beta, float eps, const lr_half* w, int ldw, const float* bias, int Cout, lr_half* y, lr_stream_t s) { return lr_gn_conv_out_t<f16>(x, B, H, W, C, gpart, chunks, gamma, beta, eps, w, ldw, bias, Cout, y, s); }
extern "C" int lr_gn_conv_out_bf16(const lr_half* x, int B, int H, int W, int C, const float* gpart, int chunks, const float* gamma, const float* beta, float eps, const lr_half* w, int ldw, const float* bias, int Cout, lr_half* y, lr_stream_t s) { return lr_gn_conv_out_t<bf16>(x, B, H, W, C, gpart, chunks, gamma, beta, eps, w, ldw, bias, Cout, y, s); }
