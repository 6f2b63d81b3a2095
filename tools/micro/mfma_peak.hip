// Sustained dense-fp16 MFMA rate of the whole chip and the shader clock it runs at (MI355X, gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/micro/mfma_peak.hip && /tmp/mfma_peak
// Every wave runs a register-only loop of independent v_mfma chains (no LDS, no memory) for >= 20 ms per launch; the kernel
// reads the shader-clock counter (s_memtime) at both ends, so  clock = cycles / wall time  and  TFLOP/s = flops / wall time.
// Variants: 16x16x32 and 32x32x16 f16, 1 or 2 waves per SIMD, operands random in [-1, 1) or all zero (the chip clocks to its
// power budget: zero operands toggle nothing and run faster -- the random-data number is the one a real GEMM can be held against).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>   // 16 or 32
__global__ __launch_bounds__(512) void peak_kernel(const f16x8* __restrict__ ab, float* __restrict__ out,
                                                   unsigned long long* __restrict__ cyc, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  f16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = ab[(t * 8 + i) & 65535]; b[i] = ab[(t * 8 + 4 + i) & 65535]; }
  const unsigned long long t0 = __builtin_readcyclecounter();
  float s = 0.f;
  if constexpr (SHAPE == 16) {
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 32; ++r)
        acc[r & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[r & 3], b[(r >> 2) & 3], acc[r & 7], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  } else {
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        acc[r & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[r & 3], b[(r >> 2) & 3], acc[r & 3], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) s += acc[j][i];
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[t] = s;
  if ((threadIdx.x & 63) == 0) cyc[t >> 6] = t1 - t0;
}

int main() {
  const int blocks = 256;
  f16x8* ab; float* out; unsigned long long* cyc;
  hipMalloc(&ab, 65536 * sizeof(f16x8)); hipMalloc(&out, blocks * 512 * 4); hipMalloc(&cyc, blocks * 8 * 8);
  std::vector<_Float16> h(65536 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int zero = 0; zero < 2; ++zero) {
    srand(1);
    for (auto& x : h) x = zero ? (_Float16)0.f : (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f);
    hipMemcpy(ab, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int shape : {16, 32}) {
      for (int wps = 1; wps <= 2; ++wps) {
        const int threads = 256 * wps;
        // flops per wave per iteration: 32 x (16*16*32*2) = 16 x (32*32*16*2) = 524288
        const int iters = 80000 / wps;
        auto launch = [&](int n) {
          if (shape == 16) hipLaunchKernelGGL(peak_kernel<16>, dim3(blocks), dim3(threads), 0, 0, ab, out, cyc, n);
          else hipLaunchKernelGGL(peak_kernel<32>, dim3(blocks), dim3(threads), 0, 0, ab, out, cyc, n);
        };
        launch(iters / 8);
        hipDeviceSynchronize();
        hipEventRecord(e0); launch(iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> c(blocks * 4 * wps);
        hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
        double mean = 0; for (auto v : c) mean += (double)v; mean /= c.size();
        const double flops = (double)blocks * 4 * wps * iters * 524288.0;
        printf("%s operands  v_mfma_f32_%s_f16  %d wave(s)/SIMD: %7.2f ms  %7.1f TFLOP/s  shader clock %5.0f MHz  (%.1f cycles per MFMA per SIMD)\n",
               zero ? "zero  " : "random", shape == 16 ? "16x16x32" : "32x32x16", wps, ms, flops / (ms * 1e-3) / 1e12,
               mean / (ms * 1e-3) / 1e6, mean / ((double)iters * (shape == 16 ? 32 : 16) * wps));
      }
    }
  }
  return 0;
}
