// How many independent VALU instructions hide under one v_mfma_f32_32x32x16_f16 (gfx950)?  Stream: [MFMA, NF fillers] x REP
// with independent registers; 1 or 2 waves per SIMD; accumulators in VGPRs or AGPRs (compile twice: with and without
// -mllvm -amdgpu-mfma-vgpr-form).  Prints shader cycles per MFMA step.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
template <int KIND, int NF>
__global__ void k(float* out, unsigned long long* cyc, int iters, int slot) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  f2 p[4] = {{v[0], v[1]}, {v[2], v[3]}, {v[4], v[5]}, {v[6], v[7]}};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc[r & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[r & 3], 0, 0, 0);
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int i = (r * NF + f) & 7;
        if (KIND == 0) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
        if (KIND == 1) v[i] = __builtin_amdgcn_exp2f(v[i]);
        if (KIND == 2) p[i & 3] = __builtin_elementwise_fma(p[i & 3], (f2){1.0001f, 1.0001f}, (f2){0.5f, 0.5f});
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int i = 0; i < 4; ++i) s += p[i][0] + p[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[slot] = t1 - t0;
}
static float g_us[64];
template <int KIND, int NF>
void run(float* o, unsigned long long* c, int wps, int slot) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, NF>), dim3(256), dim3(256 * wps), 0, 0, o, c, 500, slot);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<KIND, NF>), dim3(256), dim3(256 * wps), 0, 0, o, c, 500, slot);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  g_us[slot] = ms * 1e3f;
}
int main() {
  float* o; unsigned long long* c; unsigned long long h[64];
  hipMalloc(&o, 256 * 1024 * 4); hipMalloc(&c, 512);
  const char* kn[] = {"v_fma_f32", "v_exp_f32", "v_pk_fma_f32"};
  for (int wps = 1; wps <= 2; ++wps) {
    int s = 0;
    run<0, 0>(o, c, wps, s++); run<0, 2>(o, c, wps, s++); run<0, 4>(o, c, wps, s++); run<0, 6>(o, c, wps, s++); run<0, 8>(o, c, wps, s++); run<0, 12>(o, c, wps, s++);
    run<1, 2>(o, c, wps, s++); run<1, 4>(o, c, wps, s++); run<1, 6>(o, c, wps, s++);
    run<2, 4>(o, c, wps, s++); run<2, 8>(o, c, wps, s++);
    hipMemcpy(h, c, 512, hipMemcpyDeviceToHost);
    const int nf[] = {0, 2, 4, 6, 8, 12, 2, 4, 6, 4, 8};
    const int kd[] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 2, 2};
    for (int i = 0; i < s; ++i)
      printf("%d wave(s)/SIMD  MFMA + %2d x %-13s : %6.1f cycles per step   wall %7.1f us = %6.1f ns per step (%.0f TF MFMA)\n", wps, nf[i],
             kn[kd[i]], (double)h[i] / (500.0 * 16), g_us[i], g_us[i] * 1e3 / (500.0 * 16),
             256.0 * 4 * wps * 500 * 16 * 32768.0 / (g_us[i] * 1e-6) / 1e12);
  }
  return 0;
}
