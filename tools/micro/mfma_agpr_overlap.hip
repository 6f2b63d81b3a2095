// Does the register file of the MFMA operands decide how many VALU instructions hide under a v_mfma_f32_32x32x16_f16?
// FORM 0: builtin (everything in arch VGPRs)   FORM 1: accumulator in AGPRs (inline asm, "+a"), A/B in VGPRs
// FORM 2: accumulator and A/B in AGPRs.  One or two waves per SIMD; wall time per MFMA step.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int FORM, int KIND, int NF>
__global__ void k(float* out, int iters) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  if (FORM >= 2) { asm volatile("" : "+a"(a)); if (FORM != 4) asm volatile("" : "+a"(b)); }   // park the operands in AGPRs
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (FORM == 0) acc[r & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[r & 3], 0, 0, 0);
      if (FORM == 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[r & 3]) : "v"(a), "v"(b));
      if (FORM == 2) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[r & 3]) : "a"(a), "a"(b));
      if (FORM == 3) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[r & 3]) : "a"(a), "a"(b));
      if (FORM == 4) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[r & 3]) : "a"(a), "v"(b));
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int i = (r * NF + f) & 7;
        if (KIND == 0) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
        if (KIND == 1) v[i] = __builtin_amdgcn_exp2f(v[i]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int FORM, int KIND, int NF>
float run(float* o, int wps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<FORM, KIND, NF>), dim3(256), dim3(256 * wps), 0, 0, o, 500);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<FORM, KIND, NF>), dim3(256), dim3(256 * wps), 0, 0, o, 500);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6f / (500.0f * 16);
}
template <int FORM>
void all(float* o) {
  const char* fn[] = {"VGPR (builtin)", "acc AGPR", "acc+A+B AGPR", "A+B AGPR accV", "acc+A AGPR  BV"};
  for (int wps = 1; wps <= 2; ++wps) {
    printf("%-14s %d wave/SIMD  ns per MFMA step:  +0: %5.1f | fma +2 %5.1f +4 %5.1f +6 %5.1f +8 %5.1f | exp +2 %5.1f +4 %5.1f +6 %5.1f\n", fn[FORM], wps,
           run<FORM, 0, 0>(o, wps), run<FORM, 0, 2>(o, wps), run<FORM, 0, 4>(o, wps), run<FORM, 0, 6>(o, wps), run<FORM, 0, 8>(o, wps),
           run<FORM, 1, 2>(o, wps), run<FORM, 1, 4>(o, wps), run<FORM, 1, 6>(o, wps));
  }
}
int main() {
  float* o;
  hipMalloc(&o, 256 * 1024 * 4);
  all<0>(o); all<1>(o); all<2>(o); all<3>(o); all<4>(o);
  return 0;
}
