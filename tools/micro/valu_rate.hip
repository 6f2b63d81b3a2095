// Issue cost of VALU instruction kinds on gfx950: one wave per SIMD (4 waves per block, one block per CU), long dependent-free
// streams; prints shader cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 64
template <int KIND>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p[4] = {{a[0], a[1]}, {a[2], a[3]}, {a[4], a[5]}, {a[6], a[7]}};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (KIND == 0) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
        if (KIND == 1) a[i] = __builtin_amdgcn_exp2f(a[i]);
        if (KIND == 2) { p[i & 3] = __builtin_elementwise_fma(p[i & 3], (f2){1.0001f, 1.0001f}, (f2){0.5f, 0.5f}); }
        if (KIND == 3) a[i] = __builtin_amdgcn_rcpf(a[i]);
        if (KIND == 4) a[i] = fmaxf(fmaxf(a[i], a[(i + 1) & 7]), a[(i + 2) & 7]);
        if (KIND == 5) { _Float16 h = (_Float16)a[i]; asm volatile("v_exp_f16 %0, %0" : "+v"(h)); a[i] = (float)h; }
        if (KIND == 6) { p[i & 3] += (f2){0.5f, 0.25f}; }
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  for (int i = 0; i < 4; ++i) s += p[i][0] + p[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[KIND] = t1 - t0;
}
int main() {
  float* o; unsigned long long* c; unsigned long long h[8];
  hipMalloc(&o, 256 * 1024 * 4); hipMalloc(&c, 64);
  const int iters = 2000;
  const char* names[] = {"v_fma_f32", "v_exp_f32", "v_pk_fma_f32", "v_rcp_f32", "v_max3_f32", "cvt+v_exp_f16+cvt", "v_pk_add_f32"};
  for (int wpb = 1; wpb <= 2; ++wpb) {
    hipLaunchKernelGGL(k<0>, dim3(256), dim3(256 * wpb), 0, 0, o, c, iters);
    hipLaunchKernelGGL(k<1>, dim3(256), dim3(256 * wpb), 0, 0, o, c, iters);
    hipLaunchKernelGGL(k<2>, dim3(256), dim3(256 * wpb), 0, 0, o, c, iters);
    hipLaunchKernelGGL(k<3>, dim3(256), dim3(256 * wpb), 0, 0, o, c, iters);
    hipLaunchKernelGGL(k<4>, dim3(256), dim3(256 * wpb), 0, 0, o, c, iters);
    hipLaunchKernelGGL(k<5>, dim3(256), dim3(256 * wpb), 0, 0, o, c, iters);
    hipLaunchKernelGGL(k<6>, dim3(256), dim3(256 * wpb), 0, 0, o, c, iters);
    hipMemcpy(h, c, 64, hipMemcpyDeviceToHost);
    for (int i = 0; i < 7; ++i) printf("%d wave(s)/SIMD  %-20s %.2f cycles per wave-instruction\n", wpb, names[i], (double)h[i] / (iters * REP));
  }
  return 0;
}
