// Shared device helpers for the gfx950 (CDNA4, wave64) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/leftrefill_hip.h"

#define LR_WAVE 64

typedef _Float16 f16;
typedef __bf16 bf16;          // the second 16-bit activation / weight type (LR_DTYPE_BF16): same layouts, same kernels
template <typename T> using vec2 = T __attribute__((ext_vector_type(2)));
template <typename T> using vec4 = T __attribute__((ext_vector_type(4)));
template <typename T> using vec8 = T __attribute__((ext_vector_type(8)));
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 256 B of zeros; target of every padded / out-of-range 16-byte (LDS-DMA) load.  One private copy per
// translation unit (no relocatable device code needed).
static __device__ uint4 lr_zero_page[16];

static inline int lr_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

__device__ __forceinline__ float lr_silu(float x) { return x / (1.0f + __expf(-x)); }

// erf-form GELU, as F.gelu default (attention.py:58).  erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below
// the fp16 output rounding) -- ~3x fewer VALU ops than libm erff, which matters in the GEGLU GEMM epilogue.
__device__ __forceinline__ float lr_erf(float x) {
  const float ax = fabsf(x);
  const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __expf(-ax * ax);
  const float r = fmaf(-p * t, e, 1.0f);
  return copysignf(r, x);
}
__device__ __forceinline__ float lr_gelu_erf(float x) { return 0.5f * x * (1.0f + lr_erf(x * 0.70710678118654752f)); }

// Two erf-GELUs per instruction stream: the polynomial / scaling steps as packed fp32 math (v_pk_fma_f32 / v_pk_mul_f32,
// two values per VALU issue), only rcp / exp2 stay scalar.  Same formula and constants as lr_gelu_erf (bit-identical
// results are not required between the two, both are within 1.5e-7 of erf).  The GEGLU epilogue is VALU-bound on this.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t lr_gelu_erf2(const f32x2_t x) {
  const f32x2_t z = x * 0.70710678118654752f;
  const f32x2_t az = {fabsf(z[0]), fabsf(z[1])};
  const f32x2_t one = {1.0f, 1.0f};
  const f32x2_t d = __builtin_elementwise_fma(az, (f32x2_t){0.3275911f, 0.3275911f}, one);
  const f32x2_t t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  f32x2_t p = __builtin_elementwise_fma(t, (f32x2_t){1.061405429f, 1.061405429f}, (f32x2_t){-1.453152027f, -1.453152027f});
  p = __builtin_elementwise_fma(p, t, (f32x2_t){1.421413741f, 1.421413741f});
  p = __builtin_elementwise_fma(p, t, (f32x2_t){-0.284496736f, -0.284496736f});
  p = __builtin_elementwise_fma(p, t, (f32x2_t){0.254829592f, 0.254829592f});
  const f32x2_t a2 = az * az * -1.44269504088896340736f;          // exp(-z^2) = exp2(-z^2 log2 e)
  const f32x2_t e = {__builtin_amdgcn_exp2f(a2[0]), __builtin_amdgcn_exp2f(a2[1])};
  const f32x2_t r = __builtin_elementwise_fma(-(p * t), e, one);   // erf(|z|)
  const f32x2_t er = {copysignf(r[0], z[0]), copysignf(r[1], z[1])};
  return (x * 0.5f) * (er + one);
}

__device__ __forceinline__ float lr_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float lr_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// 16-byte vector of 8 halves (fp16 or bf16) <-> floats
template <typename T = f16>
__device__ __forceinline__ void lr_unpack8(const uint4& u, float* f) {
  const vec8<T> h = __builtin_bit_cast(vec8<T>, u);
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = (float)h[i];
}
template <typename T = f16>
__device__ __forceinline__ uint4 lr_pack8(const float* f) {
  vec8<T> h;
#pragma unroll
  for (int i = 0; i < 8; ++i) h[i] = (T)f[i];
  return __builtin_bit_cast(uint4, h);
}

// matrix-core products on either 16-bit type (fp32 accumulate)
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ f32x4_t lr_mfma16(vec8<f16> a, vec8<f16> b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4_t lr_mfma16(vec8<bf16> a, vec8<bf16> b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16_t lr_mfma32(vec8<f16> a, vec8<f16> b, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16_t lr_mfma32(vec8<bf16> a, vec8<bf16> b, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
#else   // host pass of hipcc: kernel bodies are parsed but never run; the matrix-core builtins only exist for the device
template <typename V> __device__ f32x4_t lr_mfma16(V, V, f32x4_t c) { return c; }
template <typename V> __device__ f32x16_t lr_mfma32(V, V, f32x16_t c) { return c; }
#endif

// Buffer descriptor from provably wave-uniform pieces (readfirstlane), otherwise hipcc wraps every buffer op in a
// waterfall loop (guide T20).  The descriptor type only exists in the device pass of hipcc, hence the guard (the host
// pass still has to see the kernel declaration to emit its launch stub).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* p, size_t bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  const int n = __builtin_amdgcn_readfirstlane((int)bytes);
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, n, 0x00020000);
}
#endif
