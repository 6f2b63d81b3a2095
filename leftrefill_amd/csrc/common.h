// Shared device helpers for the gfx950 (CDNA4, wave64) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/leftrefill_hip.h"

#define LR_WAVE 64

// Developer knobs.  The PRODUCT library has none: LR_DEV(name, dflt) is the constant `dflt`, and nothing in csrc/ reads the environment,
// a file or any other process-global switch ("no global state inside").  A -DLR_DEV_VARIANTS build (tools/build_variant.sh)
// additionally compiles the measured-and-lost kernel variants (profiles/README.md lists them) and exports lr_dev_set(name, value): the
// Python front end forwards LR_* environment variables through it (leftrefill_amd/_lib.py), so the A/B scripts of tools/ keep working
// against a variant library.
#ifdef LR_DEV_VARIANTS
extern "C" int lr_dev_set(const char* name, int value);
int lr_dev_get(const char* name, int dflt);
#define LR_DEV(name, dflt) lr_dev_get(name, dflt)
#else
#define LR_DEV(name, dflt) (dflt)
#endif

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

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE property of a kernel: one bit per device ordinal records where
// it has been set (a process that launches on a second GPU would otherwise fail there: ADVICE r3).
static inline bool lr_attr_needed(unsigned long long* done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return true;
  const unsigned long long bit = 1ull << (dev & 63);
  if (*done & bit) return false;
  *done |= bit;
  return true;
}

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

// erf-GELU for the GEMM epilogues (two values per instruction stream, VALU-bound there): GELU(x) = x Phi(x) with
//   Phi(-a) = 0.5 erfc(a / sqrt 2) = exp2(P7(a)),  a = min(|x|, 6.081)      (Phi(-6.08) = 6e-10)
//   Phi(x)  = 0.5 + copysign(0.5 - Phi(-|x|), x)
// P7 = degree-7 fit of log2 Phi(-a) at Chebyshev nodes of [0, 6.081]: ONE transcendental (exp2) and 7 packed FMAs per
// value instead of rcp + exp2 + a degree-5 polynomial; |GELU error| <= 6.4e-7 absolute over all x in fp32 evaluation
// (checked on 1.8 M points, tools/fit_gelu.py) -- 3 orders below the fp16 / bf16 rounding of the stored result.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// Phi(x) - 0.5 for two values (the part of lr_gelu_erf2 the GEGLU backward shares: gelu'(x) = Phi(x) + x phi(x))
__device__ __forceinline__ f32x2_t lr_phi_mhalf2(const f32x2_t x) {
  const f32x2_t a = {fminf(fabsf(x[0]), 6.08111832f), fminf(fabsf(x[1]), 6.08111832f)};
  f32x2_t p = __builtin_elementwise_fma(a, (f32x2_t){-1.874598518e-06f, -1.874598518e-06f}, (f32x2_t){6.238633299e-05f, 6.238633299e-05f});
  p = __builtin_elementwise_fma(p, a, (f32x2_t){-9.366052953e-04f, -9.366052953e-04f});
  p = __builtin_elementwise_fma(p, a, (f32x2_t){8.530917476e-03f, 8.530917476e-03f});
  p = __builtin_elementwise_fma(p, a, (f32x2_t){-5.400426252e-02f, -5.400426252e-02f});
  p = __builtin_elementwise_fma(p, a, (f32x2_t){-4.584246621e-01f, -4.584246621e-01f});
  p = __builtin_elementwise_fma(p, a, (f32x2_t){-1.151264151e+00f, -1.151264151e+00f});
  p = __builtin_elementwise_fma(p, a, (f32x2_t){-9.999946099e-01f, -9.999946099e-01f});
  const f32x2_t e = {__builtin_amdgcn_exp2f(p[0]), __builtin_amdgcn_exp2f(p[1])};     // Phi(-|x|)
  const f32x2_t h = (f32x2_t){0.5f, 0.5f} - e;
  return (f32x2_t){copysignf(h[0], x[0]), copysignf(h[1], x[1])};
}
__device__ __forceinline__ f32x2_t lr_gelu_erf2(const f32x2_t x) {
  return __builtin_elementwise_fma(x, lr_phi_mhalf2(x), x * 0.5f);
}

// scalar form: the same formula, so the unfused training forward (lr_geglu_fwd) and the fused epilogue agree
__device__ __forceinline__ float lr_gelu_erf(float x) { return lr_gelu_erf2((f32x2_t){x, x})[0]; }

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
