// Helpers shared by the register-chained fused blocks (xattn_block.hip, ffn_block.hip): each wave owns 16 rows and hands the
// accumulators of one 16x16x32 MFMA product on as the B operand of the next.
#pragma once
#include "common.h"

typedef __attribute__((address_space(3))) void* lptr_t;

template <int N> __device__ __forceinline__ void xa_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// reductions over the four lanes (fr, fq = 0..3) that share a row
__device__ __forceinline__ float xa_row4_sum(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = __builtin_bit_cast(float, (unsigned)a[0]) + __builtin_bit_cast(float, (unsigned)a[1]);
  const unsigned u2 = __builtin_bit_cast(unsigned, v);
  const auto b = __builtin_amdgcn_permlane32_swap(u2, u2, false, false);
  v = __builtin_bit_cast(float, (unsigned)b[0]) + __builtin_bit_cast(float, (unsigned)b[1]);
#endif
  return v;
}
__device__ __forceinline__ float xa_row4_max(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = fmaxf(__builtin_bit_cast(float, (unsigned)a[0]), __builtin_bit_cast(float, (unsigned)a[1]));
  const unsigned u2 = __builtin_bit_cast(unsigned, v);
  const auto b = __builtin_amdgcn_permlane32_swap(u2, u2, false, false);
  v = fmaxf(__builtin_bit_cast(float, (unsigned)b[0]), __builtin_bit_cast(float, (unsigned)b[1]));
#endif
  return v;
}

template <typename T>
__device__ __forceinline__ vec8<T> xa_pack(const f32x4& a, const f32x4& b) {
  vec8<T> r;
#pragma unroll
  for (int i = 0; i < 4; ++i) { r[i] = (T)a[i]; r[4 + i] = (T)b[i]; }
  return r;
}
// xattn_block640.hip: the C = 640 / 10-head instance of lr_xattn_block_f16 (64-row blocks, 4 waves); arguments already checked
int lr_xattn640_launch(const lr_xattn_args* a, int bf16_, lr_stream_t s);

// Two swapped-form accumulator tiles (A = columns 16 j .., B = columns 16 (j + 1) .. of the same 16 rows; lane (fr, fq) holds columns
// 4 fq .. + 3 of row fr in each) -> after the exchange the lane owns EIGHT consecutive columns of row fr: columns
// 16 (j + (fq & 1)) + 8 (fq >> 1) .. + 7, in (a[0..3], b[0..3]) (the register epilogue of gemm_common.h: one 16-byte store per lane).
__device__ __forceinline__ void xa_swap_rows16(f32x4& a, f32x4& b) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float ar = a[r], br = b[r];
    const auto s = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, ar), __builtin_bit_cast(unsigned, br), false, false);
    a[r] = __builtin_bit_cast(float, (unsigned)s[0]);
    b[r] = __builtin_bit_cast(float, (unsigned)s[1]);
  }
#endif
}

// GroupNorm(32, affine) of a row-resident kernel's INPUT rows: per-channel scale / shift of sample n from the producer's per-group (sum,
// sumsq) partials gp [samples][chunks][32][2] -> LDS tables tabA / tabB [C].  Arithmetic of gn_apply_kernel (norm.hip), operation for
// operation -- chunk sums in fp64 by four lanes per group in the same order, mean / rstd rounded to fp32, a = rstd * gamma, b = beta - mean * a
// -- so `(T)fmaf(x, a, b)` on the rows equals what lr_groupnorm_apply_n would have written.  Block-wide (>= 128 threads); two barriers.
__device__ __forceinline__ void xa_gn_tables(const float* gp, int nchunks, int n, int HW, int C, const float* gamma, const float* beta,
                                             float eps, float* tabA, float* tabB, float* s_mr /* [64] */, int t, int nthreads) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int Cg = C / 32;
  if (t < 128) {
    const int g = t >> 2, sub = t & 3;
    double s = 0.0, q = 0.0;
    const float* ps = gp + ((size_t)n * nchunks * 32 + g) * 2;
#pragma unroll 4
    for (int c = sub; c < nchunks; c += 4) { s += (double)ps[c * 64]; q += (double)ps[c * 64 + 1]; }
#pragma unroll
    for (int sh = 2; sh > 0; sh >>= 1) { s += __shfl_xor(s, sh, 64); q += __shfl_xor(q, sh, 64); }
    if (sub == 0) {
      const double cnt = (double)HW * (double)Cg;
      const double mean = s / cnt;
      double var = q / cnt - mean * mean;
      if (var < 0.0) var = 0.0;
      s_mr[g] = (float)mean;
      s_mr[32 + g] = (float)(1.0 / sqrt(var + (double)eps));
    }
  }
  __syncthreads();
  for (int c = t; c < C; c += nthreads) {
    const int g = c / Cg;
    const float a = s_mr[32 + g] * gamma[c];
    tabA[c] = a;
    tabB[c] = beta[c] - s_mr[g] * a;
  }
  __syncthreads();
#endif
}

// stin_block.hip: entry of a SpatialTransformer at C = 320 (proj_in + LayerNorm + fused q|k|v projection in one launch)
