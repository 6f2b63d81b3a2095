// Backward of the fused attention (d_head = 64): dQ, dK, dV from dO, recomputing P from the saved log-sum-exp
// (flash-attention style, no N x N matrix in memory).  Deterministic: two kernels, no atomics.
//
//   P  = exp2(c * q k^T - lse)          c = scale * log2(e), lse from lr_attention_lse_f16
//   D  = rowsum(dO o O)                 (prologue of attn_bwd_dq_kernel, handed to attn_bwd_dkv_kernel through `dsum`)
//   dV = P^T dO        dP = dO V^T      dS = P o (dP - D)
//   dQ = scale * dS K  dK = scale * dS^T Q
//
// Both kernels keep the forward's register conventions (attention.hip): 32x32x16 MFMAs, one column (query or key) per
// lane, the probability / dS tile feeds the next MFMA straight from the accumulator registers.  The operand that has to be
// read "k-major" (K^T for dQ; Q^T and dO^T for dK / dV) is the SAME natural [row][d] tile staged a second time by DMA with the
// forward's transpose-read swizzle and gathered by ds_read_b64_tr_b16 (attention.hip, VM = 2): no pre-transposed copies exist
// (round 6; rounds 2-5 made them with three lr_transpose_v_f16 launches per attention: 93 launches / 0.63 ms per training step).
//
//   attn_bwd_dq_kernel : block = 4 waves x 32 queries, loops over 64-key tiles.      12 of the 28 MFMA groups
//   attn_bwd_dkv_kernel: block = 4 waves x 32 keys,    loops over 64-query tiles.    16 of the 28 MFMA groups
#include "common.h"

#include <type_traits>

#define AB_THREADS 256
#define AB_TILE 64

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <typename T>
struct AttnBwdParams {
  const T* q; const T* k; const T* v; const T* o; const T* dout;
  const float* lse; float* dsum;                   // [B][heads][Nq]
  T* dq; T* dk; T* dv;
  int ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv, heads, Nq, Nkv, nblocks, ntile_blocks;
  int q_splits; float* kv_ws;      // dK / dV kernel: query tiles split over q_splits blocks per key block, fp32 partials [split][bh][kblk][dk | dv][128][64]
  float c, scale;
};

// 64 rows x 128 B tile -> LDS by DMA with the forward's swizzle; rows >= nrows read the zero page
template <typename T>
__device__ __forceinline__ void ab_stage_rows(char* dst, const T* base, int ld, int row0, int nrows, int w, int lane,
                                              const T* zero) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rbase = (i * 4 + w) * 8;
    const int row = rbase + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    const T* g = row0 + row < nrows ? base + (size_t)(row0 + row) * ld + chunk * 8 : zero;
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(dst + rbase * 128), 16, 0, 0);
  }
}
// the same 64 rows x 128 B tile for the transpose read: 16-byte chunk c of row r sits in slot c ^ 4 ((r >> 1) & 1), so the four
// rows of one ds_read_b64_tr_b16 (128 bytes apart) cover 64 distinct banks per half-wave (attention.hip, stage_vn)
template <typename T>
__device__ __forceinline__ void ab_stage_rows_tr(char* dst, const T* base, int ld, int row0, int nrows, int w, int lane,
                                                 const T* zero) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rbase = (i * 4 + w) * 8;
    const int row = rbase + (lane >> 3);
    const int chunk = (lane & 7) ^ (((row >> 1) & 1) << 2);
    const T* g = row0 + row < nrows ? base + (size_t)(row0 + row) * ld + chunk * 8 : zero;
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(dst + rbase * 128), 16, 0, 0);
  }
}
// A operand X^T[d = db*32 + ql][k-slots] of MFMA (rb, tt) out of the natural tile X[row][d]: k-slot (hi*8 + jj) is row
// rb*32 + 16*tt + 4*hi + jj (jj < 4) and rb*32 + 16*tt + 8 + 4*hi + (jj - 4) (jj >= 4) -- exactly the rows one accumulator-fed
// B operand (P or dS) carries.  Two transpose reads: a 16-lane group reads a [4 rows][16 d] block, every lane receives ONE d column.
struct AbTr {
  int off, chunk, swz;
  __device__ __forceinline__ AbTr(int lane) {
    const int hi = lane >> 5, j = (lane & 15) >> 2, q = lane & 3;
    off = (4 * hi + j) * 128 + (q & 1) * 8;
    chunk = 2 * ((lane >> 4) & 1) + (q >> 1);
    swz = (j >> 1) << 2;
  }
};
template <typename T>
__device__ __forceinline__ vec8<T> ab_frag_tr(const char* tile, const AbTr& tr, int rb, int tt, int db) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;
  const char* a = tile + (rb * 32 + 16 * tt) * 128 + tr.off + (((4 * db + tr.chunk) ^ tr.swz) << 4);
  const vec4<T> va = __builtin_bit_cast(vec4<T>, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)a));
  const vec4<T> vb = __builtin_bit_cast(vec4<T>, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(a + 8 * 128)));
  vec8<T> f;
  f[0] = va[0]; f[1] = va[1]; f[2] = va[2]; f[3] = va[3];
  f[4] = vb[0]; f[5] = vb[1]; f[6] = vb[2]; f[7] = vb[3];
  return f;
}
template <typename T>
__device__ __forceinline__ vec8<T> ab_frag(const char* tile, int row, int chunk) {
  return *reinterpret_cast<const vec8<T>*>(tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}

// ---------------------------------------------------------------------------------------------------------------------
// dQ: one query per lane (ql), the wave's 32 queries against every key tile.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(AB_THREADS) void attn_bwd_dq_kernel(const AttnBwdParams<T> P) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 3 * AB_TILE * 128];   // {K, V, K in the transpose-read swizzle} x 2 buffers
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int qblk = blockIdx.x % P.ntile_blocks;
  const int bh = blockIdx.x / P.ntile_blocks;
  const int h = bh % P.heads, b = bh / P.heads;
  const int ql = lane & 31, hi = lane >> 5;
  const T* zero = reinterpret_cast<const T*>(lr_zero_page);

  const T* kp = P.k + (size_t)b * P.Nkv * P.ldk + h * 64;
  const T* vp = P.v + (size_t)b * P.Nkv * P.ldv + h * 64;
  const AbTr tr(lane);
  const int qrow = qblk * 128 + w * 32 + ql;
  const int qc = min(qrow, P.Nq - 1);
  vec8<T> qf[4], gf[4];      // Q^T / dO^T B-operands: lane holds row qc, columns s4*16 + hi*8 .. +8
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    qf[s4] = *reinterpret_cast<const vec8<T>*>(P.q + ((size_t)b * P.Nq + qc) * P.ldq + h * 64 + s4 * 16 + hi * 8);
    gf[s4] = *reinterpret_cast<const vec8<T>*>(P.dout + ((size_t)b * P.Nq + qc) * P.lddo + h * 64 + s4 * 16 + hi * 8);
  }
  const size_t sidx = ((size_t)b * P.heads + h) * P.Nq + qc;
  const float lse = P.lse[sidx];
  // D[q] = sum_d dO[q][d] O[q][d]: the lane already holds half of the query's dO row (columns s4*16 + hi*8 .. +8), the other half sits
  // in lane ^ 32 -- computed here and written for the dK / dV kernel that follows on the stream (round 6: attn_bwd_prep_kernel, one
  // uncoalesced launch per attention, is gone)
  float dsum = 0.f;
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    const vec8<T> of = *reinterpret_cast<const vec8<T>*>(P.o + ((size_t)b * P.Nq + qc) * P.ldo + h * 64 + s4 * 16 + hi * 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) dsum = fmaf((float)gf[s4][i], (float)of[i], dsum);
  }
  dsum += __shfl_xor(dsum, 32, 64);
  if (hi == 0 && qrow < P.Nq) P.dsum[sidx] = dsum;

  f32x16 dq[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[d][r] = 0.f;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  const int ntiles = (P.Nkv + AB_TILE - 1) / AB_TILE;
  auto stage = [&](int buf, int tile) {
    char* base = smem + buf * (3 * AB_TILE * 128);
    ab_stage_rows(base, kp, P.ldk, tile * AB_TILE, P.Nkv, w, lane, zero);
    ab_stage_rows(base + AB_TILE * 128, vp, P.ldv, tile * AB_TILE, P.Nkv, w, lane, zero);
    ab_stage_rows_tr(base + 2 * AB_TILE * 128, kp, P.ldk, tile * AB_TILE, P.Nkv, w, lane, zero);
  };
  stage(0, 0);
  __syncthreads();

  auto process = [&](int tile, auto tail_tag) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    const int cur = tile & 1;
    if (tile + 1 < ntiles) stage(cur ^ 1, tile + 1);
    const char* Ks = smem + cur * (3 * AB_TILE * 128);
    const char* Vs = Ks + AB_TILE * 128;
    const char* Kt = Ks + 2 * AB_TILE * 128;
    f32x16 sacc[2], pacc[2];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int row = kb * 32 + ql;
        sacc[kb] = lr_mfma32(ab_frag<T>(Ks, row, s4 * 2 + hi), qf[s4], s4 == 0 ? zero16 : sacc[kb]);
        pacc[kb] = lr_mfma32(ab_frag<T>(Vs, row, s4 * 2 + hi), gf[s4], s4 == 0 ? zero16 : pacc[kb]);
      }
    vec8<T> dsf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float p = __builtin_amdgcn_exp2f(fmaf(sacc[kb][r], P.c, -lse));
        if constexpr (TAIL) {
          const int key = tile * AB_TILE + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key >= P.Nkv) p = 0.f;
        }
        dsf[kb][r >> 3][r & 7] = (T)(p * (pacc[kb][r] - dsum));
      }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int db = 0; db < 2; ++db)
          dq[db] = lr_mfma32(ab_frag_tr<T>(Kt, tr, kb, tt, db), dsf[kb][tt], dq[db]);
    __syncthreads();
  };
  const int nfull = P.Nkv / AB_TILE;
  for (int tile = 0; tile < nfull; ++tile) process(tile, std::false_type{});
  if (nfull < ntiles) process(nfull, std::true_type{});

  if (qrow < P.Nq) {
    T* dst = P.dq + ((size_t)b * P.Nq + qrow) * P.lddq + h * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        vec4<T> ov;
#pragma unroll
        for (int i = 0; i < 4; ++i) ov[i] = (T)(dq[db][g * 4 + i] * P.scale);
        *reinterpret_cast<vec4<T>*>(dst + db * 32 + 8 * g + 4 * hi) = ov;
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// dK, dV: one key per lane (ql), the wave's 32 keys against every query tile.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(AB_THREADS) void attn_bwd_dkv_kernel(const AttnBwdParams<T> P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // {Q, dO, and both again in the transpose-read swizzle} x 2 buffers (64 KB) + lse, D (1 KB)
  float (*s_lse)[AB_TILE] = reinterpret_cast<float (*)[AB_TILE]>(smem + 2 * 4 * AB_TILE * 128);
  float (*s_dsum)[AB_TILE] = s_lse + 2;
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int kblk = blockIdx.x % P.ntile_blocks;
  const int rest = blockIdx.x / P.ntile_blocks;
  const int sp = rest % P.q_splits;      // this block's slice of the query tiles (q_splits = 1: all of them)
  const int bh = rest / P.q_splits;
  const int h = bh % P.heads, b = bh / P.heads;
  const int ql = lane & 31, hi = lane >> 5;
  const T* zero = reinterpret_cast<const T*>(lr_zero_page);

  const T* qp = P.q + (size_t)b * P.Nq * P.ldq + h * 64;
  const T* gp = P.dout + (size_t)b * P.Nq * P.lddo + h * 64;
  const AbTr tr(lane);
  const float* lsep = P.lse + ((size_t)b * P.heads + h) * P.Nq;
  const float* dsp = P.dsum + ((size_t)b * P.heads + h) * P.Nq;
  const int krow = kblk * 128 + w * 32 + ql;
  const int kc = min(krow, P.Nkv - 1);
  vec8<T> kf[4], vf[4];      // K^T / V^T B-operands: lane holds row kc, columns s4*16 + hi*8 .. +8
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    kf[s4] = *reinterpret_cast<const vec8<T>*>(P.k + ((size_t)b * P.Nkv + kc) * P.ldk + h * 64 + s4 * 16 + hi * 8);
    vf[s4] = *reinterpret_cast<const vec8<T>*>(P.v + ((size_t)b * P.Nkv + kc) * P.ldv + h * 64 + s4 * 16 + hi * 8);
  }
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[d][r] = 0.f; dv[d][r] = 0.f; }
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  const int ntiles_all = (P.Nq + AB_TILE - 1) / AB_TILE;
  const int t_per = (ntiles_all + P.q_splits - 1) / P.q_splits;
  const int t0 = sp * t_per, ntiles = min(ntiles_all, t0 + t_per);      // this block runs query tiles [t0, ntiles)
  auto stage = [&](int buf, int tile) {
    char* base = smem + buf * (4 * AB_TILE * 128);
    ab_stage_rows(base, qp, P.ldq, tile * AB_TILE, P.Nq, w, lane, zero);
    ab_stage_rows(base + AB_TILE * 128, gp, P.lddo, tile * AB_TILE, P.Nq, w, lane, zero);
    ab_stage_rows_tr(base + 2 * AB_TILE * 128, qp, P.ldq, tile * AB_TILE, P.Nq, w, lane, zero);
    ab_stage_rows_tr(base + 3 * AB_TILE * 128, gp, P.lddo, tile * AB_TILE, P.Nq, w, lane, zero);
    if (t < 2 * AB_TILE) {       // rows past Nq: lse = +inf -> P = 0, D = 0
      const int i = t & (AB_TILE - 1), qi = tile * AB_TILE + i;
      if (t < AB_TILE) s_lse[buf][i] = qi < P.Nq ? lsep[qi] : INFINITY;
      else s_dsum[buf][i] = qi < P.Nq ? dsp[qi] : 0.f;
    }
  };
  if (t0 < ntiles) stage(0, t0);
  __syncthreads();

  for (int tile = t0; tile < ntiles; ++tile) {
    const int cur = (tile - t0) & 1;
    if (tile + 1 < ntiles) stage(cur ^ 1, tile + 1);
    const char* Qs = smem + cur * (4 * AB_TILE * 128);
    const char* Gs = Qs + AB_TILE * 128;
    const char* Qt = Qs + 2 * AB_TILE * 128;
    const char* Gt = Qs + 3 * AB_TILE * 128;
    // S[q][key] = Q K^T and dP[q][key] = dO V^T: rows = the tile's queries (2 blocks of 32), column = this lane's key
    f32x16 sacc[2], pacc[2];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        const int row = qb * 32 + ql;
        sacc[qb] = lr_mfma32(ab_frag<T>(Qs, row, s4 * 2 + hi), kf[s4], s4 == 0 ? zero16 : sacc[qb]);
        pacc[qb] = lr_mfma32(ab_frag<T>(Gs, row, s4 * 2 + hi), vf[s4], s4 == 0 ? zero16 : pacc[qb]);
      }
    // accumulator reg r of block qb is query qb*32 + (r&3) + 8*(r>>2) + 4*hi: its lse / D come from LDS, 4 at a time
    vec8<T> pf[2][2], dsf[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(&s_lse[cur][qb * 32 + 8 * g + 4 * hi]);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(&s_dsum[cur][qb * 32 + 8 * g + 4 * hi]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = g * 4 + i;
          const float p = __builtin_amdgcn_exp2f(fmaf(sacc[qb][r], P.c, -l4[i]));
          pf[qb][r >> 3][r & 7] = (T)p;
          dsf[qb][r >> 3][r & 7] = (T)(p * (pacc[qb][r] - d4[i]));
        }
      }
    // dV^T[d][key] += dO^T[d][q] P[q][key];  dK^T[d][key] += Q^T[d][q] dS[q][key]
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          dv[db] = lr_mfma32(ab_frag_tr<T>(Gt, tr, qb, tt, db), pf[qb][tt], dv[db]);
          dk[db] = lr_mfma32(ab_frag_tr<T>(Qt, tr, qb, tt, db), dsf[qb][tt], dk[db]);
        }
    __syncthreads();
  }

  if (P.q_splits > 1) {      // fp32 partials of this query slice (unscaled dK), summed in a fixed order by attn_bwd_kv_reduce_kernel
    const int nbh = gridDim.x / (P.q_splits * P.ntile_blocks);
    float* wp = P.kv_ws + ((((size_t)sp * nbh + bh) * P.ntile_blocks + kblk) * 2) * (128 * 64) + (size_t)(w * 32 + ql) * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 a = {dk[db][g * 4], dk[db][g * 4 + 1], dk[db][g * 4 + 2], dk[db][g * 4 + 3]};
        const f32x4 c2 = {dv[db][g * 4], dv[db][g * 4 + 1], dv[db][g * 4 + 2], dv[db][g * 4 + 3]};
        *reinterpret_cast<f32x4*>(wp + db * 32 + 8 * g + 4 * hi) = a;
        *reinterpret_cast<f32x4*>(wp + 128 * 64 + db * 32 + 8 * g + 4 * hi) = c2;
      }
    return;
  }
  if (krow < P.Nkv) {
    T* dkp = P.dk + ((size_t)b * P.Nkv + krow) * P.lddk + h * 64;
    T* dvp = P.dv + ((size_t)b * P.Nkv + krow) * P.lddv + h * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        vec4<T> a, c2;
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = (T)(dk[db][g * 4 + i] * P.scale); c2[i] = (T)dv[db][g * 4 + i]; }
        *reinterpret_cast<vec4<T>*>(dkp + db * 32 + 8 * g + 4 * hi) = a;
        *reinterpret_cast<vec4<T>*>(dvp + db * 32 + 8 * g + 4 * hi) = c2;
      }
  }
}

// sum of the q_splits partial dK / dV tiles of a (batch, head, key block), splits in order: one thread per 4 consecutive d of a key
template <typename T>
__global__ void attn_bwd_kv_reduce_kernel(const AttnBwdParams<T> P, int nbh) {
  const long long id = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const long long total = (long long)nbh * P.ntile_blocks * 128 * 16;
  if (id >= total) return;
  const int d4 = (int)(id & 15) * 4;
  const int key = (int)((id >> 4) & 127);
  const long long tile = id >> 11;      // bh * ntile_blocks + kblk
  const int kblk = (int)(tile % P.ntile_blocks);
  const int bh = (int)(tile / P.ntile_blocks);
  const int krow = kblk * 128 + key;
  if (krow >= P.Nkv) return;
  const size_t split_stride = (size_t)nbh * P.ntile_blocks * 2 * (128 * 64);
  const float* src = P.kv_ws + (size_t)tile * 2 * (128 * 64) + (size_t)key * 64 + d4;
  f32x4 a = {0.f, 0.f, 0.f, 0.f}, c2 = a;
  for (int sidx = 0; sidx < P.q_splits; ++sidx) {
    a += *reinterpret_cast<const f32x4*>(src + sidx * split_stride);
    c2 += *reinterpret_cast<const f32x4*>(src + sidx * split_stride + 128 * 64);
  }
  const int h = bh % P.heads, b = bh / P.heads;
  vec4<T> ka, va;
#pragma unroll
  for (int i = 0; i < 4; ++i) { ka[i] = (T)(a[i] * P.scale); va[i] = (T)c2[i]; }
  *reinterpret_cast<vec4<T>*>(P.dk + ((size_t)b * P.Nkv + krow) * P.lddk + h * 64 + d4) = ka;
  *reinterpret_cast<vec4<T>*>(P.dv + ((size_t)b * P.Nkv + krow) * P.lddv + h * 64 + d4) = va;
}

template <typename T>
static int lr_attention_bwd_t(const lr_attn_bwd_args* a, lr_stream_t s) {
  if (!a || !a->q || !a->k || !a->v || !a->o || !a->dout || !a->lse || !a->dsum || !a->dq || !a->dk || !a->dv)
    return LR_E_ARG;
  if (a->B <= 0 || a->heads <= 0 || a->Nq <= 0 || a->Nkv <= 0) return LR_E_ARG;
  if ((a->ldq | a->ldk | a->ldv | a->ldo | a->lddo) % 8 || (a->lddq | a->lddk | a->lddv) % 4) return LR_E_ALIGN;
  if (((uintptr_t)a->q | (uintptr_t)a->k | (uintptr_t)a->v | (uintptr_t)a->o | (uintptr_t)a->dout) & 15) return LR_E_ALIGN;
  AttnBwdParams<T> P;
  P.q = (const T*)a->q; P.k = (const T*)a->k; P.v = (const T*)a->v; P.o = (const T*)a->o;
  P.dout = (const T*)a->dout;
  P.lse = a->lse; P.dsum = a->dsum;
  P.dq = (T*)a->dq; P.dk = (T*)a->dk; P.dv = (T*)a->dv;
  P.ldq = a->ldq; P.ldk = a->ldk; P.ldv = a->ldv; P.ldo = a->ldo; P.lddo = a->lddo;
  P.lddq = a->lddq; P.lddk = a->lddk; P.lddv = a->lddv;
  P.heads = a->heads; P.Nq = a->Nq; P.Nkv = a->Nkv;
  P.scale = a->scale;
  P.c = a->scale * 1.44269504088896340736f;
  P.q_splits = 1;
  P.kv_ws = nullptr;
  hipStream_t st = (hipStream_t)s;
  int rc;
  P.ntile_blocks = (a->Nq + 127) / 128;
  hipLaunchKernelGGL(attn_bwd_dq_kernel<T>, dim3(P.ntile_blocks * a->heads * a->B), dim3(AB_THREADS), 0, st, P);
  rc = lr_launch_status();
  if (rc) return rc;
  P.ntile_blocks = (a->Nkv + 127) / 128;
  const int dkv_smem = 2 * 4 * AB_TILE * 128 + 4 * AB_TILE * (int)sizeof(float);
  static unsigned long long attr_done = 0;
  if (lr_attr_needed(&attr_done)) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, dkv_smem);
  }
  // ABI 26: `qt` / `ld_qt` of the argument struct carry the query split of this kernel (few key blocks against many queries -- the 77-key
  // cross-attention of the 64 x 128 level is 1 key block per (batch, head): 80 blocks for 256 CUs): ld_qt = number of query slices (0 / 1 =
  // none), qt = fp32 workspace of ld_qt * B * heads * ceil(Nkv / 128) * 2 * 128 * 64 floats
  P.q_splits = a->ld_qt > 1 && a->qt ? a->ld_qt : 1;
  P.kv_ws = P.q_splits > 1 ? (float*)const_cast<lr_half*>(a->qt) : nullptr;
  if (P.q_splits > 64 || (P.kv_ws && ((uintptr_t)P.kv_ws & 15))) return LR_E_ARG;
  hipLaunchKernelGGL(attn_bwd_dkv_kernel<T>, dim3(P.ntile_blocks * P.q_splits * a->heads * a->B), dim3(AB_THREADS), dkv_smem, st, P);
  rc = lr_launch_status();
  if (rc || P.q_splits == 1) return rc;
  const int nbh = a->heads * a->B;
  const long long total = (long long)nbh * P.ntile_blocks * 128 * 16;
  hipLaunchKernelGGL(attn_bwd_kv_reduce_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, P, nbh);
  return lr_launch_status();
}

// ---- C ABI: every entry point in its fp16 and bf16 form -------------------------------------------------------------
extern "C" int lr_attention_bwd_f16(const lr_attn_bwd_args* a, lr_stream_t s) { return lr_attention_bwd_t<f16>(a, s); }
extern "C" int lr_attention_bwd_bf16(const lr_attn_bwd_args* a, lr_stream_t s) { return lr_attention_bwd_t<bf16>(a, s); }
