// Fused scaled-dot-product attention forward (flash-style online softmax), d_head = 64, fp16 in / fp32 accumulate.
//
// Work split: block = 4 waves = 256 queries of one (batch, head); each wave owns 64 queries (two 32-query blocks that
// share every K / V fragment read from LDS) and streams the K/V sequence in 64-key tiles.  Per tile and wave:
// S^T = K Q^T (16 x mfma 32x32x16) -> online softmax in registers -> O^T += V^T P^T (16 x mfma 32x32x16).
//
// * "Swapped" products: computing S^T / O^T puts ONE query per lane (column = lane & 31), so the row-max / row-sum
//   are in-lane reductions plus a single lane <-> lane+32 exchange, and the P^T B-operand of the second product is
//   exactly the accumulator registers of the first (the MFMA sums over k, so the k order of P^T and V^T only has to
//   agree: V^T fragments are read in the accumulator's key order).
// * K tile in LDS: [64 keys][64 d] fp16, 128-byte rows, filled by 16-byte LDS-DMA with the same source-side XOR
//   swizzle as the GEMM (slot = chunk ^ ((row >> 1) & 7)) => conflict-free ds_read_b128 fragment reads.
// * V tile is transposed on the way in: [64 d][64 keys] with a 136-byte pitch, written as packed key pairs
//   (ds_write_b32), read as two ds_read_b64 per fragment (conflict-free at this pitch).
// * Q fragments live in registers for the whole kernel; K/V double-buffered, one barrier per tile.
// * The two query blocks of a wave are independent instruction streams: the QK^T / PV MFMAs of one overlap the
//   exp2 / max / convert VALU work of the other.  The running-max rescale of O is deferred until a row max grows by
//   more than 2^8 (exp2 domain).
#include "common.h"

#include <stdlib.h>
#include <type_traits>

#define ATT_THREADS 256
#define ATT_QB 256
#define ATT_KB 64
#define VT_PITCH 136  // bytes per V^T row (64 keys * 2 B + 8 pad)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <typename T>
struct AttnParams {
  const T* q; const T* k; const T* v; T* o;
  int ldq, ldk, ldv, ldo, heads, Nq, Nkv, nqt, nblocks;
  float c;  // scale * log2(e)
  float* lse;  // optional [B][heads][Nq]: log2-domain log-sum-exp m*c + log2(l) (saved for the backward), or NULL
};

// Online softmax of one 32-query block of a wave over one 64-key tile: S^T accumulators -> P^T (16-bit, the PV B operand), running
// max / sum, deferred rescale of O (only when some row's max grew by more than 2^8: P stays <= 256).  One query per lane; the
// partner lane ^ 32 holds the other half of the keys.
// FOLD = false: sacc holds raw q.k, m_run the raw running max (starts at -inf), p = exp2(c (s - m)).
// FOLD = true : Q was pre-multiplied by c = scale log2(e) and the accumulators were INITIALISED with -m_run (`minit`, the C operand of
//               the first QK^T MFMA), so sacc already is the exp2 argument: no per-element multiply-subtract (64 of the ~185
//               non-transcendental VALU instructions of a tile; the kernel is VALU-bound at d_head = 64).  m_run starts at 0 and the
//               first tile always takes the rescale path (sets m_run to the tile's row max).  Q's extra fp16 rounding moves a logit
//               by <= 2^-12 relative per term -- far inside the fp16 rounding of P; the training forward (log-sum-exp output for the
//               backward, which recomputes P from unscaled q, k) keeps FOLD = false.
template <typename T, bool FOLD>
__device__ __forceinline__ void softmax_block(f32x16 (&sacc)[2], f32x16 (&oacc)[2], vec8<T> (&pf)[2][2], float& m_run, float& l_run,
                                              f32x16& minit, const float c, const bool first) {
  float mx = sacc[0][0];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[kb][r]);
  {  // max across the two half-waves with one VALU lane swap (v_permlane32_swap) instead of an LDS round trip
    const unsigned u = __builtin_bit_cast(unsigned, mx);
    const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    mx = fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
  }
  if constexpr (FOLD) {
    if (first || !__all(mx <= 8.0f)) {
      const float d = first ? mx : fmaxf(mx, 0.f);
      const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-d);
      m_run += d;
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[kb][r] -= d;
#pragma unroll
      for (int r = 0; r < 16; ++r) minit[r] = -m_run;
    }
    float ps = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = __builtin_amdgcn_exp2f(sacc[kb][r]), p1 = __builtin_amdgcn_exp2f(sacc[kb][r + 1]);
        ps += p0 + p1;
        pf[kb][r >> 3][r & 7] = (T)p0;
        pf[kb][r >> 3][(r & 7) + 1] = (T)p1;
      }
    l_run += ps;
  } else {
    if (!__all((mx - m_run) * c <= 8.0f)) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
    }
    // packed fp32 math (v_pk_fma_f32 / v_pk_add_f32: two values per VALU issue) for the exp2 argument and the row sum
    const f32x2 c2 = {c, c};
    const f32x2 mc2 = {m_run * c, m_run * c};
    f32x2 ps2 = {0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 sv = {sacc[kb][r], sacc[kb][r + 1]};
        const f32x2 a = __builtin_elementwise_fma(sv, c2, -mc2);
        const f32x2 pv = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
        ps2 += pv;
        pf[kb][r >> 3][r & 7] = (T)pv[0];
        pf[kb][r >> 3][(r & 7) + 1] = (T)pv[1];
      }
    l_run += ps2[0] + ps2[1];
  }
}

// Q^T fragment of a query row, pre-multiplied by c when the scale is folded into the operand (softmax_block<FOLD = true>)
template <typename T, bool FOLD>
__device__ __forceinline__ vec8<T> load_q(const T* p, const float c) {
  vec8<T> q = *reinterpret_cast<const vec8<T>*>(p);
  if constexpr (FOLD) {
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = (T)((float)q[i] * c);
  }
  return q;
}

// VM = how the V tile reaches the PV product's A operand (V^T fragments: lane = one d row, 8 k-slots = keys):
//   0  V [key][d] is loaded into registers, transposed there and written to LDS as [d][key] (packed key pairs);
//   1  P.v is the pre-transposed, key-permuted V^T produced by lr_transpose_v_f16 ([B][heads*64][ldv], see below): the V tile
//      then takes the same LDS-DMA + XOR-swizzle path as K (no registers, no VALU packing) and every PV fragment is ONE ds_read_b128;
//   2  V stays [key][d] all the way: LDS-DMA like K, and the fragment is gathered by the LDS transpose read of gfx950
//      (ds_read_b64_tr_b16: a 16-lane group reads a [4 keys][16 d] block, 8 bytes per lane, and every lane receives ONE d column of
//      the four keys).  The four keys of a read are exactly one k-slot group of the S^T accumulator (key0 .. key0 + 3), so two reads
//      make a fragment and no transposed / permuted copy of V exists anywhere.  16-byte chunk c of row r sits in slot
//      c ^ (4 ((r >> 1) & 1)): the four rows of a read (128 bytes apart) then cover 64 distinct banks per half-wave.
// CAUSAL = true: key j is visible to query i only if j <= i (the text tower's attn_mask); every tile takes the masked path.
// EXACT = true: unscaled Q, softmax_block<FOLD = false> (the training forward with its log-sum-exp output).
// NQB = 32-query blocks per wave: 2 (256 queries per block, every K / V fragment feeds both) or 1 (128 queries per block: twice the
//       blocks at about half the registers, for shapes whose 256-query blocks leave the chip a badly filled last round).
template <typename T, int VM, bool CAUSAL = false, bool EXACT = false, int NQB = 2>
__global__ __launch_bounds__(ATT_THREADS) void attention_kernel(const AttnParams<T> P) {
  constexpr bool VT = VM == 1, TR = VM == 2;
#ifdef LR_ATTN_NOFOLD      // developer A/B build (tools/build_variant.sh nofold -DLR_ATTN_NOFOLD)
  constexpr bool FOLD = false;
#else
  constexpr bool FOLD = !EXACT && !CAUSAL;      // (the causal text-tower instance would need 272 registers with the fold: kept exact)
#endif
  __shared__ __attribute__((aligned(16))) char smem[2 * ATT_KB * 128 + 2 * 64 * VT_PITCH];
  char* Ksm = smem;
  char* Vsm = smem + 2 * ATT_KB * 128;

  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  int bid = blockIdx.x;
  {
    const int q = P.nblocks >> 3, r = P.nblocks & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int qt = bid % P.nqt;
  const int bh = bid / P.nqt;
  const int h = bh % P.heads, b = bh / P.heads;

  const T* qp = P.q + (size_t)b * P.Nq * P.ldq + h * 64;
  const T* kp = P.k + (size_t)b * P.Nkv * P.ldk + h * 64;
  const T* vp = VT ? P.v + ((size_t)b * P.heads + h) * 64 * P.ldv : P.v + (size_t)b * P.Nkv * P.ldv + h * 64;
  T* op = P.o + (size_t)b * P.Nq * P.ldo + h * 64;
  const T* zero = reinterpret_cast<const T*>(lr_zero_page);

  const int ql = lane & 31, hi = lane >> 5;
  // each wave owns 64 queries as two 32-query blocks that share every K / V fragment read from LDS
  int qrow[NQB];
  vec8<T> qf[NQB][4];   // Q^T fragments: B-operand of mfma(K, Q^T): lane holds Q[q][s*16 + hi*8 .. +8]
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    qrow[qb] = qt * (128 * NQB) + w * (32 * NQB) + qb * 32 + ql;
    const int qc = min(qrow[qb], P.Nq - 1);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
      qf[qb][s4] = load_q<T, FOLD>(qp + (size_t)qc * P.ldq + s4 * 16 + hi * 8, P.c);
  }

  const int ntiles = (P.Nkv + ATT_KB - 1) / ATT_KB;

  // K staging: 64 rows x 8 chunks = 512 x 16 B; wave w, instr i covers rows (i*4 + w)*8 .. +8
  auto stage_k = [&](int buf, int tile) {
    char* Ks = Ksm + buf * (ATT_KB * 128);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rbase = (i * 4 + w) * 8;
      const int row = rbase + (lane >> 3);
      const int key = tile * ATT_KB + row;
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      const T* g = key < P.Nkv ? kp + (size_t)key * P.ldk + chunk * 8 : zero;
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(Ks + rbase * 128), 16, 0, 0);
    }
  };
  // V^T staging (VT): rows = d, 128 bytes = the tile's 64 (permuted) keys; ldv is padded to whole tiles and zero filled
  auto stage_vt = [&](int buf, int tile) {
    char* Vs = Vsm + buf * (64 * VT_PITCH);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rbase = (i * 4 + w) * 8;
      const int row = rbase + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      const T* g = vp + (size_t)row * P.ldv + tile * ATT_KB + chunk * 8;
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(Vs + rbase * 128), 16, 0, 0);
    }
  };
  // natural V staging (TR): rows = keys like K; chunk slot = chunk ^ 4 ((row >> 1) & 1) (see the header of this kernel)
  auto stage_vn = [&](int buf, int tile) {
    char* Vs = Vsm + buf * (64 * VT_PITCH);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rbase = (i * 4 + w) * 8;
      const int row = rbase + (lane >> 3);
      const int key = tile * ATT_KB + row;
      const int chunk = (lane & 7) ^ (((row >> 1) & 1) << 2);
      const T* g = key < P.Nkv ? vp + (size_t)key * P.ldv + chunk * 8 : zero;
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(Vs + rbase * 128), 16, 0, 0);
    }
  };
  // per-lane byte offset of its 8 bytes inside a [4 keys][16 d] block of the transpose read: lane i of a 16-lane group supplies
  // key j = i >> 2, d columns 4 (i & 3) .. + 3; the group's d base is 16 ((lane >> 4) & 1) (+ 32 db), its key base 4 hi (+ ...)
  const int tr_j = (lane & 15) >> 2, tr_q = lane & 3;
  const int tr_off = (4 * hi + tr_j) * 128 + (tr_q & 1) * 8;
  const int tr_chunk = 2 * ((lane >> 4) & 1) + (tr_q >> 1);      // + 4 db, then ^ 4 (tr_j >> 1)
  // V staging through registers: thread owns key pair kpair = w*8 + lane/8 and d-chunk j = lane%8
  const int vj = lane & 7, vkp = w * 8 + (lane >> 3);
  auto load_v = [&](int tile, uint4& v0, uint4& v1) {
    const int key0 = tile * ATT_KB + 2 * vkp;
    v0 = key0 < P.Nkv ? *reinterpret_cast<const uint4*>(vp + (size_t)key0 * P.ldv + vj * 8) : make_uint4(0, 0, 0, 0);
    v1 = key0 + 1 < P.Nkv ? *reinterpret_cast<const uint4*>(vp + (size_t)(key0 + 1) * P.ldv + vj * 8)
                          : make_uint4(0, 0, 0, 0);
  };
  auto write_v = [&](int buf, const uint4& v0, const uint4& v1) {
    char* Vs = Vsm + buf * (64 * VT_PITCH);
    const unsigned a[4] = {v0.x, v0.y, v0.z, v0.w};
    const unsigned c[4] = {v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // halves (2i, 2i+1) of both keys -> rows d = 8*vj + 2i, +1 ; dword = (key0 value | key1 value << 16)
      const unsigned lo = (a[i] & 0xffffu) | (c[i] << 16);
      const unsigned hi2 = (a[i] >> 16) | (c[i] & 0xffff0000u);
      *reinterpret_cast<unsigned*>(Vs + (8 * vj + 2 * i) * VT_PITCH + vkp * 4) = lo;
      *reinterpret_cast<unsigned*>(Vs + (8 * vj + 2 * i + 1) * VT_PITCH + vkp * 4) = hi2;
    }
  };

  f32x16 oacc[NQB][2];
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[qb][d][r] = 0.f;
  float m_run[NQB], l_run[NQB];
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) { m_run[qb] = FOLD ? 0.f : -INFINITY; l_run[qb] = 0.f; }
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 minit[NQB];
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) minit[qb] = zero16;      // FOLD: -m_run of the query, the C operand of the first QK^T MFMA of every tile

  stage_k(0, 0);
  if constexpr (VT) {
    stage_vt(0, 0);
  } else if constexpr (TR) {
    stage_vn(0, 0);
  } else {
    uint4 v0, v1;
    load_v(0, v0, v1);
    write_v(0, v0, v1);
  }
  __syncthreads();

  // One K/V tile.  TAIL is a compile-time tag: only a partial last tile carries the key-index compares of the -inf mask
  // (left in the common path they cost ~70 VALU instructions per tile -- the compiler hoists them above the branch).
  auto process_tile = [&](int tile, auto tail_tag) {
    constexpr bool TAIL = decltype(tail_tag)::value || CAUSAL;
    const int cur = tile & 1;
    const bool more = tile + 1 < ntiles;
    uint4 nv0 = make_uint4(0, 0, 0, 0), nv1 = nv0;
    if (more) {
      stage_k(cur ^ 1, tile + 1);
      if constexpr (VT) stage_vt(cur ^ 1, tile + 1);
      else if constexpr (TR) stage_vn(cur ^ 1, tile + 1);
      else load_v(tile + 1, nv0, nv1);
    }
    const char* Ks = Ksm + cur * (ATT_KB * 128);
    const char* Vs = Vsm + cur * (64 * VT_PITCH);

    // ---- S^T[key][q] = sum_d K[key][d] Q[q][d]: four independent accumulator chains (2 query blocks x 2 key blocks),
    // every K fragment read from LDS feeds both query blocks
    f32x16 sacc[NQB][2];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int row = kb * 32 + ql;
        const int kc = s4 * 2 + hi;
        const vec8<T> kf = *reinterpret_cast<const vec8<T>*>(Ks + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb)
          sacc[qb][kb] = lr_mfma32(kf, qf[qb][s4], s4 == 0 ? (FOLD ? minit[qb] : zero16) : sacc[qb][kb]);
      }

    vec8<T> pf[NQB][2][2];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
      // ---- mask the tail tile: accumulator reg r of block kb is key kb*32 + (r&3) + 8*(r>>2) + 4*hi
      if constexpr (TAIL) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = tile * ATT_KB + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (key >= P.Nkv || (CAUSAL && key > qrow[qb])) sacc[qb][kb][r] = -INFINITY;
          }
      }
      softmax_block<T, FOLD>(sacc[qb], oacc[qb], pf[qb], m_run[qb], l_run[qb], minit[qb], P.c, tile == 0);
    }

    // ---- O^T[d][q] += sum_key V^T[d][key] P^T[key][q]; k-slot (hi*8 + jj) of MFMA (kb, tt) is key
    //      kb*32 + 16*tt + 4*hi + jj (jj < 4) and kb*32 + 16*tt + 8 + 4*hi + (jj - 4) (jj >= 4).
    //      Every V^T fragment feeds both query blocks.
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const int drow = db * 32 + ql;
          vec8<T> vf;
          if constexpr (VT) {
            const int vc = kb * 4 + tt * 2 + hi;     // 16-byte chunk = this lane's 8 k-slots, contiguous after the permute
            vf = *reinterpret_cast<const vec8<T>*>(Vs + drow * 128 + ((vc ^ ((drow >> 1) & 7)) << 4));
          } else if constexpr (TR) {
            typedef short s16x4 __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;
            const char* a = Vs + (kb * 32 + 16 * tt) * 128 + tr_off + (((4 * db + tr_chunk) ^ ((tr_j >> 1) << 2)) << 4);
            const vec4<T> va = __builtin_bit_cast(vec4<T>, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)a));
            const vec4<T> vb = __builtin_bit_cast(vec4<T>, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(a + 8 * 128)));
            vf[0] = va[0]; vf[1] = va[1]; vf[2] = va[2]; vf[3] = va[3];
            vf[4] = vb[0]; vf[5] = vb[1]; vf[6] = vb[2]; vf[7] = vb[3];
          } else {
            const int key0 = kb * 32 + 16 * tt + 4 * hi;
            const vec4<T> va = *reinterpret_cast<const vec4<T>*>(Vs + drow * VT_PITCH + key0 * 2);
            const vec4<T> vb = *reinterpret_cast<const vec4<T>*>(Vs + drow * VT_PITCH + (key0 + 8) * 2);
            vf[0] = va[0]; vf[1] = va[1]; vf[2] = va[2]; vf[3] = va[3];
            vf[4] = vb[0]; vf[5] = vb[1]; vf[6] = vb[2]; vf[7] = vb[3];
          }
#pragma unroll
          for (int qb = 0; qb < NQB; ++qb)
            oacc[qb][db] = lr_mfma32(vf, pf[qb][kb][tt], oacc[qb][db]);
        }
    if constexpr (VM == 0) {
      if (more) write_v(cur ^ 1, nv0, nv1);
    }
    __syncthreads();
  };
  const int nfull = P.Nkv / ATT_KB;
  for (int tile = 0; tile < nfull; ++tile) process_tile(tile, std::false_type{});
  if (nfull < ntiles) process_tile(nfull, std::true_type{});

  // ---- finalize: O[q][d] = O^T[d][q] / l ; lane holds d = db*32 + (r&3) + 8*(r>>2) + 4*hi
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    float lt = l_run[qb];
    {
      const unsigned u = __builtin_bit_cast(unsigned, lt);
      const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
      lt = __builtin_bit_cast(float, (unsigned)sw[0]) + __builtin_bit_cast(float, (unsigned)sw[1]);
    }
    const float inv = 1.0f / lt;
    if (P.lse && hi == 0 && qrow[qb] < P.Nq)
      P.lse[((size_t)b * P.heads + h) * P.Nq + qrow[qb]] = (FOLD ? m_run[qb] : m_run[qb] * P.c) + __builtin_amdgcn_logf(lt);
    // The two half-waves of a query (lane, lane ^ 32) hold the two 4-channel halves of every 8-channel group: one
    // v_permlane32_swap per dword hands group g to the lower and group g + 1 to the upper half-wave, so each lane stores
    // 16 contiguous bytes (8 x dwordx4 per lane instead of 16 x dwordx2: the store tail is issue-bound, and it is most of
    // the kernel for the 77-key cross-attention).
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        vec4<T> x, y;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          x[i] = (T)(oacc[qb][db][g * 4 + i] * inv);
          y[i] = (T)(oacc[qb][db][(g + 1) * 4 + i] * inv);
        }
        const uint2 xu = __builtin_bit_cast(uint2, x), yu = __builtin_bit_cast(uint2, y);
        const auto s0 = __builtin_amdgcn_permlane32_swap(xu.x, yu.x, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(xu.y, yu.y, false, false);
        // lower half-wave: (x.lo, x.hi) = channels 8 g .. 8 g + 7 ; upper half-wave: (y.lo, y.hi) = 8 (g + 1) .. + 7
        const uint4 o4 = make_uint4((unsigned)s0[0], (unsigned)s1[0], (unsigned)s0[1], (unsigned)s1[1]);
        if (qrow[qb] < P.Nq)
          *reinterpret_cast<uint4*>(op + (size_t)qrow[qb] * P.ldo + db * 32 + 8 * (g + hi)) = o4;
      }
  }
}

#ifdef LR_DEV_VARIANTS
// =====================================================================================================================
// Ping-pong variant for long key sequences (pre-transposed V, Nkv a multiple of 64): block = 8 waves = two GROUPS of four.
//
// Why: two co-resident 4-wave blocks of attention_kernel start together and run the same instruction stream, so the two
// waves of a SIMD sit in their MFMA stretch (QK^T, PV) at the same time and in their softmax stretch (VALU / exp) at the same
// time -- measured: time(MFMA + softmax) = time(MFMA) + time(softmax), a block alone on its CU runs 1.8x faster than one of a
// pair.  Here the two waves of every SIMD (waves w and w + 4 of the block) are held in ANTI-phase by two block-wide barriers
// per key tile: while group 0 issues the 32 MFMAs of [PV(j-1), QK^T(j)], group 1 runs the softmax of its tile j-1 on the VALU /
// transcendental pipes, then they swap (MI355X_MICROARCH.md, "Two waves per SIMD": the matrix pipe is per SIMD, VALU issue is
// arbitrated between the two waves -- complementary segments are what pays).
//
//   iteration j = 0 .. T (T key tiles)     group 0                         group 1
//     even phase                           PV(j-1) [j >= 1], QK^T(j) [j < T]    softmax(j-1) [j >= 1]
//     barrier (LDS-DMA stays in flight)
//     odd phase                            softmax(j) [j < T]              PV(j-1) [j >= 1], QK^T(j) [j < T]
//     vmcnt(0), barrier
//
// K / V^T tiles: 3-slot LDS ring of 16 KB stages (stage s = K tile s + V^T tile s), one 1-KB LDS-DMA of each per wave, issued at
// the top of iteration s - 1: K_s is read in iteration s, V_s in iteration s + 1, its slot is refilled in iteration s + 2.
// The arithmetic of a wave is instruction for instruction that of attention_kernel (same MFMA order per accumulator, same
// softmax code), so the two kernels agree bit for bit (tests/test_gpu_ops.py).
// NQB = 32-query blocks per wave (2: 512 queries per block; 1: 256 queries per block, for sequences too short to fill the chip
// with 512-query blocks).
// =====================================================================================================================
#define PP_THREADS 512
#define PP_STAGE_BYTES (2 * ATT_KB * 128)
#define PP_NSTAGE 3

template <typename T, int NQB>
__global__ __launch_bounds__(PP_THREADS) void attention_pp_kernel(const AttnParams<T> P) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) char smem[PP_NSTAGE * PP_STAGE_BYTES];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp = w >> 2;
  int bid = blockIdx.x;
  {
    const int q = P.nblocks >> 3, r = P.nblocks & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int qt = bid % P.nqt;
  const int bh = bid / P.nqt;
  const int h = bh % P.heads, b = bh / P.heads;

  const T* qp = P.q + (size_t)b * P.Nq * P.ldq + h * 64;
  const T* kp = P.k + (size_t)b * P.Nkv * P.ldk + h * 64;
  const T* vp = P.v + ((size_t)b * P.heads + h) * 64 * P.ldv;
  T* op = P.o + (size_t)b * P.Nq * P.ldo + h * 64;

  const int ql = lane & 31, hi = lane >> 5;
  int qrow[NQB];
  vec8<T> qf[NQB][4];
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    qrow[qb] = qt * (8 * 32 * NQB) + w * (32 * NQB) + qb * 32 + ql;
    const int qc = min(qrow[qb], P.Nq - 1);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
      qf[qb][s4] = load_q<T, true>(qp + (size_t)qc * P.ldq + s4 * 16 + hi * 8, P.c);
  }
  const int T_ = P.Nkv / ATT_KB;

  // stage s -> slot s % 3: wave w moves rows 8 w .. 8 w + 7 of the K tile and of the V^T tile (one 1-KB LDS-DMA each)
  const int srow = w * 8 + (lane >> 3);
  const int schunk = (lane & 7) ^ ((srow >> 1) & 7);
  const T* ksrc = kp + (size_t)srow * P.ldk + schunk * 8;
  const T* vsrc = vp + (size_t)srow * P.ldv + schunk * 8;
  auto stage = [&](int s) {
    char* Ks = smem + (s % PP_NSTAGE) * PP_STAGE_BYTES;
    char* Vs = Ks + ATT_KB * 128;
    __builtin_amdgcn_global_load_lds((gptr_t)(ksrc + (size_t)s * ATT_KB * P.ldk), (lptr_t)(Ks + w * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(vsrc + s * ATT_KB), (lptr_t)(Vs + w * 1024), 16, 0, 0);
  };

  f32x16 oacc[NQB][2];
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[qb][d][r] = 0.f;
  float m_run[NQB], l_run[NQB];
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) { m_run[qb] = 0.f; l_run[qb] = 0.f; }
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 minit[NQB];
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) minit[qb] = zero16;
  f32x16 sacc[NQB][2];
  vec8<T> pf[NQB][2][2];

  // ---- matrix segment: O^T += V^T(jv) P^T(jv)  (steps 0-3, PV) and S^T(jk) = K(jk) Q^T  (steps 4-7, QK).  Eight steps of two
  // fragment reads + four MFMAs (2 query blocks x 2 row blocks); the reads of step i + 1 are issued BEFORE the MFMAs of step i
  // (two register sets, order pinned with sched_barrier) -- left alone hipcc reads each fragment pair right before its MFMAs and
  // the matrix pipe idles through every LDS round trip.
  // Fragment addresses: chunk c = 2 sigma + hi of row ql (sigma = step & 3), swizzled slot c ^ ((ql >> 1) & 7); the row 32 + ql
  // has the same swizzle term, i.e. the second fragment of a step sits 4096 bytes further (an immediate offset).  The reads are
  // inline asm with hand-counted lgkmcnt waits: hipcc waits lgkmcnt(0) before every MFMA group here, i.e. also for the pair it
  // has just issued, and the matrix pipe would idle through an LDS round trip every other step.
  unsigned fa_off[4];
#pragma unroll
  for (int sg = 0; sg < 4; ++sg) fa_off[sg] = (unsigned)(ql * 128 + (((2 * sg + hi) ^ ((ql >> 1) & 7)) << 4));
  const unsigned lds0 = __builtin_bit_cast(unsigned, (lptr_t)smem);
  auto frag_read = [&](const unsigned addr, vec8<T>& f0, vec8<T>& f1) {
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:4096" : "=&v"(f0), "=&v"(f1) : "v"(addr));
  };
  // wait until at most N of this wave's LDS reads are outstanding; naming the fragments makes their consumers depend on the wait
  auto frag_wait2 = [&](vec8<T>& f0, vec8<T>& f1) { asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(f0), "+v"(f1)); };
  auto frag_wait0 = [&](vec8<T>& f0, vec8<T>& f1) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f0), "+v"(f1)); };
  auto frag_mma = [&](const int step, const vec8<T>& f0, const vec8<T>& f1) {
    if (step < 4) {
      const int kb = step >> 1, tt = step & 1;
#pragma unroll
      for (int qb = 0; qb < NQB; ++qb) {
        oacc[qb][0] = lr_mfma32(f0, pf[qb][kb][tt], oacc[qb][0]);
        oacc[qb][1] = lr_mfma32(f1, pf[qb][kb][tt], oacc[qb][1]);
      }
    } else {
      const int s4 = step - 4;
#pragma unroll
      for (int qb = 0; qb < NQB; ++qb) {
        sacc[qb][0] = lr_mfma32(f0, qf[qb][s4], s4 == 0 ? minit[qb] : sacc[qb][0]);
        sacc[qb][1] = lr_mfma32(f1, qf[qb][s4], s4 == 0 ? minit[qb] : sacc[qb][1]);
      }
    }
  };
  // FIRST / LAST: compile-time step range (0..7 both products, 4..7 only QK: the first tile, 0..3 only PV: the drain)
  auto matrix_segment = [&](auto first_tag, auto last_tag, int jv, int jk) {
    constexpr int S0 = decltype(first_tag)::value, S1 = decltype(last_tag)::value;
    const unsigned vbase = lds0 + (unsigned)((jv % PP_NSTAGE) * PP_STAGE_BYTES + ATT_KB * 128);
    const unsigned kbase = lds0 + (unsigned)((jk % PP_NSTAGE) * PP_STAGE_BYTES);
    auto addr = [&](const int step) -> unsigned { return (step < 4 ? vbase : kbase) + fa_off[step & 3]; };
    vec8<T> fa[2], fb[2];
    frag_read(addr(S0), fa[0], fa[1]);
#pragma unroll
    for (int step = S0; step < S1; step += 2) {
      frag_read(addr(step + 1), fb[0], fb[1]);
      frag_wait2(fa[0], fa[1]);
      __builtin_amdgcn_sched_barrier(0);
      frag_mma(step, fa[0], fa[1]);
      __builtin_amdgcn_sched_barrier(0);
      if (step + 2 < S1) {
        frag_read(addr(step + 2), fa[0], fa[1]);
        frag_wait2(fb[0], fb[1]);
      } else {
        frag_wait0(fb[0], fb[1]);
      }
      __builtin_amdgcn_sched_barrier(0);
      frag_mma(step + 1, fb[0], fb[1]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  using IC0 = std::integral_constant<int, 0>;
  using IC4 = std::integral_constant<int, 4>;
  using IC8 = std::integral_constant<int, 8>;
  // ---- softmax segment: S^T(j) -> P^T(j) (fp16 / bf16, the PV B-operand), running max / sum, deferred rescale of O
  auto softmax_segment = [&](const bool first) {
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb)
      softmax_block<T, true>(sacc[qb], oacc[qb], pf[qb], m_run[qb], l_run[qb], minit[qb], P.c, first);
  };

  // phase ends: `mid` keeps the LDS-DMA of the next stage in flight, `end` retires it (it is first read in the next iteration)
  // (sched_barrier(0) on both sides: the scheduler may not move register-only work -- softmax VALU, MFMAs -- across the barrier,
  // the whole point of which is to keep the two groups in complementary segments)
  auto mid = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto end = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  stage(0);
  end();
  // The two groups run separate loops with the same barrier count (2 per iteration, T_ + 1 iterations), so in each loop only
  // ONE of the big register arrays is carried across iterations (group 0: P^T, group 1: S^T).
  if (grp == 0) {
    if (T_ > 1) stage(1);
    matrix_segment(IC4{}, IC8{}, 0, 0); mid();
    softmax_segment(true); end();
    for (int j = 1; j < T_; ++j) {
      if (j + 1 < T_) stage(j + 1);
      matrix_segment(IC0{}, IC8{}, j - 1, j); mid();
      softmax_segment(false); end();
    }
    matrix_segment(IC0{}, IC4{}, T_ - 1, 0); mid();
    end();
  } else {
    if (T_ > 1) stage(1);
    mid();
    matrix_segment(IC4{}, IC8{}, 0, 0); end();
    for (int j = 1; j < T_; ++j) {
      if (j + 1 < T_) stage(j + 1);
      softmax_segment(j == 1); mid();
      matrix_segment(IC0{}, IC8{}, j - 1, j); end();
    }
    softmax_segment(T_ == 1); mid();
    matrix_segment(IC0{}, IC4{}, T_ - 1, 0); end();
  }

#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    float lt = l_run[qb];
    {
      const unsigned u = __builtin_bit_cast(unsigned, lt);
      const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
      lt = __builtin_bit_cast(float, (unsigned)sw[0]) + __builtin_bit_cast(float, (unsigned)sw[1]);
    }
    const float inv = 1.0f / lt;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        vec4<T> x, y;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          x[i] = (T)(oacc[qb][db][g * 4 + i] * inv);
          y[i] = (T)(oacc[qb][db][(g + 1) * 4 + i] * inv);
        }
        const uint2 xu = __builtin_bit_cast(uint2, x), yu = __builtin_bit_cast(uint2, y);
        const auto s0 = __builtin_amdgcn_permlane32_swap(xu.x, yu.x, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(xu.y, yu.y, false, false);
        const uint4 o4 = make_uint4((unsigned)s0[0], (unsigned)s1[0], (unsigned)s0[1], (unsigned)s1[1]);
        if (qrow[qb] < P.Nq)
          *reinterpret_cast<uint4*>(op + (size_t)qrow[qb] * P.ldo + db * 32 + 8 * (g + hi)) = o4;
      }
  }
#endif
}

#endif  // LR_DEV_VARIANTS

// Developer build only (LR_DEV, common.h):
//   LR_ATTN_PP: 0 (default) = attention_kernel for every shape; 1 = the ping-pong kernel with the block size picked by the rule in
//     launch_attention, 2 / 3 = with 512- / 256-query blocks forced.  Measured on MI355X (profiles/r04_attn_pingpong.txt): 8192^2 753 us
//     (attention_kernel) vs 860 / 797 us, 2048^2 116 vs 151 / 121, 20480^2 1408 vs 1383 / 1495 -- the complementary-segment schedule does
//     not pay at d_head = 64, so attention_pp_kernel is not compiled into the product library.
//   LR_ATTN_NQB = 1 | 2 forces 128- / 256-query blocks for lr_attention_f16 on natural V; unset: the rule in launch_attention
//   LR_ATTN_TR = 0 sends natural-layout V through the register transpose (VM = 0) instead of the LDS transpose read
#ifdef LR_DEV_VARIANTS
static int attn_pp_mode() { return LR_DEV("LR_ATTN_PP", 0); }
#endif
static int attn_nqb_mode() { return LR_DEV("LR_ATTN_NQB", 0); }
static bool attn_tr_mode() { return LR_DEV("LR_ATTN_TR", 1) != 0; }

template <typename T>
static int launch_attention(const lr_half* q, int ldq, const lr_half* k, int ldk, const lr_half* v, int ldv, lr_half* o,
                            int ldo, int B, int heads, int Nq, int Nkv, float scale, lr_stream_t s, bool vt,
                            float* lse = nullptr) {
  if (!q || !k || !v || !o || B <= 0 || heads <= 0 || Nq <= 0 || Nkv <= 0) return LR_E_ARG;
  if (ldq % 8 || ldk % 8 || ldv % 8 || ldo % 8) return LR_E_ALIGN;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15) return LR_E_ALIGN;
  if (vt && (ldv % ATT_KB || ldv < Nkv)) return LR_E_ALIGN;
  AttnParams<T> P;
  P.q = (const T*)q; P.k = (const T*)k; P.v = (const T*)v; P.o = (T*)o;
  P.ldq = ldq; P.ldk = ldk; P.ldv = ldv; P.ldo = ldo;
  P.heads = heads; P.Nq = Nq; P.Nkv = Nkv;
  P.nqt = (Nq + ATT_QB - 1) / ATT_QB;
  P.nblocks = P.nqt * heads * B;
  P.c = scale * 1.44269504088896340736f;
  P.lse = lse;
#ifdef LR_DEV_VARIANTS
  const int pp = attn_pp_mode();
  if (vt && !lse && pp && Nkv % ATT_KB == 0 && Nkv >= 4 * ATT_KB) {
    // ping-pong kernel: 512-query blocks when they fill the chip for at least two rounds, else 256-query blocks
    const int nb2 = ((Nq + 511) / 512) * heads * B;
    const bool two = pp == 2 || (pp == 1 && nb2 >= 512);
    if (two) {
      P.nqt = (Nq + 511) / 512; P.nblocks = P.nqt * heads * B;
      hipLaunchKernelGGL((attention_pp_kernel<T, 2>), dim3(P.nblocks), dim3(PP_THREADS), 0, (hipStream_t)s, P);
    } else {
      P.nqt = (Nq + 255) / 256; P.nblocks = P.nqt * heads * B;
      hipLaunchKernelGGL((attention_pp_kernel<T, 1>), dim3(P.nblocks), dim3(PP_THREADS), 0, (hipStream_t)s, P);
    }
    return lr_launch_status();
  }
#endif
  if (lse) {      // training forward: exact scale handling, the backward recomputes P from the unscaled operands and this log-sum-exp
    if (vt) return LR_E_UNSUPPORTED;
    hipLaunchKernelGGL((attention_kernel<T, 2, false, true>), dim3(P.nblocks), dim3(ATT_THREADS), 0, (hipStream_t)s, P);
  } else if (vt) hipLaunchKernelGGL((attention_kernel<T, 1>), dim3(P.nblocks), dim3(ATT_THREADS), 0, (hipStream_t)s, P);
  else if (attn_tr_mode()) {
    // 256-query blocks (two 32-query blocks per wave) run two per CU, 128-query blocks three per CU (142 registers).  The kernel is
    // VALU-bound: a CU delivers about the same throughput with one, two or three resident blocks (a block alone runs ~1.8x faster than
    // one of a pair), so what costs time is a last round that leaves CUs EMPTY for a whole block time.  Small cost model in units of the
    // time of a paired 256-query block (measured shapes: profiles/r04_attn_nqb.txt): the block size with the shorter estimate wins,
    // 128-query blocks only with a 5 % margin (they share no K / V fragment between query blocks: +8 % work).
    auto tail = [](int r, int per_cu, float full, float two, float one) {      // cost of a partial round of r blocks
      return r == 0 ? 0.f : r <= 256 ? one : (per_cu == 2 || r <= 512) ? two : full;
    };
    const int nb2 = P.nblocks, nb1 = ((Nq + 127) / 128) * heads * B;
    const float t2 = (float)(nb2 / 512) + tail(nb2 % 512, 2, 1.f, 1.f, 0.556f);
    const float t1 = 1.08f * (0.75f * (float)(nb1 / 768) + tail(nb1 % 768, 3, 0.75f, 0.5f, 0.28f));
    const int mode = attn_nqb_mode();
    const bool small = mode == 1 || (mode == 0 && t1 < 0.95f * t2);
    if (small) {
      P.nqt = (Nq + 127) / 128; P.nblocks = P.nqt * heads * B;
      hipLaunchKernelGGL((attention_kernel<T, 2, false, false, 1>), dim3(P.nblocks), dim3(ATT_THREADS), 0, (hipStream_t)s, P);
    } else {
      hipLaunchKernelGGL((attention_kernel<T, 2>), dim3(P.nblocks), dim3(ATT_THREADS), 0, (hipStream_t)s, P);
    }
  }
#ifdef LR_DEV_VARIANTS
  else hipLaunchKernelGGL((attention_kernel<T, 0>), dim3(P.nblocks), dim3(ATT_THREADS), 0, (hipStream_t)s, P);
#endif
  return lr_launch_status();
}

template <typename T>
static int lr_attention_t(const lr_half* q, int ldq, const lr_half* k, int ldk, const lr_half* v, int ldv,
                                lr_half* o, int ldo, int B, int heads, int Nq, int Nkv, float scale, lr_stream_t s) {
  return launch_attention<T>(q, ldq, k, ldk, v, ldv, o, ldo, B, heads, Nq, Nkv, scale, s, false);
}

template <typename T>
static int lr_attention_causal_t(const lr_half* q, int ldq, const lr_half* k, int ldk, const lr_half* v, int ldv,
                                       lr_half* o, int ldo, int B, int heads, int N, float scale, lr_stream_t s) {
  if (!q || !k || !v || !o || B <= 0 || heads <= 0 || N <= 0) return LR_E_ARG;
  if (ldq % 8 || ldk % 8 || ldv % 8 || ldo % 8) return LR_E_ALIGN;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15) return LR_E_ALIGN;
  AttnParams<T> P;
  P.q = (const T*)q; P.k = (const T*)k; P.v = (const T*)v; P.o = (T*)o;
  P.ldq = ldq; P.ldk = ldk; P.ldv = ldv; P.ldo = ldo;
  P.heads = heads; P.Nq = N; P.Nkv = N;
  P.nqt = (N + ATT_QB - 1) / ATT_QB;
  P.nblocks = P.nqt * heads * B;
  P.c = scale * 1.44269504088896340736f;
  P.lse = nullptr;
  hipLaunchKernelGGL((attention_kernel<T, 2, true>), dim3(P.nblocks), dim3(ATT_THREADS), 0, (hipStream_t)s, P);
  return lr_launch_status();
}

template <typename T>
static int lr_attention_lse_t(const lr_half* q, int ldq, const lr_half* k, int ldk, const lr_half* v, int ldv,
                                    lr_half* o, int ldo, float* lse, int B, int heads, int Nq, int Nkv, float scale,
                                    lr_stream_t s) {
  if (!lse) return LR_E_ARG;
  return launch_attention<T>(q, ldq, k, ldk, v, ldv, o, ldo, B, heads, Nq, Nkv, scale, s, false, lse);
}

template <typename T>
static int lr_attention_vt_t(const lr_half* q, int ldq, const lr_half* k, int ldk, const lr_half* vt, int ld_vt,
                                   lr_half* o, int ldo, int B, int heads, int Nq, int Nkv, float scale, lr_stream_t s) {
  return launch_attention<T>(q, ldq, k, ldk, vt, ld_vt, o, ldo, B, heads, Nq, Nkv, scale, s, true);
}

// ---------------------------------------------------------------------------------------------------------------
// V [B][Nkv][ldv] (head h = columns h*64 .. +64)  ->  V^T [B][heads*64][ld_vt], ld_vt = Nkv rounded up to 64, tail keys
// zero.  Within every group of 16 keys the order is [0-3, 8-11, 4-7, 12-15]: the 8 k-slots an MFMA lane needs for
// P^T (the S^T accumulator's key order, see attention_kernel) become one contiguous 16-byte piece.
// grid = (key tiles, heads, B), block = 256.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void transpose_v_kernel(const T* __restrict__ v, int ldv, T* __restrict__ vt, int ld_vt,
                                                          int heads, int Nkv) {
  __shared__ T tile[64][72];     // [key][d], 144-byte pitch
  const int t = threadIdx.x;
  const int kt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const T* src = v + (size_t)b * Nkv * ldv + h * 64;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int key = i * 32 + (t >> 3), j = t & 7;
    const int gk = kt * 64 + key;
    uint4 u = make_uint4(0, 0, 0, 0);
    if (gk < Nkv) u = *reinterpret_cast<const uint4*>(src + (size_t)gk * ldv + j * 8);
    *reinterpret_cast<uint4*>(&tile[key][j * 8]) = u;
  }
  __syncthreads();
  const int d = t >> 2, g = t & 3;      // output row d, 16-key group g
  T r[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int ko = (i & 3) + ((i >> 2) == 0 ? 0 : (i >> 2) == 1 ? 8 : (i >> 2) == 2 ? 4 : 12);
    r[i] = tile[g * 16 + ko][d];
  }
  T* dst = vt + (((size_t)b * heads + h) * 64 + d) * ld_vt + kt * 64 + g * 16;
  *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(&r[0]);
  *reinterpret_cast<uint4*>(dst + 8) = *reinterpret_cast<const uint4*>(&r[8]);
}

template <typename T>
static int lr_transpose_v_t(const lr_half* v, int ldv, lr_half* vt, int ld_vt, int B, int heads, int Nkv,
                                  lr_stream_t s) {
  if (!v || !vt || B <= 0 || heads <= 0 || Nkv <= 0) return LR_E_ARG;
  if (ldv % 8 || ld_vt % ATT_KB || ld_vt < Nkv || (((uintptr_t)v | (uintptr_t)vt) & 15)) return LR_E_ALIGN;
  dim3 grid(ld_vt / ATT_KB, heads, B);
  hipLaunchKernelGGL(transpose_v_kernel<T>, grid, dim3(256), 0, (hipStream_t)s, (const T*)v, ldv, (T*)vt, ld_vt, heads, Nkv);
  return lr_launch_status();
}

// ---- C ABI: every entry point in its fp16 and bf16 form -------------------------------------------------------------
extern "C" int lr_attention_f16(const lr_half* q, int ldq, const lr_half* k, int ldk, const lr_half* v, int ldv, lr_half* o, int ldo, int B, int heads, int Nq, int Nkv, float scale, lr_stream_t s) { return lr_attention_t<f16>(q, ldq, k, ldk, v, ldv, o, ldo, B, heads, Nq, Nkv, scale, s); }
extern "C" int lr_attention_bf16(const lr_half* q, int ldq, const lr_half* k, int ldk, const lr_half* v, int ldv, lr_half* o, int ldo, int B, int heads, int Nq, int Nkv, float scale, lr_stream_t s) { return lr_attention_t<bf16>(q, ldq, k, ldk, v, ldv, o, ldo, B, heads, Nq, Nkv, scale, s); }
extern "C" int lr_attention_causal_f16(const lr_half* q, int ldq, const lr_half* k, int ldk, const lr_half* v, int ldv, lr_half* o, int ldo, int B, int heads, int N, float scale, lr_stream_t s) { return lr_attention_causal_t<f16>(q, ldq, k, ldk, v, ldv, o, ldo, B, heads, N, scale, s); }
extern "C" int lr_attention_causal_bf16(const lr_half* q, int ldq, const lr_half* k, int ldk, const lr_half* v, int ldv, lr_half* o, int ldo, int B, int heads, int N, float scale, lr_stream_t s) { return lr_attention_causal_t<bf16>(q, ldq, k, ldk, v, ldv, o, ldo, B, heads, N, scale, s); }
extern "C" int lr_attention_lse_f16(const lr_half* q, int ldq, const lr_half* k, int ldk, const lr_half* v, int ldv, lr_half* o, int ldo, float* lse, int B, int heads, int Nq, int Nkv, float scale, lr_stream_t s) { return lr_attention_lse_t<f16>(q, ldq, k, ldk, v, ldv, o, ldo, lse, B, heads, Nq, Nkv, scale, s); }
extern "C" int lr_attention_lse_bf16(const lr_half* q, int ldq, const lr_half* k, int ldk, const lr_half* v, int ldv, lr_half* o, int ldo, float* lse, int B, int heads, int Nq, int Nkv, float scale, lr_stream_t s) { return lr_attention_lse_t<bf16>(q, ldq, k, ldk, v, ldv, o, ldo, lse, B, heads, Nq, Nkv, scale, s); }
extern "C" int lr_attention_vt_f16(const lr_half* q, int ldq, const lr_half* k, int ldk, const lr_half* vt, int ld_vt, lr_half* o, int ldo, int B, int heads, int Nq, int Nkv, float scale, lr_stream_t s) { return lr_attention_vt_t<f16>(q, ldq, k, ldk, vt, ld_vt, o, ldo, B, heads, Nq, Nkv, scale, s); }
extern "C" int lr_attention_vt_bf16(const lr_half* q, int ldq, const lr_half* k, int ldk, const lr_half* vt, int ld_vt, lr_half* o, int ldo, int B, int heads, int Nq, int Nkv, float scale, lr_stream_t s) { return lr_attention_vt_t<bf16>(q, ldq, k, ldk, vt, ld_vt, o, ldo, B, heads, Nq, Nkv, scale, s); }
extern "C" int lr_transpose_v_f16(const lr_half* v, int ldv, lr_half* vt, int ld_vt, int B, int heads, int Nkv, lr_stream_t s) { return lr_transpose_v_t<f16>(v, ldv, vt, ld_vt, B, heads, Nkv, s); }
extern "C" int lr_transpose_v_bf16(const lr_half* v, int ldv, lr_half* vt, int ld_vt, int B, int heads, int Nkv, lr_stream_t s) { return lr_transpose_v_t<bf16>(v, ldv, vt, ld_vt, B, heads, Nkv, s); }
