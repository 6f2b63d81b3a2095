// Fused scaled-dot-product attention forward (flash-style online softmax), d_head = 64, fp16 in / fp32 accumulate.
//
// Work split: block = 4 waves = 128 queries of one (batch, head); each wave owns 32 queries and streams the K/V
// sequence in 64-key tiles.  Per tile and wave: S^T = K Q^T (8 x mfma 32x32x16) -> online softmax in registers ->
// O^T += V^T P^T (8 x mfma 32x32x16).
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
// * Software pipeline inside each wave: the QK^T MFMAs of tile j+1 are issued ahead of the softmax VALU stream of
//   tile j, so the matrix pipe is busy while exp2 / max / fp16 convert run; the running-max rescale of O is deferred
//   until a row max grows by more than 2^8 (exp2 domain).
#include "common.h"

#define ATT_THREADS 256
#define ATT_QB 128
#define ATT_KB 64
#define VT_PITCH 136  // bytes per V^T row (64 keys * 2 B + 8 pad)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct AttnParams {
  const f16* q; const f16* k; const f16* v; f16* o;
  int ldq, ldk, ldv, ldo, heads, Nq, Nkv, nqt, nblocks;
  float c;  // scale * log2(e)
};

__global__ __launch_bounds__(ATT_THREADS) void attention_kernel(const AttnParams P) {
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

  const f16* qp = P.q + (size_t)b * P.Nq * P.ldq + h * 64;
  const f16* kp = P.k + (size_t)b * P.Nkv * P.ldk + h * 64;
  const f16* vp = P.v + (size_t)b * P.Nkv * P.ldv + h * 64;
  f16* op = P.o + (size_t)b * P.Nq * P.ldo + h * 64;
  const f16* zero = reinterpret_cast<const f16*>(lr_zero_page);

  const int ql = lane & 31, hi = lane >> 5;
  const int qrow = qt * ATT_QB + w * 32 + ql;
  const int qrow_c = min(qrow, P.Nq - 1);
  // Q^T fragments: B-operand of mfma(K, Q^T): lane holds Q[q][s*16 + hi*8 .. +8]
  f16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
    qf[s] = *reinterpret_cast<const f16x8*>(qp + (size_t)qrow_c * P.ldq + s * 16 + hi * 8);

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
      const f16* g = key < P.Nkv ? kp + (size_t)key * P.ldk + chunk * 8 : zero;
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(Ks + rbase * 128), 16, 0, 0);
    }
  };
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

  f32x16 oacc[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // S^T[key][q] = sum_d K[key][d] Q[q][d] for one 64-key tile held in K buffer `buf`
  auto qk = [&](f32x16 (&sacc)[2], int buf) {
    const char* Ks = Ksm + buf * (ATT_KB * 128);
    // the two 32-key blocks are independent accumulator chains: alternate them so consecutive MFMAs never wait on
    // each other's 16-pass latency
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int row = kb * 32 + ql;
        const int kc = s4 * 2 + hi;
        const f16x8 kf = *reinterpret_cast<const f16x8*>(Ks + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
        sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s4], s4 == 0 ? zero16 : sacc[kb], 0, 0, 0);
      }
    }
  };

  // One pipeline step: the QK^T MFMAs of tile+1 are issued BEFORE the softmax VALU work of `tile` (independent
  // registers), so the matrix pipe runs under the exp/max/convert stream of the same wave; then P V of `tile`.
  //   K ring: K[tile+1] is read here, K[tile+2] is loaded into the buffer K[tile] vacated (its reads finished before
  //   the barrier that ended the previous step).  V ring: V[tile] read, V[tile+1] written after the P V MFMAs.
  auto step = [&](f32x16 (&sc)[2], f32x16 (&sn)[2], int tile) {
    const bool more1 = tile + 1 < ntiles, more2 = tile + 2 < ntiles;
    uint4 nv0 = make_uint4(0, 0, 0, 0), nv1 = nv0;
    if (more2) stage_k(tile & 1, tile + 2);
    if (more1) load_v(tile + 1, nv0, nv1);
    if (more1) qk(sn, (tile + 1) & 1);

    // ---- mask the tail tile: accumulator reg r of block kb is key kb*32 + (r&3) + 8*(r>>2) + 4*hi
    if (tile * ATT_KB + ATT_KB > P.Nkv) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = tile * ATT_KB + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key >= P.Nkv) sc[kb][r] = -INFINITY;
        }
    }
    // ---- online softmax, one query per lane (partner lane ^ 32 holds the other half of the keys).
    // The running max is only raised (and O, l rescaled) when some row's max grew by more than 2^8 in the exp2
    // domain: P stays <= 256, exactly representable headroom in fp16, and the common path skips 32 accumulator
    // multiplies per lane.  m_run starts at -inf, so the first tile always takes the rescale path.
    float mx = sc[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[kb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (!__all((mx - m_run) * P.c <= 8.0f)) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * P.c);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
    }
    const float mc = m_run * P.c;
    float psum = 0.f;
    f16x8 pf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(sc[kb][r], P.c, -mc));
        psum += p;
        pf[kb][r >> 3][r & 7] = (f16)p;
      }
    l_run += psum;

    // ---- O^T[d][q] += sum_key V^T[d][key] P^T[key][q]; k-slot (hi*8 + jj) of MFMA (kb, tt) is key
    //      kb*32 + 16*tt + 4*hi + jj (jj < 4) and kb*32 + 16*tt + 8 + 4*hi + (jj - 4) (jj >= 4)
    const char* Vs = Vsm + (tile & 1) * (64 * VT_PITCH);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const int drow = db * 32 + ql;
          const int key0 = kb * 32 + 16 * tt + 4 * hi;
          const f16x4 va = *reinterpret_cast<const f16x4*>(Vs + drow * VT_PITCH + key0 * 2);
          const f16x4 vb = *reinterpret_cast<const f16x4*>(Vs + drow * VT_PITCH + (key0 + 8) * 2);
          f16x8 vf;
          vf[0] = va[0]; vf[1] = va[1]; vf[2] = va[2]; vf[3] = va[3];
          vf[4] = vb[0]; vf[5] = vb[1]; vf[6] = vb[2]; vf[7] = vb[3];
          oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kb][tt], oacc[db], 0, 0, 0);
        }
    if (more1) write_v((tile + 1) & 1, nv0, nv1);
    __syncthreads();
  };

  {
    uint4 v0, v1;
    stage_k(0, 0);
    if (ntiles > 1) stage_k(1, 1);
    load_v(0, v0, v1);
    write_v(0, v0, v1);
  }
  __syncthreads();
  f32x16 sA[2], sB[2];
  qk(sA, 0);
  __syncthreads();   // nobody may overwrite K buffer 0 (tile 2) before every wave has finished its first QK^T
  for (int tile = 0; tile < ntiles; tile += 2) {
    step(sA, sB, tile);
    if (tile + 1 < ntiles) step(sB, sA, tile + 1);
  }

  // ---- finalize: O[q][d] = O^T[d][q] / l ; lane holds d = db*32 + (r&3) + 8*(r>>2) + 4*hi
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (qrow < P.Nq) {
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f16x4 ov;
#pragma unroll
        for (int i = 0; i < 4; ++i) ov[i] = (f16)(oacc[db][g * 4 + i] * inv);
        *reinterpret_cast<f16x4*>(op + (size_t)qrow * P.ldo + db * 32 + 8 * g + 4 * hi) = ov;
      }
  }
}

extern "C" int lr_attention_f16(const lr_half* q, int ldq, const lr_half* k, int ldk, const lr_half* v, int ldv,
                                lr_half* o, int ldo, int B, int heads, int Nq, int Nkv, float scale, lr_stream_t s) {
  if (!q || !k || !v || !o || B <= 0 || heads <= 0 || Nq <= 0 || Nkv <= 0) return LR_E_ARG;
  if (ldq % 8 || ldk % 8 || ldv % 8 || ldo % 4) return LR_E_ALIGN;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15 || ((uintptr_t)o & 7)) return LR_E_ALIGN;
  AttnParams P;
  P.q = (const f16*)q; P.k = (const f16*)k; P.v = (const f16*)v; P.o = (f16*)o;
  P.ldq = ldq; P.ldk = ldk; P.ldv = ldv; P.ldo = ldo;
  P.heads = heads; P.Nq = Nq; P.Nkv = Nkv;
  P.nqt = (Nq + ATT_QB - 1) / ATT_QB;
  P.nblocks = P.nqt * heads * B;
  P.c = scale * 1.44269504088896340736f;
  hipLaunchKernelGGL(attention_kernel, dim3(P.nblocks), dim3(ATT_THREADS), 0, (hipStream_t)s, P);
  return lr_launch_status();
}
