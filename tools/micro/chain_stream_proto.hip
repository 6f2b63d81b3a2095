// Prototype of the memory / matrix skeleton of a register-chained transformer-block kernel at C = 640 (level 1 of the SD2 UNet, M = 16384
// rows), to price the design VERDICT r3 #1 asks for in TIME units before building it:
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/chain_proto tools/micro/chain_stream_proto.hip && /tmp/chain_proto
//
// A row-local chain (attn1.to_out + x -> LayerNorm -> to_q -> 77-key attention -> to_out + x, or LayerNorm -> GEGLU -> ff2) needs whole
// rows per block, and 16384 rows fill 256 CUs only with 64-row blocks.  Every block then streams ALL weights of the chain through its
// LDS ring (xattn chain: 3 x 640 x 640 + K / V of 77 keys = 2.65 MB; feed-forward chain: 5120 x 640 + 640 x 2560 + 640 x 640 = 10.6 MB)
// and multiplies each 40 KB piece (320 weight rows x 64 k) against its 64 rows: 8 waves = 4 row tiles x 2 column halves, 20 fragment
// reads + 20 v_mfma_f32_16x16x32_f16 per wave and piece -- the structure of xattn_block / ffn_block (3-slot ring, 16-byte LDS-DMA, one
// barrier per piece, counted vmcnt), with the per-row work (row loads, LayerNorm, softmax, gate, epilogue) left out.  The measured time
// is therefore a LOWER bound of the chained kernel; it is compared with the launches it would replace (profiles/r04_chain_proto.txt).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#define PIECE (320 * 128)      // 40 KB: 320 weight rows x 64 k (fp16)
#define NSLOT 3

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__global__ __launch_bounds__(512) void chain_kernel(const char* __restrict__ w, int npieces, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int ch = wv >> 2;                       // column half: weight rows 160 ch .. 160 ch + 159 of the piece
  const int fr = lane & 15, fq = lane >> 4;
  // a piece = 40 wave-instructions of 1 KB: 5 per wave; source-side XOR swizzle like the GEMM (rows of 128 B)
  auto stage = [&](int slot, int p) {
    const char* src = w + (size_t)p * PIECE;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int row = (i * 8 + wv) * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      __builtin_amdgcn_global_load_lds((gptr_t)(src + row * 128 + chunk * 16), (lptr_t)(smem + slot * PIECE + (i * 8 + wv) * 1024), 16, 0, 0);
    }
  };
  f32x4 acc[10];
#pragma unroll
  for (int j = 0; j < 10; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f16x8 xb[2];      // the wave's activation fragments (registers in the real kernel)
#pragma unroll
  for (int i = 0; i < 8; ++i) { xb[0][i] = (_Float16)(0.01f * (lane + i)); xb[1][i] = (_Float16)(0.02f * (lane - i)); }
  stage(0, 0);
  if (npieces > 1) stage(1, 1);
  for (int p = 0; p < npieces; ++p) {
    if (p + 1 < npieces) wait_vmcnt<5>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (p + 2 < npieces) stage((p + 2) % NSLOT, p + 2);
    const char* S = smem + (p % NSLOT) * PIECE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 10; ++j) {
        const int row = ch * 160 + j * 16 + fr;
        const int kc = ks * 4 + fq;
        const f16x8 wf = *reinterpret_cast<const f16x8*>(S + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xb[ks], acc[j], 0, 0, 0);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 10; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[blockIdx.x * 512 + t] = s;
}

int main() {
  const size_t maxb = 16u << 20;
  char* w; float* out;
  hipMalloc(&w, maxb); hipMalloc(&out, 256 * 512 * 4);
  std::vector<_Float16> h(maxb / 2);
  srand(3);
  for (auto& x : h) x = (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f);
  hipMemcpy(w, h.data(), maxb, hipMemcpyHostToDevice);
  hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, NSLOT * PIECE);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct Cfg { const char* what; double bytes; double replaced_us; };
  const Cfg cfgs[] = {
      {"level 1 (M = 16384, C = 640) attn1.to_out + LN + to_q + 77-key attention + to_out: 3 x 640x640 + K|V", 3 * 640 * 640 * 2.0 + 2 * 77 * 640 * 2.0, 28.4 * 3 + 20.1},
      {"level 1 (M = 16384, C = 640) LN + GEGLU 5120x640 + ff2 640x2560 + proj_out 640x640", (5120.0 * 640 + 640.0 * 2560 + 640.0 * 640) * 2, 131.4 + 60.5 + 28.4},
  };
  for (const auto& c : cfgs) {
    const int npieces = (int)((c.bytes + PIECE - 1) / PIECE);
    for (int blocks : {256}) {
      auto launch = [&]() { hipLaunchKernelGGL(chain_kernel, dim3(blocks), dim3(512), NSLOT * PIECE, 0, w, npieces, out); };
      launch(); hipDeviceSynchronize();
      float best = 1e9f;
      for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
      }
      const double flops = (double)blocks * 8 * npieces * 20 * 16384.0;
      printf("%s\n   %d blocks of 64 rows, %d pieces of 40 KB per block (%.2f MB): %.1f us  (L2 -> LDS %.1f TB/s, %.0f TFLOP/s);  the launches it would replace: %.1f us\n",
             c.what, blocks, npieces, npieces * (double)PIECE / 1e6, best * 1e3, (double)blocks * npieces * PIECE / (best * 1e-3) / 1e12,
             flops / (best * 1e-3) / 1e12, c.replaced_us);
    }
  }
  return 0;
}
