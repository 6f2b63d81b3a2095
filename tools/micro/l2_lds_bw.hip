// Micro-benchmark: achievable global(L2)->LDS bandwidth of 16-byte LDS-DMA loads (the GEMM's operand path) on MI355X.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/l2_lds_bw tools/micro/l2_lds_bw.hip && /tmp/l2_lds_bw
// Every block streams a `span`-byte window (shared by all blocks of the same XCD => L2 hits after the first pass) into
// LDS with `DEPTH` wave-level loads in flight, no MFMA, no ds_read.  Reports TB/s for several windows / occupancies.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int DEPTH>
__global__ __launch_bounds__(512) void stream_kernel(const char* __restrict__ base, size_t span, int iters, int per_xcd) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int xcd = blockIdx.x & 7;
  const char* win = per_xcd ? base + (size_t)xcd * span : base;
  // each wave instruction moves 64 lanes x 16 B = 1 KB: 8 rows of 128 B (like a GEMM operand tile)
  size_t off = ((size_t)(blockIdx.x >> 3) * 8 + w) * 1024 + lane * 16;
  const size_t stride = (size_t)(gridDim.x >> 3) * 8 * 1024;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      size_t o = off % span;
      __builtin_amdgcn_global_load_lds((gptr_t)(win + o), (lptr_t)(smem + (d * 8 + w) * 1024), 16, 0, 0);
      off += stride;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

int main() {
  const size_t total = 512ull << 20;
  char* buf;
  hipMalloc(&buf, total);
  hipMemset(buf, 1, total);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 400;
  struct Cfg { size_t span; int per_xcd; int blocks; const char* what; };
  std::vector<Cfg> cfgs = {
      {2u << 20, 1, 256, "2 MB window per XCD (L2 hit), 1 block/CU"},
      {2u << 20, 1, 512, "2 MB window per XCD (L2 hit), 2 blocks/CU"},
      {1u << 20, 1, 512, "1 MB window per XCD (L2 hit), 2 blocks/CU"},
      {16u << 20, 0, 512, "16 MB shared window (MALL / L2 mix), 2 blocks/CU"},
      {64u << 20, 0, 512, "64 MB shared window (MALL), 2 blocks/CU"},
      {512u << 20, 0, 512, "512 MB (HBM stream), 2 blocks/CU"},
  };
  for (auto& c : cfgs) {
    for (int depth : {2, 4, 8}) {
      auto launch = [&]() {
        const size_t lds = (size_t)depth * 8 * 1024;
        if (depth == 2) hipLaunchKernelGGL(stream_kernel<2>, dim3(c.blocks), dim3(512), lds, 0, buf, c.span, iters, c.per_xcd);
        if (depth == 4) hipLaunchKernelGGL(stream_kernel<4>, dim3(c.blocks), dim3(512), lds, 0, buf, c.span, iters, c.per_xcd);
        if (depth == 8) hipLaunchKernelGGL(stream_kernel<8>, dim3(c.blocks), dim3(512), lds, 0, buf, c.span, iters, c.per_xcd);
      };
      launch();
      hipDeviceSynchronize();
      hipEventRecord(e0);
      launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double bytes = (double)c.blocks * 8 * 1024.0 * depth * iters;
      printf("%-52s depth %d: %7.2f TB/s\n", c.what, depth, bytes / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
