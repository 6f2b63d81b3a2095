// Does the lane -> address pattern of a 16-byte-per-lane store / load matter on gfx950?  Every wave moves the same
// 16 rows x 64 B patch per instruction (the GEMM epilogue's unit: rows of one M tile, pitch = N * 2 bytes):
//   pattern 0: lane l -> row (l & 15), chunk (l >> 4)      (MFMA accumulator layout after v_permlane16_swap: a
//              quarter-wave touches 16 different rows, 16 B each)
//   pattern 1: lane l -> row (l >> 2), chunk (l & 3)        (4 consecutive lanes cover the 64 contiguous bytes of a row)
// Grid = 256 x 8 waves, each wave walks its own rows; prints chip-wide GB/s for store-only, load-only and load+store.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int PATTERN, int MODE>   // MODE 0 store, 1 load, 2 load + store (residual add)
__global__ __launch_bounds__(512) void k(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int pitch16, int rows_per_wave,
                                         int col_chunks) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = PATTERN == 0 ? (lane & 15) : (lane >> 2);
  const int chunk = PATTERN == 0 ? (lane >> 4) : (lane & 3);
  const size_t wave_row0 = ((size_t)blockIdx.x * 8 + w) * rows_per_wave;
  u32x4 acc = {1u, 2u, 3u, (unsigned)lane};
  for (int r0 = 0; r0 < rows_per_wave; r0 += 16)
    for (int c = 0; c < col_chunks; c += 4) {
      const size_t idx = (wave_row0 + r0 + row) * pitch16 + c + chunk;
      if (MODE >= 1) { const u32x4 v = src[idx]; acc += v; }
      if (MODE != 1) dst[idx] = acc;
    }
  if (MODE == 1 && acc[0] == 0x12345678u) dst[0] = acc;
}

template <int PATTERN, int MODE>
static double run(const u32x4* s, u32x4* d, int pitch16, int rpw, int cc) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<PATTERN, MODE>), dim3(256), dim3(512), 0, 0, s, d, pitch16, rpw, cc);
  hipEventRecord(e0);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<PATTERN, MODE>), dim3(256), dim3(512), 0, 0, s, d, pitch16, rpw, cc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps * 1e-3;
}

int main() {
  // 65536 rows x 320 channels fp16 (the level-0 activation): pitch 640 B = 40 chunks of 16 B; 256 blocks x 8 waves x 32 rows
  const int rows = 65536, pitch16 = 40, rpw = rows / (256 * 8);
  const size_t bytes = (size_t)rows * pitch16 * 16;
  u32x4 *s, *d;
  hipMalloc(&s, bytes); hipMalloc(&d, bytes);
  hipMemset(s, 1, bytes); hipMemset(d, 0, bytes);
  const char* mode[] = {"store only", "load only", "load + store"};
  double t;
  t = run<0, 0>(s, d, pitch16, rpw, pitch16); printf("pattern 0 (row = lane & 15)  %-13s %7.1f us  %6.0f GB/s\n", mode[0], t * 1e6, bytes / t / 1e9);
  t = run<1, 0>(s, d, pitch16, rpw, pitch16); printf("pattern 1 (row = lane >> 2)  %-13s %7.1f us  %6.0f GB/s\n", mode[0], t * 1e6, bytes / t / 1e9);
  t = run<0, 1>(s, d, pitch16, rpw, pitch16); printf("pattern 0 (row = lane & 15)  %-13s %7.1f us  %6.0f GB/s\n", mode[1], t * 1e6, bytes / t / 1e9);
  t = run<1, 1>(s, d, pitch16, rpw, pitch16); printf("pattern 1 (row = lane >> 2)  %-13s %7.1f us  %6.0f GB/s\n", mode[1], t * 1e6, bytes / t / 1e9);
  t = run<0, 2>(s, d, pitch16, rpw, pitch16); printf("pattern 0 (row = lane & 15)  %-13s %7.1f us  %6.0f GB/s\n", mode[2], t * 1e6, 2 * bytes / t / 1e9);
  t = run<1, 2>(s, d, pitch16, rpw, pitch16); printf("pattern 1 (row = lane >> 2)  %-13s %7.1f us  %6.0f GB/s\n", mode[2], t * 1e6, 2 * bytes / t / 1e9);
  return 0;
}
