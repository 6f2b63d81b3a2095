// Probe of v_permlane16_swap / v_permlane32_swap lane semantics (prints which (operand, lane) each output lane holds).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* o) {
  const unsigned l = threadIdx.x;
  unsigned a = 0x100 + l, b = 0x200 + l;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  o[l] = r[0]; o[64 + l] = r[1];
  auto s = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[128 + l] = s[0]; o[192 + l] = s[1];
}
int main() {
  unsigned* d; unsigned h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"p16 r0", "p16 r1", "p32 r0", "p32 r1"};
  for (int v = 0; v < 4; ++v) {
    printf("%s:", names[v]);
    for (int row = 0; row < 4; ++row) printf("  lanes %2d-%2d <- %c[%2d..]", row * 16, row * 16 + 15, (h[v * 64 + row * 16] >> 8) == 1 ? 'a' : 'b', h[v * 64 + row * 16] & 0xff);
    printf("\n");
  }
  return 0;
}
