// Stand-alone consumer of the C ABI (no Python, no torch): what a binding in another host language would do.
//   hipcc --offload-arch=gfx950 -I include tests/cabi/cabi_smoke.cpp -L leftrefill_amd/lib -lleftrefill_hip -o cabi_smoke
// Runs a 1x1 "conv" (GEMM) with an identity weight + bias + residual and a LayerNorm on caller-owned device buffers, on a
// caller-owned stream, and checks the results on the host.  Exit code 0 = ok.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "leftrefill_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 2; } } while (0)

int main() {
  if (lr_abi_version() < 15) { printf("unexpected ABI version %d\n", lr_abi_version()); return 1; }
  const int M = 300, C = 128;
  std::vector<__half> hx(M * C), hw(C * C), hr(M * C), hy(M * C), hz(M * C);
  std::vector<float> hb(C), hg(C, 1.0f), hbeta(C, 0.0f);
  for (int i = 0; i < M * C; ++i) { hx[i] = __float2half((float)((i * 37) % 101) / 50.0f - 1.0f); hr[i] = __float2half(0.25f); }
  for (int n = 0; n < C; ++n) { hb[n] = 0.5f; for (int k = 0; k < C; ++k) hw[n * C + k] = __float2half(n == k ? 2.0f : 0.0f); }
  __half *dx, *dw, *dr, *dy, *dz;
  float *db, *dg, *dbeta;
  CK(hipMalloc(&dx, M * C * 2)); CK(hipMalloc(&dw, C * C * 2)); CK(hipMalloc(&dr, M * C * 2));
  CK(hipMalloc(&dy, M * C * 2)); CK(hipMalloc(&dz, M * C * 2));
  CK(hipMalloc(&db, C * 4)); CK(hipMalloc(&dg, C * 4)); CK(hipMalloc(&dbeta, C * 4));
  CK(hipMemcpy(dx, hx.data(), M * C * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), C * C * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dr, hr.data(), M * C * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), C * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dg, hg.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dbeta, hbeta.data(), C * 4, hipMemcpyHostToDevice));
  hipStream_t st;
  CK(hipStreamCreate(&st));

  lr_gemm_args a = {};
  a.p1 = (const lr_half*)dx; a.C1 = C; a.B = 1; a.H = 1; a.W = M; a.Hs = 1; a.Ws = M; a.taps = 1; a.stride = 1;
  a.wt = (const lr_half*)dw; a.N = C; a.bias = db; a.resid = (const lr_half*)dr; a.ld_resid = C; a.out = (lr_half*)dy; a.ld_out = C;
  int rc = lr_gemm_conv_f16(&a, st);
  if (rc) { printf("lr_gemm_conv_f16 rc=%d\n", rc); return 1; }
  rc = lr_layernorm((const lr_half*)dy, dg, dbeta, 1e-5f, (lr_half*)dz, M, C, st);
  if (rc) { printf("lr_layernorm rc=%d\n", rc); return 1; }
  a.taps = 4;
  if (lr_gemm_conv_f16(&a, st) != LR_E_UNSUPPORTED) { printf("bad taps not rejected\n"); return 1; }
  CK(hipStreamSynchronize(st));
  CK(hipMemcpy(hy.data(), dy, M * C * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hz.data(), dz, M * C * 2, hipMemcpyDeviceToHost));
  double worst = 0.0, worst_ln = 0.0;
  for (int m = 0; m < M; ++m) {
    double mean = 0.0, var = 0.0;
    for (int c = 0; c < C; ++c) {
      const double ref = 2.0 * __half2float(hx[m * C + c]) + 0.5 + 0.25;
      worst = fmax(worst, fabs(ref - __half2float(hy[m * C + c])));
      mean += __half2float(hy[m * C + c]);
    }
    mean /= C;
    for (int c = 0; c < C; ++c) { const double d = __half2float(hy[m * C + c]) - mean; var += d * d; }
    const double rstd = 1.0 / sqrt(var / C + 1e-5);
    for (int c = 0; c < C; ++c)
      worst_ln = fmax(worst_ln, fabs((__half2float(hy[m * C + c]) - mean) * rstd - __half2float(hz[m * C + c])));
  }
  printf("cabi_smoke: gemm max err %.3e, layernorm max err %.3e\n", worst, worst_ln);
  return (worst < 4e-3 && worst_ln < 4e-3) ? 0 : 1;
}
