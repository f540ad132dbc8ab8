// Probe the lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950: out[p][q] = D_q when only A lane p is 1 and B lane l holds l+1.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(double* out) {
  const int l = threadIdx.x;
  for (int p = 0; p < 64; ++p) {
    const double a = (l == p) ? 1.0 : 0.0;
    const double b = (double)(l + 1);
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[p * 64 + l] = d;
  }
}
int main() {
  double* d; hipMalloc(&d, 64 * 64 * sizeof(double));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  static double h[64 * 64];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int p = 0; p < 64; ++p) {
    printf("A lane %2d ->", p);
    for (int q = 0; q < 64; ++q) if (h[p * 64 + q] != 0.0) printf(" D%d<-B%d", q, (int)h[p * 64 + q] - 1);
    printf("\n");
  }
  return 0;
}
