// Pure-issue ceiling of v_mfma_f64_16x16x4_f64 on this part: no memory traffic at all, NACC independent accumulators per
// wave (the Ritz back-transform keeps 13 or 26 in flight), W waves per SIMD.  Prints TFLOP/s and the implied cycles per
// MFMA at the nominal 2.4 GHz.  Build: hipcc -O3 --offload-arch=gfx950 mfma_f64_peak.hip -o mfma_f64_peak
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k_mfma(double* out, int iters, double a0, double b0) {
  double4_t acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (double4_t){0.0, 0.0, 0.0, 0.0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 1.2345e300) out[0] = s;
}

template <int NACC>
void run(int waves_per_simd, int iters) {
  double* out;
  hipMalloc(&out, 8);
  const int threads = 64 * 4 * waves_per_simd;  // one block per CU
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mfma<NACC>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0, 0.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep == 2) {
      const double nmfma = 256.0 * 4 * waves_per_simd * (double)iters * NACC;
      const double tf = nmfma * 2048.0 / (ms * 1e-3) / 1e12;
      printf("{\"nacc\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"tflops\": %.2f, \"ns_per_mfma_per_simd\": %.2f, \"cycles_at_2.4GHz\": %.1f}\n", NACC,
             waves_per_simd, ms, tf, ms * 1e6 / ((double)iters * NACC * waves_per_simd), ms * 1e6 / ((double)iters * NACC * waves_per_simd) * 2.4);
    }
  }
  hipFree(out);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  for (int w : {1, 2, 4}) {
    run<4>(w, iters);
    run<13>(w, iters);
    run<26>(w, iters / 2);
  }
  return 0;
}
