// How much of the FP64-MFMA issue ceiling survives when every MFMA's B operand is freshly loaded?  Mimics the Ritz
// back-transform's k-step: NB operand loads (8 bytes per lane, rows of a small L2-resident matrix S) feeding NB * NA
// MFMAs (NA = row tiles sharing each B).  No A stream, no stores: pure issue + operand fetch.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_f64_operands.hip -o mfma_f64_operands
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NA, int NB, int SRC>  // SRC 0: global (L1/L2), 1: LDS
__global__ __launch_bounds__(256) void k(const double* __restrict__ S, int ldb, int nsteps, int iters, double* out) {
  extern __shared__ double sS[];
  const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
  if (SRC == 1) {
    for (int i = threadIdx.x; i < 16 * ldb; i += 256) sS[i] = S[i];
    __syncthreads();
  }
  double4_t acc[NA][NB];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};
  double av[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) av[a] = 1.0 + a + lane * 1e-9;
  for (int it = 0; it < iters; ++it) {
    double bcur[NB];
    {
      const double* sr = (SRC == 1 ? sS + (lk) * ldb : S + (int64_t)lk * ldb);
#pragma unroll
      for (int b = 0; b < NB; ++b) bcur[b] = sr[16 * b + lr];
    }
    for (int st = 0; st < nsteps; ++st) {
      double bnxt[NB];
      const int nx = st + 1 < nsteps ? st + 1 : 0;
      const double* sr = (SRC == 1 ? sS + ((4 * nx + lk) % 16) * ldb : S + (int64_t)(4 * nx + lk) * ldb);
#pragma unroll
      for (int b = 0; b < NB; ++b) bnxt[b] = sr[16 * b + lr];
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int a = 0; a < NA; ++a) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bcur[b], acc[a][b], 0, 0, 0);
#pragma unroll
      for (int b = 0; b < NB; ++b) bcur[b] = bnxt[b];
    }
  }
  double s = 0.0;
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) s += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
  if (s == 1.2345e300) out[0] = s;
}

template <int NA, int NB, int SRC>
void run(const double* S, double* out, int ldb, int nsteps, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const size_t lds = SRC == 1 ? (size_t)16 * ldb * 8 : 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NA, NB, SRC>), dim3(256), dim3(256), lds, 0, S, ldb, nsteps, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep == 2) {
      const double nmfma = 256.0 * 4 * (double)iters * nsteps * NA * NB;
      printf("{\"row_tiles\": %d, \"b_operands\": %d, \"source\": \"%s\", \"loads_per_mfma\": %.2f, \"ms\": %.3f, \"tflops\": %.2f}\n", NA, NB,
             SRC ? "lds" : "global(L1/L2)", 1.0 / NA, ms, nmfma * 2048.0 / (ms * 1e-3) / 1e12);
    }
  }
}

int main() {
  const int ldb = 208, nsteps = 50, iters = 40;
  double *S, *out;
  hipMalloc(&S, (size_t)ldb * 208 * 8);
  hipMemset(S, 0, (size_t)ldb * 208 * 8);
  hipMalloc(&out, 8);
  run<1, 13, 0>(S, out, ldb, nsteps, iters);
  run<2, 13, 0>(S, out, ldb, nsteps, iters);
  run<3, 13, 0>(S, out, ldb, nsteps, iters);
  run<1, 13, 1>(S, out, ldb, nsteps, iters);
  run<2, 13, 1>(S, out, ldb, nsteps, iters);
  run<3, 13, 1>(S, out, ldb, nsteps, iters);
  run<2, 7, 0>(S, out, ldb, nsteps, iters);
  run<4, 7, 0>(S, out, ldb, nsteps, iters);
  return 0;
}
