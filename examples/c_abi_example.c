/* Plain-C client of liblanczos_hip.so (no Python, no C++): builds a 2-D periodic 5-point Laplacian, runs k Lanczos
 * steps through the C ABI and prints the extreme Ritz-value bounds from the Gershgorin interval of T; then the round-3 entry
 * points: the back-transform with S = I (Y must equal the basis), a row window of Y (resident and forced-chunked), and a
 * checkpoint (basis + residual + coefficients) resumed to 2 k steps, which must reproduce an uninterrupted 2 k-step run bit for bit.
 *
 *   gcc -std=c99 -Iinclude examples/c_abi_example.c -Llanczos_amd -llanczos_hip -Wl,-rpath,$PWD/lanczos_amd -lm -o c_abi_example
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "lanczos_hip.h"

#define CHECK(call)                                                              \
  do {                                                                           \
    int st_ = (call);                                                            \
    if (st_ != LZ_OK) {                                                          \
      fprintf(stderr, "%s failed (%d): %s\n", #call, st_, lz_last_error(h));     \
      return 2;                                                                  \
    }                                                                            \
  } while (0)

int main(int argc, char** argv) {
  const int nx = argc > 1 ? atoi(argv[1]) : 64, ny = argc > 2 ? atoi(argv[2]) : 48, k = argc > 3 ? atoi(argv[3]) : 24;
  const int64_t M = (int64_t)nx * ny, nnz = 5 * M;
  int32_t* rowptr = malloc((M + 1) * sizeof *rowptr);
  int32_t* colidx = malloc(nnz * sizeof *colidx);
  double* vals = malloc(nnz * sizeof *vals);
  double* v0 = malloc(M * sizeof *v0);
  double* alpha = malloc(k * sizeof *alpha);
  double* beta = malloc(k * sizeof *beta);
  lz_handle h = NULL;
  if (lz_create(&h, 0) != LZ_OK) {
    fprintf(stderr, "lz_create: %s\n", lz_last_error(NULL));
    return 3; /* no GPU: the library never falls back to the CPU */
  }
  for (int64_t r = 0; r < M; ++r) {
    const int x = (int)(r % nx), y = (int)(r / nx);
    int64_t c[5] = {r, r - x + (x + 1) % nx, r - x + (x + nx - 1) % nx, (int64_t)x + (int64_t)((y + 1) % ny) * nx,
                    (int64_t)x + (int64_t)((y + ny - 1) % ny) * nx};
    for (int i = 1; i < 5; ++i) /* insertion sort: CSR columns ascending */
      for (int j = i; j > 0 && c[j - 1] > c[j]; --j) {
        int64_t t = c[j];
        c[j] = c[j - 1];
        c[j - 1] = t;
      }
    rowptr[r] = (int32_t)(5 * r);
    for (int i = 0; i < 5; ++i) {
      colidx[5 * r + i] = (int32_t)c[i];
      vals[5 * r + i] = c[i] == r ? 4.0 : -1.0;
    }
  }
  rowptr[M] = (int32_t)nnz;
  double nrm = 0.0;
  for (int64_t i = 0; i < M; ++i) {
    v0[i] = sin(0.37 * (double)i) + 0.5;
    nrm += v0[i] * v0[i];
  }
  nrm = sqrt(nrm);
  for (int64_t i = 0; i < M; ++i) v0[i] /= nrm;
  CHECK(lz_set_csr(h, M, 0, M, M, nnz, rowptr, colidx, vals));
  CHECK(lz_run(h, k, v0, alpha, beta));
  double lo = 1e300, hi = -1e300;
  for (int j = 0; j < k; ++j) {
    const double off = (j > 0 ? fabs(beta[j - 1]) : 0.0) + (j + 1 < k ? fabs(beta[j]) : 0.0);
    if (alpha[j] - off < lo) lo = alpha[j] - off;
    if (alpha[j] + off > hi) hi = alpha[j] + off;
  }
  int sweeps = 0;
  CHECK(lz_last_sweeps(h, &sweeps));
  printf("M=%lld k=%d alpha[0]=%.15g beta[0]=%.15g Gershgorin(T)=[%.6f, %.6f] sweeps=%d\n", (long long)M, k, alpha[0], beta[0], lo, hi,
         sweeps);
  /* ---- Y = V S with S = I: the Ritz vectors are the basis vectors; fetch a window of rows, resident and chunked ---- */
  double* S = calloc((size_t)k * k, sizeof *S);
  double* V = malloc((size_t)k * M * sizeof *V);
  double* Yw = malloc((size_t)40 * k * sizeof *Yw);
  for (int j = 0; j < k; ++j) S[(size_t)j * k + j] = 1.0;
  CHECK(lz_get_basis(h, V, M));
  int bad = 0;
  for (int pass = 0; pass < 2; ++pass) {
    CHECK(lz_set_tuning(h, 16, pass ? 512 : 0)); /* pass 1: force the chunked mode (what a 160 GB basis gets on one GPU) */
    CHECK(lz_ritz_vectors(h, S, NULL));
    int64_t chunk = -1;
    CHECK(lz_ritz_info(h, &chunk, NULL));
    CHECK(lz_get_ritz_rows(h, M / 2 - 7, 40, Yw));
    for (int r = 0; r < 40; ++r)
      for (int j = 0; j < k; ++j) bad += Yw[(size_t)r * k + j] != V[(size_t)j * M + (M / 2 - 7 + r)];
    printf("ritz rows (%s, chunk_rows=%lld): %d mismatches\n", pass ? "chunked" : "resident", (long long)chunk, bad);
    if ((pass == 1) != (chunk > 0)) bad += 1;
  }
  CHECK(lz_set_tuning(h, 16, 0));
  /* ---- checkpoint after k steps, resume to 2 k, compare with an uninterrupted 2 k-step run ---- */
  const int k2 = 2 * k;
  double* r = malloc(M * sizeof *r);
  double* a2 = malloc(k2 * sizeof *a2);
  double* b2 = malloc(k2 * sizeof *b2);
  double* a3 = malloc(k2 * sizeof *a3);
  double* b3 = malloc(k2 * sizeof *b3);
  CHECK(lz_get_residual(h, r));
  CHECK(lz_run_resume(h, k2, k, V, M, r, alpha, beta, a2, b2));
  CHECK(lz_run(h, k2, v0, a3, b3));
  for (int j = 0; j < k2; ++j) bad += a2[j] != a3[j];
  for (int j = 0; j + 1 < k2; ++j) bad += b2[j] != b3[j];
  printf("resume %d -> %d steps vs one run of %d: %d mismatches\n", k, k2, k2, bad);
  lz_destroy(h);
  free(rowptr), free(colidx), free(vals), free(v0), free(alpha), free(beta);
  free(S), free(V), free(Yw), free(r), free(a2), free(b2), free(a3), free(b3);
  return (isfinite(lo) && isfinite(hi) && sweeps == k && bad == 0) ? 0 : 1;
}
