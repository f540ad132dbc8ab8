// Two-sided (bi-orthogonal) Lanczos of the Irregular copy (IrrLanczos.py:77-187, 390-443): host side of the C ABI - the
// Gram-Schmidt link chains, lz_run_two_sided, the bireorthogonalize step API.  Kernels: lz_twosided.hip.
#include "lz_context.h"

using namespace lz;
using namespace lz::api;


namespace {

double* bi_base(lz_handle h, int which) {  // 0 = Q (the published basis), 1 = P, 2 = Qb, 3 = Pb
  return which == 0 ? h->d_V : h->d_B3 + (size_t)(which - 1) * (size_t)h->n * (size_t)h->ldv;
}
double* bi_row(lz_handle h, int which, int j) { return bi_base(h, which) + (int64_t)j * h->ldv; }
// d_bi[7] doubles as the ticket counter of the single-launch A/B arm (tune[11] == 2: the last block folds the partials
// behind a __threadfence()).  Measured (tools/two_sided_probe.py, profiles/r01/ab_two_sided_links.json): the agent-scope
// release has to write back the L2 lines the kernel just dirtied, which costs far more than the launch it saves - 157
// vs 41 ms at M = 2.6e5, 250 vs 98 ms at M = 1e6, 178 vs 75 ms at M = 1e7, a tie at M = 9e4.  Default: two launches.  (A third arm, the fold deferred into the
// consumer's prologue, tune[11] == 3, is no faster either: see bi_reorth.)
#ifdef LZ_KBENCH
unsigned* bi_ticket(lz_handle h) { return h->tune[11] == 2 ? reinterpret_cast<unsigned*>(h->d_bi + 7) : nullptr; }
bool bi_defer(lz_handle h) { return h->tune[11] == 3; }
#else  // both arms are retired from the product library (lz_set_tuning refuses knob 11 >= 2)
unsigned* bi_ticket(lz_handle) { return nullptr; }
bool bi_defer(lz_handle) { return false; }
#endif

int bi_alloc(lz_handle h, int n, int zero_rows) {
  if (h->kind != 1) return fail(h, LZ_ERR_STATE, "two-sided Lanczos needs a CSR matrix (lz_set_csr)");
  if (h->world > 1) return fail(h, LZ_ERR_STATE, "two-sided Lanczos is single-rank only");
  if (h->rows != h->ncols_ext || h->rows != h->Mg) return fail(h, LZ_ERR_STATE, "two-sided Lanczos needs the whole square matrix on this rank");
  LZ_TRY(basis_alloc(h, n, zero_rows));
  const size_t one = (size_t)n * (size_t)h->ldv;
  if (!h->d_B3 || h->bi_n != n) {
    LZ_TRY(dev_alloc(h, h->d_B3, 3 * one));
    LZ_TRY(dev_alloc(h, h->d_s, (size_t)h->ldv));
    LZ_TRY(dev_alloc(h, h->d_gamma, (size_t)n + 1));
    LZ_TRY(dev_alloc(h, h->d_bi, 8));
    h->bi_n = n;
  }
  LZ_TRY(ensure_part(h, (size_t)2 * bi_partials_needed()));  // two partial buffers: a link whose fold is deferred leaves its partials for its consumer
  const size_t zr = (size_t)std::min(zero_rows, n) * (size_t)h->ldv * sizeof(double);
  for (int w = 1; w < 4; ++w) LZ_HIP(h, hipMemsetAsync(bi_base(h, w), 0, zr, h->stream));
  LZ_HIP(h, hipMemsetAsync(h->d_s, 0, (size_t)h->ldv * sizeof(double), h->stream));
  LZ_HIP(h, hipMemsetAsync(h->d_gamma, 0, ((size_t)n + 1) * sizeof(double), h->stream));
  LZ_HIP(h, hipMemsetAsync(h->d_bi, 0, 8 * sizeof(double), h->stream));
  LZ_HIP(h, hipStreamSynchronize(h->stream));
  return LZ_OK;
}

// IrrLanczos.py:408-441 on row jj (>= 1).  from_rs: the pair is formed here as (r / beta, s / gamma) with the factors the
// two-term kernel left in f (driver loop :136-137); otherwise rows jj of Q and P are taken as stored (step API).
int bi_reorth(lz_handle h, int jj, bool from_rs) {
  const int64_t len = h->rows_pad;
  double* S = h->d_bi;
  double* f = h->d_bi + 4;
  unsigned* tk = bi_ticket(h);
  double *q = bi_row(h, 0, jj), *p = bi_row(h, 1, jj), *qb = bi_row(h, 2, jj), *pb = bi_row(h, 3, jj);
  const double M = (double)h->rows;
  Scope sc(h, LZ_K_QTW, (2.0 * jj * 64.0 + 5.0 * 32.0) * M, (2.0 * jj * 12.0) * M);
  // A/B arm (tune[11] == 3): a link's four sums are not folded by a launch of their own - the link leaves its block
  // partials in one of two buffers and the NEXT link (which applies the axpy they decide) folds them in its prologue, in
  // k_bi_final's order: one launch per link instead of two, same bits.  Measured (tests/test_gpu_two_sided.py, device
  // time): 0.81-0.92x - a dependent launch costs ~4 us here and the emulated fold (16 shuffle trees per block) as much,
  // so the separate fold kernel stays the default.
  const bool defer = bi_defer(h) && !tk;
  double* pbuf[2] = {h->d_part, h->d_part + bi_partials_needed()};
  int cur = 0;  // buffer the next link writes its partials to
  const double* pend = nullptr;  // where the previous link's deferred partials are
  auto link = [&](int first, int pnd, int dots, double* x, double* y, const double* xs, const double* ys, const double* ff, const double* ap,
                  const double* bp, const double* a, const double* b, int epi) {
    launch_bi(first, pnd, dots, x, y, xs, ys, ff, ap, bp, S, a, b, len, pbuf[cur], epi, S, f, nullptr, nullptr, tk, h->stream, pnd ? pend : nullptr,
              defer);
    pend = (defer && dots == 0) ? pbuf[cur] : nullptr;
    cur ^= 1;
  };
  if (jj == 0) {
    // j = 0 (the static method's own call shape; the driver starts at j = 1): both projection loops are empty - rescale the pair
    // to q.p = +-1 (:418-420) and seed the two orthonormal bases with it (:423-424, 437-438)
    link(0, 0, 1, q, p, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1);
    link(1, 0, 2, q, p, q, p, f, nullptr, nullptr, nullptr, nullptr, 2);
    link(1, 0, 2, qb, pb, q, p, f, nullptr, nullptr, nullptr, nullptr, 2);
    link(1, 0, 3, qb, pb, qb, pb, f, nullptr, nullptr, nullptr, nullptr, 2);
    return check_launch(h, "bireorthogonalize(j = 0)");
  }
  // project q on the orthonormalised p's and p on the orthonormalised q's, one vector at a time (:409-416)
  for (int i = 0; i < jj; ++i) {
    const double *a = bi_row(h, 3, i), *b = bi_row(h, 2, i);
    if (i == 0)
      link(from_rs ? 1 : 0, 0, 0, q, p, h->d_r, h->d_s, f, nullptr, nullptr, a, b, 0);
    else
      link(0, 1, 0, q, p, nullptr, nullptr, nullptr, bi_row(h, 3, i - 1), bi_row(h, 2, i - 1), a, b, 0);
  }
  // last axpy + q.p  ->  f = {sqrt|q.p|, sqrt|q.p|, sign(q.p)}; rescale so that q.p = +-1 (:418-420) + the two norms
  link(0, 1, 1, q, p, nullptr, nullptr, nullptr, bi_row(h, 3, jj - 1), bi_row(h, 2, jj - 1), nullptr, nullptr, 1);
  link(1, 0, 2, q, p, q, p, f, nullptr, nullptr, nullptr, nullptr, 2);
  // q_basis[jj] = q / |q|, p_basis[jj] = p / |p| (:423-424), made orthogonal to the earlier basis vectors (:427-434)
  for (int i = 0; i < jj; ++i) {
    const double *a = bi_row(h, 2, i), *b = bi_row(h, 3, i);
    if (i == 0)
      link(1, 0, 0, qb, pb, q, p, f, nullptr, nullptr, a, b, 0);
    else
      link(0, 1, 0, qb, pb, nullptr, nullptr, nullptr, bi_row(h, 2, i - 1), bi_row(h, 3, i - 1), a, b, 0);
  }
  link(0, 1, 2, qb, pb, nullptr, nullptr, nullptr, bi_row(h, 2, jj - 1), bi_row(h, 3, jj - 1), nullptr, nullptr, 2);
  link(1, 0, 3, qb, pb, qb, pb, f, nullptr, nullptr, nullptr, nullptr, 2);  // :437-438
  return check_launch(h, "bireorthogonalize");
}

}  // namespace

extern "C" {

int lz_set_csr_transpose(lz_handle h, int64_t nnz, const int32_t* rowptr, const int32_t* colidx, const double* vals) {
  if (!h) return LZ_ERR_ARG;
  if (h->kind != 1) return fail(h, LZ_ERR_STATE, "lz_set_csr_transpose: call lz_set_csr first");
  if (h->world > 1 || h->rows != h->Mg || h->ncols_ext != h->rows)
    return fail(h, LZ_ERR_STATE, "lz_set_csr_transpose: needs the whole square matrix on one rank");
  LZ_HIP(h, hipSetDevice(h->dev));
  LZ_HIP(h, hipStreamSynchronize(h->stream));
  if (!rowptr) {
    h->has_T = false;
    h->T_declared = true;
    return LZ_OK;
  }
  if (nnz < 0 || nnz >= (int64_t)1 << 31 || (nnz > 0 && (!colidx || !vals))) return fail(h, LZ_ERR_ARG, "lz_set_csr_transpose: bad nnz or NULL arrays");
  int fixed_k = 0, max_nnz = 0;
  h->has_T = false;
  h->T_declared = false;
  LZ_TRY(upload_csr(h, h->csrT, "lz_set_csr_transpose", h->rows, h->rows, nnz, rowptr, colidx, vals, &fixed_k, &max_nnz));
  LZ_TRY(fill_csr_meta(h, h->csrT, rowptr, h->rows, h->rows, nnz, fixed_k, max_nnz));
  h->has_T = true;
  h->T_declared = true;
  return LZ_OK;
}

int lz_bi_alloc(lz_handle h, int n) {
  if (!h) return LZ_ERR_ARG;
  if (n < 1) return fail(h, LZ_ERR_ARG, "n must be >= 1");
  LZ_HIP(h, hipSetDevice(h->dev));
  return bi_alloc(h, n, n);
}

static int bi_check_row(lz_handle h, int which, int j) {
  if (!h->d_V || !h->d_B3 || h->bi_n != h->n) return fail(h, LZ_ERR_STATE, "no two-sided bases allocated (lz_bi_alloc / lz_run_two_sided)");
  if (which < 0 || which > 3 || j < 0 || j >= h->n) return fail(h, LZ_ERR_ARG, "basis selector or row index out of range");
  return LZ_OK;
}

int lz_bi_set_row(lz_handle h, int which, int j, const double* row) {
  if (!h || !row) return LZ_ERR_ARG;
  LZ_TRY(bi_check_row(h, which, j));
  LZ_HIP(h, hipSetDevice(h->dev));
  LZ_HIP(h, hipMemcpyAsync(bi_row(h, which, j), row, (size_t)h->rows * sizeof(double), hipMemcpyHostToDevice, h->stream));
  LZ_HIP(h, hipStreamSynchronize(h->stream));
  return LZ_OK;
}

int lz_bi_get_row(lz_handle h, int which, int j, double* row) {
  if (!h || !row) return LZ_ERR_ARG;
  LZ_TRY(bi_check_row(h, which, j));
  LZ_HIP(h, hipSetDevice(h->dev));
  LZ_HIP(h, hipMemcpyAsync(row, bi_row(h, which, j), (size_t)h->rows * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  LZ_HIP(h, hipStreamSynchronize(h->stream));
  return LZ_OK;
}

int lz_step_bireorth(lz_handle h, int j) {
  if (!h) return LZ_ERR_ARG;
  LZ_TRY(bi_check_row(h, 0, j));
  LZ_HIP(h, hipSetDevice(h->dev));
  LZ_TRY(bi_reorth(h, j, false));
  LZ_HIP(h, hipStreamSynchronize(h->stream));
  return LZ_OK;
}

int lz_step_bireorth_mem_safe(lz_handle h, int j) {
  if (!h) return LZ_ERR_ARG;
  LZ_TRY(bi_check_row(h, 0, j));
  LZ_HIP(h, hipSetDevice(h->dev));
  // IrrLanczos.py:399-403 then :405-409: V1[j] against the rows of V2, then V2[j] against the rows of V1 (its row j already
  // updated).  The n coefficients of a half live in d_gamma (n + 1 doubles, idle in the step API).
  launch_bi_mem_safe(bi_row(h, 0, j), bi_base(h, 1), h->ldv, h->n, j, h->rows_pad, h->d_gamma, h->stream);
  launch_bi_mem_safe(bi_row(h, 1, j), bi_base(h, 0), h->ldv, h->n, j, h->rows_pad, h->d_gamma, h->stream);
  LZ_TRY(check_launch(h, "bireorthogonalize (mem_safe)"));
  LZ_HIP(h, hipStreamSynchronize(h->stream));
  return LZ_OK;
}

int lz_run_two_sided(lz_handle h, int n, const double* q0, const double* p0, double* alpha_out, double* beta_out, double* gamma_out) {
  if (!h) return LZ_ERR_ARG;
  if (!q0 || !p0 || !alpha_out || !beta_out || !gamma_out) return fail(h, LZ_ERR_ARG, "lz_run_two_sided: NULL buffer");
  if (n < 2) return fail(h, LZ_ERR_ARG, "lz_run_two_sided: n must be >= 2 (H_eff[0,1] and beta[-1] exist only then)");
  if (n > h->Mg) return fail(h, LZ_ERR_ARG, "lz_run_two_sided: n cannot be larger than M");
  if (h->kind == 1 && !h->T_declared)
    return fail(h, LZ_ERR_STATE, "lz_run_two_sided: call lz_set_csr_transpose first (NULL arrays if H is symmetric)");
  LZ_HIP(h, hipSetDevice(h->dev));
  LZ_TRY(bi_alloc(h, n, 1));
  h->y_n = 0;
  const int64_t len = h->rows_pad;
  const size_t rowb = (size_t)h->rows * sizeof(double);
  double* S = h->d_bi;
  double* f = h->d_bi + 4;
  unsigned* tk = bi_ticket(h);
  LZ_HIP(h, hipMemcpyAsync(bi_row(h, 0, 0), q0, rowb, hipMemcpyHostToDevice, h->stream));
  LZ_HIP(h, hipMemcpyAsync(bi_row(h, 1, 0), p0, rowb, hipMemcpyHostToDevice, h->stream));
  LZ_HIP(h, hipEventRecord(h->run_a, h->stream));
  // q_basis[0] = q0 / |q0|, p_basis[0] = p0 / |p0|  (IrrLanczos.py:113-118)
  launch_bi(0, 0, 2, bi_row(h, 0, 0), bi_row(h, 1, 0), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, len, h->d_part, 2, S,
            f, nullptr, nullptr, tk, h->stream);
  launch_bi(1, 0, 3, bi_row(h, 2, 0), bi_row(h, 3, 0), bi_row(h, 0, 0), bi_row(h, 1, 0), f, nullptr, nullptr, nullptr, nullptr, nullptr, len,
            h->d_part, 2, S, f, nullptr, nullptr, tk, h->stream);
  LZ_TRY(check_launch(h, "two-sided start"));
  const CsrDev& AT = h->has_T ? h->csrT : h->csr;
  const double M = (double)h->rows;
  for (int j = 0; j + 1 < n; ++j) {
    double *qj = bi_row(h, 0, j), *pj = bi_row(h, 1, j);
    {
      Scope sc(h, LZ_K_SPMV, 2.0 * spmv_bytes(h), 2.0 * spmv_flops(h));
      launch_spmv_csr(h->csr, qj, h->d_r, qj, h->d_part, h->flags, h->stream);  // r = H q_j   (:124)
      launch_spmv_csr(AT, pj, h->d_s, pj, h->d_part, h->flags, h->stream);      // s = HT p_j  (:125)
      LZ_TRY(check_launch(h, "two-sided spmv"));
    }
    {
      Scope sc(h, LZ_K_THREE, (j > 0 ? 112.0 : 64.0) * M, 12.0 * M);
      // r -= gamma[j-1] q[j-1]; s -= beta[j-1] p[j-1] (at j = 0 both are the reference's zero rows: skipped);
      // alpha[j] = (p_j . r + q_j . s) / 2  (:128-132)
      if (j > 0)
        launch_bi_two_term(1, 0, h->d_r, h->d_s, bi_row(h, 0, j - 1), bi_row(h, 1, j - 1), h->d_gamma + (j - 1), h->d_beta + (j - 1), pj, qj, len,
                           h->d_part, f, h->d_alpha + j, nullptr, tk, h->stream);
      else
        launch_bi_two_term(0, 0, h->d_r, h->d_s, nullptr, nullptr, nullptr, nullptr, pj, qj, len, h->d_part, f, h->d_alpha + j, nullptr, tk, h->stream);
      // r -= alpha q_j; s -= alpha p_j; w = r . s; beta[j] = sqrt|w|; gamma[j] = w / beta[j]  (:134-141)
      launch_bi_two_term(1, 1, h->d_r, h->d_s, qj, pj, h->d_alpha + j, h->d_alpha + j, nullptr, nullptr, len, h->d_part, f, h->d_beta + j,
                         h->d_gamma + j, tk, h->stream);
      LZ_TRY(check_launch(h, "two-sided recurrence"));
    }
    LZ_TRY(bi_reorth(h, j + 1, true));  // q[j+1] = r / beta, p[j+1] = s / gamma, then bireorthogonalize (:142-161)
  }
  // alpha[n-1] = q[n-1] . r with the LAST iteration's residual (:163)
  launch_bi_two_term(0, 2, h->d_r, nullptr, nullptr, nullptr, nullptr, nullptr, bi_row(h, 0, n - 1), nullptr, len, h->d_part, f, h->d_alpha + (n - 1),
                     nullptr, tk, h->stream);
  LZ_TRY(check_launch(h, "two-sided last alpha"));
  LZ_HIP(h, hipEventRecord(h->run_b, h->stream));
  LZ_HIP(h, hipMemcpyAsync(alpha_out, h->d_alpha, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  LZ_HIP(h, hipMemcpyAsync(beta_out, h->d_beta, (size_t)(n - 1) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  LZ_HIP(h, hipMemcpyAsync(gamma_out, h->d_gamma, (size_t)(n - 1) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  LZ_HIP(h, hipStreamSynchronize(h->stream));
  float ms = 0.f;
  LZ_HIP(h, hipEventElapsedTime(&ms, h->run_a, h->run_b));
  h->acc.total_ms += ms;
  h->last_sweeps = n - 1;
  return LZ_OK;
}

}  // extern "C"
