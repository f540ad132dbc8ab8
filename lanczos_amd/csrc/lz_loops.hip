// The Krylov loops of lz_run: the individual steps, the six- / five- / three-launch loops, the one-reduce loop, the partial
// re-orthogonalisation loops (host-decided and device-resident), loop selection, lz_run / lz_run_resume / lz_get_residual / lz_reserve.
#include "lz_context.h"

using namespace lz;
using namespace lz::api;

namespace lz {
namespace api {

// ---- the individual steps (device-resident scalars, no host sync) --------
int ensure_part(lz_handle h, size_t need) {
  if (need <= h->part_cap) return LZ_OK;
  LZ_TRY(dev_alloc(h, h->d_part, need));
  h->part_cap = need;
  return LZ_OK;
}

double spmv_bytes(lz_handle h, bool ell) {
  // (a row-class coded ELL copy: one class byte per row, the values only when they are not part of the class, x once, y)
  if (h->kind == 1 && h->csr.ell_coded && (ell || (h->csr.ell_default && ell_usable(h->csr, h->flags))))
    return 17.0 * h->rows + (h->csr.ell_coded == 1 ? 8.0 * h->csr.nnz : 0.0) + (h->csr.ell_coded == 3 ? 8.0 * h->rows : 0.0);  // (3: + the diagonal)
  if (h->kind == 1) return 12.0 * h->csr.nnz + 4.0 * (h->rows + 1) + 16.0 * h->rows;
  return 8.0 * (double)h->rows * (double)h->Mg + 8.0 * (double)h->Mg + 8.0 * h->rows;  // A block, x once, y
}
double spmv_flops(lz_handle h) { return h->kind == 1 ? 2.0 * h->csr.nnz : 2.0 * (double)h->rows * (double)h->Mg; }

// r = A V[j]; alpha_dst[0] = V[j] . r, summed over ranks unless reduce == false (one-reduce mode: the partial sum rides
// in the next all-reduce)
int step_spmv(lz_handle h, int j, double* alpha_dst, bool reduce, int* np_out) {
  if (!alpha_dst) alpha_dst = h->d_alpha + j;
  const double* x = nullptr;
  if (h->halo_inflight_j == j) {  // exchange already issued on the comm stream behind the boundary update
    LZ_HIP(h, hipStreamWaitEvent(h->stream, h->e_halo, 0));
    x = h->d_V + (int64_t)j * h->ldv;
    h->halo_inflight_j = -1;
  } else {
    LZ_TRY(comm_exchange_x(h, j, &x));
  }
  const double* xown = h->d_V + (int64_t)j * h->ldv;
  int np = 0;
  {
    Scope sc(h, LZ_K_SPMV, spmv_bytes(h), spmv_flops(h));
    if (h->kind == 1)
      np = launch_spmv_csr(h->csr, x, h->d_r, xown, h->d_part, h->flags, h->stream);
    else
      np = launch_gemv_dense(h->d_dense, h->rows, h->ncols_ext, h->dense_lda, x, xown, h->d_r, h->d_part, h->stream);
    LZ_TRY(check_launch(h, "spmv"));
  }
  if (np_out) {  // fused small-problem mode: the consumer kernel adds the block partials itself
    *np_out = np;
    return LZ_OK;
  }
  {
    Scope sc(h, LZ_K_FINAL, 0, 0);
    launch_final_sum(h->d_part, np, alpha_dst, h->stream);
    LZ_TRY(check_launch(h, "final_sum(alpha)"));
  }
  return reduce ? comm_allreduce(h, alpha_dst, 1) : LZ_OK;
}

// V[j] = r / sqrt(nrm2) (if scale), then c = V[0:nrows] . V[j]; V[j] = 2 V[j] - c^T V[0:nrows]
int step_reorth(lz_handle h, int j, int nrows, bool scale, int beta_idx, bool in_run_loop) {
  const double M = (double)h->rows;
  // fused-norm mode (the Python layers default to it): the reduced sums are [V_i . r (i < j), r . r] - one all-reduce at N > 1 -; beta and the scaling by
  // 1/beta are applied afterwards.  Only valid for the in-loop call shape (row j is the newest row).
  const bool fused = scale && (h->flags & LZ_FLAG_FUSED_NORM) && !(h->flags & LZ_FLAG_REORTH_PARTIAL) && nrows == j + 1;
  h->qplan.variant = h->tune[1];  // A/B knob may change between launches on one handle (same allocation for every arm)
  {
    Scope sc(h, LZ_K_QTW, 8.0 * (nrows - 1) * M + (scale && !fused ? 16.0 : 8.0) * M, 2.0 * nrows * M);
    LZ_HIP(h, launch_qtw(h->d_V, h->ldv, h->rows_pad, nrows, j, scale ? h->d_r : nullptr, h->d_nrm2, h->d_beta + beta_idx, h->qplan,
                         h->d_part, fused ? 2 : (scale ? 1 : 0), h->stream));
    LZ_TRY(check_launch(h, "qtw"));
  }
  {
    Scope sc(h, LZ_K_FINAL, 0, 0);
    launch_final_rows(h->d_part, nrows, h->qplan.P, h->d_c, h->stream, h->qplan.family == 2);
    LZ_TRY(check_launch(h, "final_rows"));
  }
  LZ_TRY(comm_allreduce(h, h->d_c, nrows));
  const bool overlap = in_run_loop && (h->flags & LZ_FLAG_OVERLAP_HALO) && h->xmode == 1 && h->all_contig && h->comm_kind == 1 &&
                       !h->peers.empty() && (h->world > 1 || h->tune[6]);
  // inside lz_run the default (slice-owner) update kernel turns the reduced sums into beta and the coefficients itself
  const bool raw_c = fused && in_run_loop && (h->tune[8] == 0 || h->tune[8] >= 3);
  if (fused && !raw_c) {
    Scope sc(h, LZ_K_FINAL, 0, 0);
    launch_fused_prepare(h->d_c, j, h->d_beta + beta_idx, h->stream);
    LZ_TRY(check_launch(h, "fused_prepare"));
  }
  if (!overlap) {
    Scope sc(h, LZ_K_UPDATE, 8.0 * (nrows - 1) * M + 16.0 * M, 2.0 * nrows * M);
    launch_update(h->d_V, h->ldv, h->rows_pad, nrows, j, h->d_c, fused ? h->d_r : nullptr, h->d_beta + beta_idx, h->tune[8], h->stream, 0, -1,
                  raw_c ? 1 : 0);
    LZ_TRY(check_launch(h, "update"));
    return LZ_OK;
  }
  // 1. boundary positions (the faces the neighbours need), 2. their exchange on the comm stream, 3. interior
  double* vj = h->d_V + (int64_t)j * h->ldv;
  {
    Scope sc(h, LZ_K_UPDATE, 8.0 * (nrows - 1) * M + 16.0 * M, 2.0 * nrows * M);
    // both faces leave in ONE launch of the small-range kernel (the face kernel is a latency chain over the rows)
    const bool small = h->tune[8] == 0;
    for (size_t q = 0; q < h->bnd_ranges.size(); q += 2) {
      const auto& ra = h->bnd_ranges[q];
      const bool pair = small && q + 1 < h->bnd_ranges.size() && ra.second - ra.first <= 16384 &&
                        h->bnd_ranges[q + 1].second - h->bnd_ranges[q + 1].first <= 16384;
      launch_update(h->d_V, h->ldv, h->rows_pad, nrows, j, h->d_c, fused ? h->d_r : nullptr, h->d_beta + beta_idx, h->tune[8], h->stream,
                    ra.first, ra.second, raw_c ? 1 : 0, pair ? h->bnd_ranges[q + 1].first : 0, pair ? h->bnd_ranges[q + 1].second : 0);
      if (!pair && q + 1 < h->bnd_ranges.size())
        launch_update(h->d_V, h->ldv, h->rows_pad, nrows, j, h->d_c, fused ? h->d_r : nullptr, h->d_beta + beta_idx, h->tune[8], h->stream,
                      h->bnd_ranges[q + 1].first, h->bnd_ranges[q + 1].second, raw_c ? 1 : 0);
    }
    LZ_TRY(check_launch(h, "update(boundary)"));
    LZ_HIP(h, hipEventRecord(h->e_bnd, h->stream));
    for (auto& rg : h->int_ranges)
      launch_update(h->d_V, h->ldv, h->rows_pad, nrows, j, h->d_c, fused ? h->d_r : nullptr, h->d_beta + beta_idx, h->tune[8], h->stream,
                    rg.first, rg.second, raw_c ? 1 : 0);
    LZ_TRY(check_launch(h, "update(interior)"));
  }
  LZ_HIP(h, hipStreamWaitEvent(h->cstream, h->e_bnd, 0));
  h->acc.bytes[LZ_K_COMM] += 8.0 * (h->total_send + h->total_recv);
  h->acc.launches[LZ_K_COMM] += 1;
  h->n_exchange += 1;
  LZ_NCCL(h, g_rccl.GroupStart());
  for (size_t p = 0; p < h->peers.size(); ++p) {
    if (h->scount[p] > 0)
      LZ_NCCL(h, g_rccl.Send(vj + h->sstart[p], (size_t)h->scount[p], ncclDouble, h->peers[p], h->comm, h->cstream));
    if (h->rcount[p] > 0)
      LZ_NCCL(h, g_rccl.Recv(vj + h->rows_pad + h->roff[p], (size_t)h->rcount[p], ncclDouble, h->peers[p], h->comm, h->cstream));
  }
  LZ_NCCL(h, g_rccl.GroupEnd());
  LZ_HIP(h, hipEventRecord(h->e_halo, h->cstream));
  h->halo_inflight_j = j;
  return LZ_OK;
}

// r = r - alpha V[j] - beta V[jm1]; d_nrm2[0] = sum over ranks of ||r||^2
int step_three_term(lz_handle h, int j, int jm1, const double* d_alpha, const double* d_beta, bool need_norm) {
  const double M = (double)h->rows;
  int np = 0;
  {
    Scope sc(h, LZ_K_THREE, (jm1 >= 0 ? 32.0 : 24.0) * M, (jm1 >= 0 ? 6.0 : 4.0) * M);
    np = launch_three_term(h->d_r, h->d_V + (int64_t)j * h->ldv, jm1 >= 0 ? h->d_V + (int64_t)jm1 * h->ldv : nullptr, d_alpha,
                           d_beta, h->rows_pad, h->d_part, h->stream);
    LZ_TRY(check_launch(h, "three_term"));
  }
  if (!need_norm) return LZ_OK;  // fused-norm mode: ||r||^2 travels with the next Q^T r all-reduce
  {
    Scope sc(h, LZ_K_FINAL, 0, 0);
    launch_final_sum(h->d_part, np, h->d_nrm2, h->stream);
    LZ_TRY(check_launch(h, "final_sum(nrm2)"));
  }
  return comm_allreduce(h, h->d_nrm2, 1);
}

// ---- one-reduce mode (LZ_FLAG_ONE_REDUCE): the whole Krylov loop with ONE all-reduce per iteration ---------------------
// State entering step j: r holds the two-term residual r'' = A u - beta v_{j-2} of the newest vector u = v_{j-1} (at j = 0:
// r'' = A v0, u = v0 in basis row 0), and the local partial of alpha = u.(A u) sits in the reduce buffer.  Step j:
//   pass 1 dots rows 0..j-1 against BOTH columns (r'', u) + the three self terms   -> one all-reduce with alpha
//   prepare: alpha, c_i = V_i.r'' - alpha V_i.u, |r|^2 = r''.r'' - 2 alpha u.r'' + alpha^2 u.u
//   r = r'' - alpha u;  update: beta = |r|, V[j] = 2 r/beta - sum c_i/beta V_i - ...   (unchanged kernels from here)
//   exchange V[j]; r = A V[j] (alpha partial into the buffer); r'' = r - beta V[j-1]
inline int onered_ldp(int m) { return qtw_ldp(m + 2); }
inline int onered_slot(int m) { return onered_ldp(m) + m + 2; }  // where alpha lives in the reduce buffer at a step with m rows

int run_loop_onereduce(lz_handle h, int n) {
  const double M = (double)h->rows;
  LZ_TRY(step_spmv(h, 0, h->d_c + onered_slot(0), false));  // warm-up: r'' = A v0 (Lanczos.py:108), alpha0 partial
  for (int j = 0; j < n; ++j) {
    const int pstride = h->tune[7] > 1 ? h->tune[7] : 1;
    h->prof_iter = (j % pstride) == pstride / 2;
    const int bidx = (j + n - 2) % (n - 1);
    const int m = j, urow = j > 0 ? j - 1 : 0, ldp = onered_ldp(m);
    double* u = h->d_V + (int64_t)urow * h->ldv;
    h->qplan.variant = 0;
    {
      Scope sc(h, LZ_K_QTW, 8.0 * m * M + 16.0 * M, 4.0 * (m + 1) * M);
      LZ_HIP(h, launch_qtw(h->d_V, h->ldv, h->rows_pad, m, urow, h->d_r, nullptr, nullptr, h->qplan, h->d_part, 3, h->stream));
      LZ_TRY(check_launch(h, "qtw(two columns)"));
    }
    {
      Scope sc(h, LZ_K_FINAL, 0, 0);
      launch_final_rows_t(h->d_part, h->qplan.G, 2 * ldp, ldp + m + 2, h->d_c, h->stream);
      LZ_TRY(check_launch(h, "final_rows"));
    }
    LZ_TRY(comm_allreduce(h, h->d_c, onered_slot(m) + 1));  // THE collective of this iteration
    {
      Scope sc(h, LZ_K_FINAL, 0, 0);
      launch_onereduce_prepare(h->d_c, m, ldp, h->d_alpha + urow, h->d_nrm2 + 1, h->stream);  // j = 0: alpha[0] of the warm-up, rewritten below
      LZ_TRY(check_launch(h, "onereduce_prepare"));
    }
    {
      Scope sc(h, LZ_K_THREE, 24.0 * M, 2.0 * M);
      launch_three_term(h->d_r, u, nullptr, h->d_alpha + urow, nullptr, h->rows_pad, h->d_part, h->stream);  // r = r'' - alpha u
      LZ_TRY(check_launch(h, "three_term(alpha)"));
    }
    {
      Scope sc(h, LZ_K_UPDATE, 8.0 * j * M + 16.0 * M, 2.0 * (j + 1) * M);
      launch_update(h->d_V, h->ldv, h->rows_pad, j + 1, j, h->d_c, h->d_r, h->d_beta + bidx, h->tune[8] == 0 || h->tune[8] >= 3 ? h->tune[8] : 0,
                    h->stream, 0, -1, 1);
      LZ_TRY(check_launch(h, "update"));
    }
    const bool last = j == n - 1;
    LZ_TRY(step_spmv(h, j, last ? h->d_alpha + j : h->d_c + onered_slot(j + 1), last));  // the last alpha has no pass to ride on
    if (!last && j > 0) {
      Scope sc(h, LZ_K_THREE, 24.0 * M, 2.0 * M);
      launch_three_term(h->d_r, h->d_V + (int64_t)(j - 1) * h->ldv, nullptr, h->d_beta + bidx, nullptr, h->rows_pad, h->d_part, h->stream);
      LZ_TRY(check_launch(h, "three_term(beta)"));  // r'' = A V[j] - beta V[j-1]; at j = 0 the reference's V[-1] is the zero row
    }
  }
  return LZ_OK;
}

// ---- small problems: three launches per step instead of six ---------------------------------------------------------------
// When a vector is a handful of pass-1 slices, every kernel of a step does microseconds of work and the step costs what
// its six dependent launches cost.  Here the two second-stage reductions and the three-term recurrence ride in the
// prologue of their consumer: [pass 1: alpha from the SpMV's block partials, r = (y - alpha v) - beta v', stage, dots]
// [pass 2: coefficients from pass 1's block partials, update] [SpMV].  Same arithmetic, same summation trees: bit-identical
// to the six-launch path (tests/test_gpu_small.py).
// one-reduce partial loop: where the self terms' (and pass 1's) partials start in d_part - behind the alpha partials of ANY SpMV plan
size_t onered_part_off(lz_handle h) {
  size_t np = std::max<size_t>((size_t)h->rows / 4 + 2, (size_t)std::max(h->csr.n_rowblk, 1));
  if (h->csr.pb) np = std::max<size_t>(np, (size_t)pb_num_partials(h->csr.pb));
  return (np + 64 + 63) / 64 * 64;
}

size_t fused_coff(lz_handle h) {  // where pass 1's partials start in d_part (behind the SpMV's alpha partials)
  const size_t npmax = std::max<size_t>((size_t)h->rows / 4 + 2, (size_t)std::max(h->csr.n_rowblk, 1)) + 64;
  return (npmax + 63) / 64 * 64;
}

int run_loop_fused_small(lz_handle h, int n) {
  const double M = (double)h->rows;
  const size_t coff = fused_coff(h);
  int np = 0;
  LZ_TRY(step_spmv(h, 0, nullptr, false, &np));  // warm-up: y = A v0 (Lanczos.py:108)
  for (int j = 0; j < n; ++j) {
    const int pstride = h->tune[7] > 1 ? h->tune[7] : 1;
    h->prof_iter = (j % pstride) == pstride / 2;
    const int bidx = (j + n - 2) % (n - 1);
    QtwFuse fz;
    fz.apart = h->d_part;
    fz.np = np;
    fz.jprev = j > 0 ? j - 1 : 0;   // j == 0: the warm-up's alpha0 and r = A v0 - alpha0 v0 (Lanczos.py:109-110)
    fz.jprev2 = j >= 2 ? j - 2 : -1;  // the reference's V[-1] term at its step 0 is the zero row
    fz.beta_prev = h->d_beta + (j >= 2 ? j - 2 : 0);
    fz.alpha_out = h->d_alpha + fz.jprev;
    fz.r_out = h->d_r2;
    h->qplan.variant = 0;
    {
      Scope sc(h, LZ_K_QTW, 8.0 * j * M + 40.0 * M, 2.0 * (j + 1) * M + 4.0 * M);
      LZ_HIP(h, launch_qtw(h->d_V, h->ldv, h->rows_pad, j + 1, j, h->d_r, nullptr, nullptr, h->qplan, h->d_part + coff, 4, h->stream, &fz));
      LZ_TRY(check_launch(h, "qtw(fused three-term)"));
    }
    {
      Scope sc(h, LZ_K_UPDATE, 8.0 * j * M + 16.0 * M, 2.0 * (j + 1) * M);
      launch_update(h->d_V, h->ldv, h->rows_pad, j + 1, j, h->d_part + coff, h->d_r2, h->d_beta + bidx, 0, h->stream, 0, -1, 2, 0, 0, h->qplan.G,
                    qtw_ldp(j + 1));
      LZ_TRY(check_launch(h, "update(fused reduction)"));
    }
    LZ_TRY(step_spmv(h, j, nullptr, false, &np));
  }
  {
    Scope sc(h, LZ_K_FINAL, 0, 0);
    launch_final_sum(h->d_part, np, h->d_alpha + (n - 1), h->stream);  // the last alpha has no consumer kernel to ride in
    LZ_TRY(check_launch(h, "final_sum(alpha)"));
  }
  return LZ_OK;
}


// The default loop of problems that are neither small nor huge (any number of ranks, fused-norm mode, full
// re-orthogonalisation, at most kThreeTermFusedMaxRows rows per rank - measured: C2 (10^6 rows) +3.6 %, 6 400 .. 350 000 rows
// +4 .. 11 %, the headline's 10^7 rows +0.4 %: there the separate three-term kernel streams at a higher rate than the
// prologue does, and the six-launch loop stays):
// the three-term recurrence r = (A v_j - alpha_j v_j) - beta_{j-1} v_{j-1} rides in the prologue of the NEXT step's pass 1
// (k_qtw_mfma4<4>, alpha read back from its slot after the all-reduce) instead of being a pass of its own - five launches
// per step, one read-modify-write of r less, bit-identical coefficients and basis (tests/test_gpu_small.py).
// lz_set_tuning(h, 15, 1) selects the six-launch loop.
constexpr int64_t kThreeTermFusedMaxRows = 4'000'000;
int run_loop_three_term_fused(lz_handle h, int n) {
  const double M = (double)h->rows;
  LZ_TRY(step_spmv(h, 0));  // warm-up: y = A v0, alpha_0 (Lanczos.py:108-109)
  for (int j = 0; j < n; ++j) {
    const int pstride = h->tune[7] > 1 ? h->tune[7] : 1;
    h->prof_iter = (j % pstride) == pstride / 2;
    const int bidx = (j + n - 2) % (n - 1);
    QtwFuse fz;
    fz.apart = nullptr;
    fz.np = 0;
    fz.jprev = j > 0 ? j - 1 : 0;     // j == 0: the warm-up's r = A v0 - alpha0 v0 (Lanczos.py:110)
    fz.jprev2 = j >= 2 ? j - 2 : -1;  // the reference's V[-1] term at its step 0 is the zero row
    fz.beta_prev = h->d_beta + (j >= 2 ? j - 2 : 0);
    fz.alpha_out = h->d_alpha + fz.jprev;
    fz.r_out = h->d_r2;
    h->qplan.variant = 0;
    {
      Scope sc(h, LZ_K_QTW, 8.0 * j * M + 40.0 * M, 2.0 * (j + 1) * M + 4.0 * M);
      LZ_HIP(h, launch_qtw(h->d_V, h->ldv, h->rows_pad, j + 1, j, h->d_r, nullptr, nullptr, h->qplan, h->d_part, 4, h->stream, &fz));
      LZ_TRY(check_launch(h, "qtw(three-term in the prologue)"));
    }
    {
      Scope sc(h, LZ_K_FINAL, 0, 0);
      launch_final_rows(h->d_part, j + 1, h->qplan.P, h->d_c, h->stream, h->qplan.family == 2);
      LZ_TRY(check_launch(h, "final_rows"));
    }
    LZ_TRY(comm_allreduce(h, h->d_c, j + 1));
    {
      Scope sc(h, LZ_K_UPDATE, 8.0 * j * M + 16.0 * M, 2.0 * (j + 1) * M);
      launch_update(h->d_V, h->ldv, h->rows_pad, j + 1, j, h->d_c, h->d_r2, h->d_beta + bidx, 0, h->stream, 0, -1, 1);
      LZ_TRY(check_launch(h, "update"));
    }
    LZ_TRY(step_spmv(h, j));
  }
  return LZ_OK;
}

#ifdef LZ_KBENCH
constexpr int kSmallStepMaxN = 64;  // per-step kernels: every block redoes both passes over all n rows
// ---- kernel-bench build only: the retired small-problem engines (lz_small.hip) ------------------------------------------
// The whole run as ONE cooperative kernel (tune[15] == 2; 3 = on a plain grid), or one launch per step (tune[15] == 5,
// n <= 64).  Both are bit-identical to the multi-kernel path and both measured no faster than the three launches per step
// that are the default for small problems (a device-coherent round trip costs ~2 us on MI355X, about
// what a kernel boundary costs), so they left the product library in round 3; tests/test_gpu_small.py keeps their
// bit-identity checks against liblanczos_kbench.so.
bool small_args(lz_handle h, int n, SmallArgs& sa) {
  memset(&sa, 0, sizeof sa);
  sa.kind = h->kind;
  if (h->kind == 2) {
    sa.dense = h->d_dense;
    sa.lda = h->dense_lda;
    sa.nparts = (int)((h->rows + 3) / 4);
  } else {
    const CsrDev& A = h->csr;
    const bool fixed = !(h->flags & LZ_FLAG_SPMV_STREAM) && (A.fixed_k == 5 || A.fixed_k == 7);
    // (one lane walks one row in the engine: rows of more than 32 entries would turn into a chain of dependent loads)
    // (the engine replays the CSR kernels' alpha grouping; a row-class coded copy of 5- / 7-entry rows groups alpha per 512-row unit
    // exactly like k_spmv_fixed<K, 512>, so it may stand in for the default SpMV)
    const bool ell_other_grouping = A.ell_default && !(A.ell_coded && fixed);
    if (A.pb || ell_other_grouping || A.max_row_nnz > 32 || (fixed && A.fixed_rb != 512)) return false;
    sa.rowptr = A.rowptr;
    sa.colidx = A.colidx;
    sa.vals = A.vals;
    sa.rowblk = fixed ? nullptr : A.rowblk;
    sa.nparts = fixed ? (int)((h->rows + 511) / 512) : A.n_rowblk;
  }
  if (sa.nparts > 1024 || h->part_cap < (size_t)(2048 + h->rows_pad)) return false;
  sa.rows = (int)h->rows;
  sa.rows_pad = (int)h->rows_pad;
  sa.n = n;
  sa.ldv = h->ldv;
  sa.V = h->d_V;
  sa.y = h->d_r;
  sa.drow = h->d_part;
  sa.x0 = h->d_part + 2048;  // d_part holds >= 4096 doubles; rows_pad <= 1280
  sa.pc = h->d_c;
  sa.alpha = h->d_alpha;
  sa.beta = h->d_beta;
  sa.bar = reinterpret_cast<unsigned*>(h->d_nrm2);  // 16 bytes, zeroed by basis_alloc
  sa.xcc = reinterpret_cast<unsigned*>(h->d_part + 3400);
  return true;
}
bool small_engine_applies(lz_handle h) {
  SmallArgs sa;
  return small_args(h, 2, sa);
}

int run_small_engine(lz_handle h, int n, const double* v0_local, bool steps, bool* ran) {
  SmallArgs sa;
  *ran = small_args(h, n, sa);
  if (!*ran) return LZ_OK;
  LZ_HIP(h, hipMemcpyAsync(h->d_part + 2048, h->d_V, (size_t)h->rows_pad * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  if (steps) {
    // one launch per step: first SpMV, steps j = -1 .. n-2, the last alpha
    const int nb = h->kind == 2 ? (int)std::min<int64_t>(256, (h->rows + 3) / 4) : (int)std::max<int64_t>(1, (h->rows + kTPB - 1) / kTPB);
    const double Mr = (double)h->rows;
    LZ_HIP(h, launch_small_step(sa, 0, -1, nb, h->stream));
    for (int j = -1; j <= n - 2; ++j) {
      const int pstride = h->tune[7] > 1 ? h->tune[7] : 1;
      h->prof_iter = ((j + 1) % pstride) == pstride / 2;
      Scope sc(h, LZ_K_UPDATE, spmv_bytes(h) + 16.0 * (j + 2) * Mr + 40.0 * Mr, spmv_flops(h) + 4.0 * (j + 2) * Mr);
      LZ_HIP(h, launch_small_step(sa, 1, j, nb, h->stream));
    }
    h->prof_iter = true;
    LZ_HIP(h, launch_small_step(sa, 2, n - 1, 1, h->stream));
    return check_launch(h, "small_step");
  }
  h->acc.launches[LZ_K_FINAL] += 1;
  // tune[15] == 2: the participating blocks share one XCD (every eighth block of the grid); 3: plain grid over all XCDs
  LZ_HIP(h, launch_small_run(sa, small_grid(sa.rows_pad), h->tune[15] == 2, h->stream));
  LZ_TRY(check_launch(h, "small_run"));
  unsigned status = 0;
  LZ_HIP(h, hipMemcpyAsync(&status, sa.bar + 2, sizeof status, hipMemcpyDeviceToHost, h->stream));
  LZ_HIP(h, hipStreamSynchronize(h->stream));
  if (status != 0) {
    // the engine refused (its blocks were not dealt to one XCD) or a barrier timed out: the caller repeats the run
    if (getenv("LZ_DEBUG_TIMING")) fprintf(stderr, "[lz_run] small-problem engine gave up (status %u): multi-kernel path\n", status);
    *ran = false;
    LZ_TRY(basis_alloc(h, n, 1));
    LZ_HIP(h, hipMemcpyAsync(h->d_V, v0_local, (size_t)h->rows * sizeof(double), hipMemcpyHostToDevice, h->stream));
    LZ_HIP(h, hipEventRecord(h->run_a, h->stream));
  }
  return LZ_OK;
}
#endif  // LZ_KBENCH

// ---- which loop structure runs the Krylov iteration --------------------------------------------------------------------
// (the values are what lz_last_engine reports)
enum Loop {
  LOOP_SIX = 0,               // six launches per step; also the partial re-orthogonalisation mode and every A/B arm of a kernel
  LOOP_SMALL_ENGINE = 1,      // kernel-bench build only: the whole run as one cooperative kernel (lz_small.hip)
  LOOP_FUSED_SMALL = 2,       // <= 8 pass-1 slices, one rank: three launches per step (run_loop_fused_small)
  LOOP_THREE_TERM_FUSED = 3,  // up to 4e6 rows per rank: five launches per step (run_loop_three_term_fused)
  LOOP_SMALL_STEP = 4,        // kernel-bench build only: one launch per step
  LOOP_ONE_REDUCE_REPEATED = 5,  // a one-reduce run whose cancellation guard fired: repeated on the default loop
  LOOP_ONE_REDUCE = 6,        // LZ_FLAG_ONE_REDUCE: one all-reduce per iteration
  LOOP_PARTIAL_DEVICE = 7,    // LZ_FLAG_REORTH_PARTIAL, default: the omega-recurrence and the sweep decision live on the device
  LOOP_PARTIAL_ONE_REDUCE = 8 // LZ_FLAG_REORTH_PARTIAL | LZ_FLAG_ONE_REDUCE: the same with one all-reduce per iteration (look-ahead gate)
};

Loop choose_loop(lz_handle h, int n) {
  const int f = h->flags;
  const bool default_kernels = h->qplan.family == 2 && !(f & (LZ_FLAG_QTW_MFMA | LZ_FLAG_QTW_VALU)) && h->tune[1] == 0 && h->tune[8] == 0;
  const bool full_fused = (f & LZ_FLAG_FUSED_NORM) && !(f & LZ_FLAG_REORTH_PARTIAL);
  if ((f & LZ_FLAG_ONE_REDUCE) && !(f & LZ_FLAG_REORTH_PARTIAL) && h->qplan.family == 2) return LOOP_ONE_REDUCE;
  if ((f & LZ_FLAG_ONE_REDUCE) && (f & LZ_FLAG_REORTH_PARTIAL) && default_kernels && h->tune[18] != 1 && !(f & LZ_FLAG_OVERLAP_HALO))
    return LOOP_PARTIAL_ONE_REDUCE;
  // partial re-orthogonalisation: device-resident decisions with the default kernels (tune[18] == 1: the host-decided loop,
  // two scalars read back per step - kept for the bit-identity test and as an A/B arm)
  if ((f & LZ_FLAG_REORTH_PARTIAL) && default_kernels && h->tune[18] != 1 && !(f & LZ_FLAG_OVERLAP_HALO)) return LOOP_PARTIAL_DEVICE;
  const bool one_rank = h->world == 1 && h->comm_kind == 0;
#ifdef LZ_KBENCH
  const bool want_steps = h->tune[15] == 5 && n <= kSmallStepMaxN;
  if ((h->tune[15] == 2 || h->tune[15] == 3 || want_steps) && one_rank && full_fused && default_kernels && !(f & LZ_FLAG_SPMV_SCALAR) &&
      h->qplan.L == 512 && h->rows_pad <= kSmallMaxPadRows && n <= kSmallMaxPadRows && small_engine_applies(h))
    return want_steps ? LOOP_SMALL_STEP : LOOP_SMALL_ENGINE;
  const bool knob_auto = h->tune[15] == 0 || h->tune[15] == 5;
#else
  const bool knob_auto = h->tune[15] == 0;
#endif
  if (!knob_auto || !full_fused || !default_kernels) return LOOP_SIX;
  if (one_rank && h->qplan.G <= 8 && n <= 4096 && h->part_cap >= fused_coff(h) + (size_t)(n + 16) * (size_t)h->qplan.G) return LOOP_FUSED_SMALL;
  if (!(f & LZ_FLAG_OVERLAP_HALO) && h->rows_pad <= kThreeTermFusedMaxRows) return LOOP_THREE_TERM_FUSED;
  return LOOP_SIX;
}

// ---- the plain loop: six launches per step (pass 1, second-stage sums, pass 2, SpMV, alpha sum, three-term), with the
// opt-in partial re-orthogonalisation (Simon's omega-recurrence on the host) --------------------------------------------
int run_loop_six(lz_handle h, int n, int* sweeps_out, int j0 = 0) {
  const bool fused = (h->flags & LZ_FLAG_FUSED_NORM) != 0 && !(h->flags & LZ_FLAG_REORTH_PARTIAL);
  if (j0 == 0) {
    // warm-up (Lanczos.py:108-110): r = A v0; alpha0 = r.v0; r = r - alpha0 v0
    LZ_TRY(step_spmv(h, 0));
    LZ_TRY(step_three_term(h, 0, -1, h->d_alpha, nullptr, !fused));
  } else if (!fused) {
    // resumed run (lz_run_resume): steps 0 .. j0-1 are in the basis, r is the residual entering step j0; the scale-then-dot order
    // wants ||r||^2 in d_nrm2: r = r - 0 * V[0] leaves r unchanged bit for bit and refreshes it
    LZ_HIP(h, hipMemsetAsync(h->d_c + n, 0, sizeof(double), h->stream));
    LZ_TRY(step_three_term(h, 0, -1, h->d_c + n, nullptr, true));
  }
  const int pstride = h->tune[7] > 1 ? h->tune[7] : 1;
  const bool partial = (h->flags & LZ_FLAG_REORTH_PARTIAL) != 0;
  // Partial re-orthogonalisation (opt-in): Simon's omega-recurrence on the host, fed with alpha_j and beta_{j+1}
  // (two doubles copied back per step).  omega_{j,k} estimates v_j . v_k; a sweep is due when it exceeds sqrt(eps).
  const double eps = 2.220446049250313e-16, thresh = 1.4901161193847656e-08;
  std::vector<double> w_prev, w_cur, w_new, ha, hb;  // omega_{j-2,:}, omega_{j-1,:}, omega_{j,:}; alpha_k; beta_k (norm forming V[k])
  if (partial) {
    if (!h->h_pinned) LZ_HIP(h, hipHostMalloc(reinterpret_cast<void**>(&h->h_pinned), 8 * sizeof(double), hipHostMallocDefault));
    w_prev.assign((size_t)n + 1, 0.0);
    w_cur.assign((size_t)n + 1, 0.0);
    w_cur[0] = 1.0;  // omega_{0,0} = v_0 . v_0 (until round 4 this row was all zero, which made a spurious sweep due at j = 2)
    w_new.assign((size_t)n + 1, 0.0);
    ha.assign((size_t)n + 1, 0.0);
    hb.assign((size_t)n + 1, 0.0);
    double nrm2 = 0.0;
    LZ_HIP(h, hipMemcpyAsync(&nrm2, h->d_nrm2, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    LZ_HIP(h, hipStreamSynchronize(h->stream));
    h->host_syncs += 1;
    hb[0] = std::sqrt(nrm2);
  }
  bool force_next = false;
  double normA = 0.0;
  int sweeps = 0;
  for (int j = j0; j < n; ++j) {
    h->prof_iter = (j % pstride) == pstride / 2;  // centred sample: same mean j as the full run
    const int bidx = (j + n - 2) % (n - 1);  // beta[j-1] with Python's negative index at j = 0
    bool sweep = true;
    if (partial) {
      bool due = false;
      if (j >= 1) {
        // beta_j omega_{j,k} = beta_{k+1} omega_{j-1,k+1} + (alpha_k - alpha_{j-1}) omega_{j-1,k} + beta_k omega_{j-1,k-1}
        //                      - beta_{j-1} omega_{j-2,k}  (+ rounding of size eps ||A||),   k <= j-2
        std::fill(w_new.begin(), w_new.end(), 0.0);
        w_new[(size_t)j] = 1.0;
        w_new[(size_t)j - 1] = eps;
        double worst = 0.0;
        for (int k = 0; k + 2 <= j; ++k) {
          double t = hb[(size_t)k + 1] * w_cur[(size_t)k + 1] + (ha[(size_t)k] - ha[(size_t)j - 1]) * w_cur[(size_t)k] -
                     hb[(size_t)j - 1] * w_prev[(size_t)k];
          if (k > 0) t += hb[(size_t)k] * w_cur[(size_t)k - 1];
          t += (t < 0 ? -1.0 : 1.0) * 2.0 * eps * normA;
          w_new[(size_t)k] = t / hb[(size_t)j];
          worst = std::max(worst, std::fabs(w_new[(size_t)k]));
        }
        due = worst > thresh;
        std::swap(w_prev, w_cur);
        std::swap(w_cur, w_new);
      }
      sweep = (j == 0) || due || force_next;  // a due sweep also covers the next vector (both feed the recurrence)
      force_next = due;
      if (sweep)
        for (int k = 0; k < j; ++k) w_cur[(size_t)k] = eps;
    }
    if (sweep) {
      ++sweeps;
      LZ_TRY(step_reorth(h, j, j + 1, true, bidx, true));
    } else {
      Scope sc(h, LZ_K_QTW, 16.0 * (double)h->rows, (double)h->rows);
      launch_scale_store(h->d_V + (int64_t)j * h->ldv, h->d_r, h->d_nrm2, h->d_beta + bidx, h->rows_pad, h->stream);
      LZ_TRY(check_launch(h, "scale_store"));
    }
    LZ_TRY(step_spmv(h, j));
    // at j = 0 the reference subtracts beta * V[-1], the still-zero last row: a no-op
    LZ_TRY(step_three_term(h, j, j > 0 ? j - 1 : -1, h->d_alpha + j, h->d_beta + bidx, !fused || partial));
    if (partial) {
      double* two = h->h_pinned;
      LZ_HIP(h, hipMemcpyAsync(&two[0], h->d_alpha + j, sizeof(double), hipMemcpyDeviceToHost, h->stream));
      LZ_HIP(h, hipMemcpyAsync(&two[1], h->d_nrm2, sizeof(double), hipMemcpyDeviceToHost, h->stream));
      LZ_HIP(h, hipStreamSynchronize(h->stream));
      h->host_syncs += 1;
      ha[(size_t)j] = two[0];
      hb[(size_t)j + 1] = std::sqrt(two[1]);
      normA = std::max(normA, std::fabs(two[0]) + hb[(size_t)j] + hb[(size_t)j + 1]);
    }
  }
  *sweeps_out = sweeps;
  return LZ_OK;
}

// ---- partial re-orthogonalisation, device-resident (round 4; the default of LZ_FLAG_REORTH_PARTIAL) -----------------------
// The loop above with the host taken out: Simon's omega-recurrence runs in a one-block kernel behind the three-term kernel
// (k_omega, lz_reorth.hip; on one rank it also folds the ||r||^2 partials, so it costs no launch), which leaves a gate in
// device memory; the sweep kernels of the next step (pass 1, second-stage sums, pass 2) are always enqueued and return at
// once when the gate says no sweep is due.  No read-back, no hipStreamSynchronize between the first and the last launch
// (lz_last_host_syncs == 0 on one rank / over RCCL).  Every rank takes the same decision: its inputs are all-reduced sums.
// A step without a sweep on one rank with an ELL-ordered fixed-K matrix is TWO streaming kernels: the SpMV forms
// v_j = r / beta itself wherever it reads x (k_spmv_ell<.., SC>: the separate 16M-byte scale pass is gone; r and y
// ping-pong between two buffers), and the three-term kernel.  Other matrices keep the (gated) scale kernel.
// Decisions, coefficients and basis are bit-identical to the host-decided loop (tests/test_gpu_lanczos.py).
// j0 > 0 (lz_run_resume_partial): steps 0 .. j0-1 are in the basis, r is the residual entering step j0 and `state_j0` the omega state a
// run of j0 steps left behind (lz_get_omega_state: its layout is that of a j0-step run).
int run_loop_partial_device(lz_handle h, int n, int j0 = 0, const double* state_j0 = nullptr) {
  const double M = (double)h->rows;
  if (h->om_n < n) {
    LZ_TRY(dev_alloc(h, h->d_om, omega_state_doubles(n)));
    LZ_TRY(dev_alloc(h, h->d_omi, omega_state_ints(n) + 1));  // (+ the ticket of pass 1's folded second stage)
    h->om_n = n;
  }
  h->om_run_n = n;
  if (j0 > 0) {
    // the state of the j0-step run, re-laid for n steps: [normA, force | hb[0 .. n + 2) | three rows of n + 1]; entries past j0 are zero,
    // exactly what an uninterrupted n-step run holds there at this point
    std::vector<double> st(omega_state_doubles(n), 0.0);
    st[0] = state_j0[0];
    st[1] = state_j0[1];
    const double* hb_s = state_j0 + 2;
    const double* W_s = hb_s + (j0 + 2);
    double* hb_d = st.data() + 2;
    double* W_d = hb_d + (n + 2);
    for (int k = 0; k < j0 + 2; ++k) hb_d[k] = hb_s[k];
    for (int r = 0; r < 3; ++r)
      for (int k = 0; k <= j0; ++k) W_d[(size_t)r * (n + 1) + k] = W_s[(size_t)r * (j0 + 1) + k];
    LZ_HIP(h, hipMemcpyAsync(h->d_om, st.data(), st.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    LZ_HIP(h, hipMemsetAsync(h->d_omi, 0, (omega_state_ints(n) + 1) * sizeof(int), h->stream));
    LZ_HIP(h, hipStreamSynchronize(h->stream));  // (`st` is a local)
  }
  LZ_HIP(h, hipMemsetAsync(h->d_omi + omega_state_ints(n), 0, sizeof(int), h->stream));
  const int* gate = h->d_omi;
  const bool one_rank = h->world <= 1 && !(h->tune[6] && h->comm_kind);
  if (one_rank && h->kind == 1 && h->tune[18] != 2 && h->tune[17] != 1 && !h->csr.ell_rb && !h->csr.pb &&
      (h->csr.fixed_k == 5 || h->csr.fixed_k == 7))
    // first partial run on this matrix: the ELL copy the fused r / beta needs.  (5 and 7 entries per row: its alpha partials
    // are grouped exactly like k_spmv_fixed's - 512-row blocks, rows t and t + 256 per lane - so the fused and the unfused loop
    // produce the same bits; 27-point rows keep the scale kernel + the CSR-stream SpMV)
    LZ_HIP(h, ell_build(h->csr, 0, h->stream));
  const bool fuse_scale = one_rank && h->kind == 1 && ell_usable(h->csr, h->flags) && h->tune[18] != 2;
  // warm-up (Lanczos.py:108-110): r = A v0; alpha0 = r.v0; r = r - alpha0 v0; ||r||^2
  if (j0 == 0) LZ_TRY(step_spmv(h, 0));
  int np = 0;
  auto three_term_and_decide = [&](int j, int jm1, const double* d_alpha, const double* d_beta, double* r, bool decide, int jn) -> int {
    {
      Scope sc(h, LZ_K_THREE, (jm1 >= 0 ? 32.0 : 24.0) * M, (jm1 >= 0 ? 6.0 : 4.0) * M);
      np = launch_three_term(r, h->d_V + (int64_t)j * h->ldv, jm1 >= 0 ? h->d_V + (int64_t)jm1 * h->ldv : nullptr, d_alpha, d_beta, h->rows_pad,
                             h->d_part, h->stream);
      LZ_TRY(check_launch(h, "three_term"));
    }
    Scope sc(h, LZ_K_FINAL, 0, 0);
    if (one_rank && decide) {
      launch_omega(h->d_part, np, h->d_nrm2, h->d_alpha, jn, n, h->d_om, h->d_omi, h->stream);
      return check_launch(h, "final_sum(nrm2) + omega");
    }
    launch_final_sum(h->d_part, np, h->d_nrm2, h->stream);
    LZ_TRY(check_launch(h, "final_sum(nrm2)"));
    LZ_TRY(comm_allreduce(h, h->d_nrm2, 1));
    if (decide) {
      launch_omega(nullptr, 0, h->d_nrm2, h->d_alpha, jn, n, h->d_om, h->d_omi, h->stream, h->d_c);  // (d_c: zero unless step jn sweeps - it is all-reduced every step)
      LZ_TRY(check_launch(h, "omega"));
    }
    return LZ_OK;
  };
  if (j0 == 0) {
    LZ_TRY(three_term_and_decide(0, -1, h->d_alpha, nullptr, h->d_r, true, 0));
  } else {
    // resumed: r = r - 0 * V[0] leaves r unchanged bit for bit, refreshes ||r||^2 with the very kernel (and summation) that produced it
    // in the uninterrupted run, and k_omega takes the decision of step j0 from it
    LZ_HIP(h, hipMemsetAsync(h->d_c + n, 0, sizeof(double), h->stream));
    LZ_TRY(three_term_and_decide(0, -1, h->d_c + n, nullptr, h->d_r, true, j0));
  }
  const int pstride = h->tune[7] > 1 ? h->tune[7] : 1;
  double* rcur = h->d_r;   // the residual entering the step
  double* rnext = h->d_r2; // where the fused SpMV writes y (it reads r through its gathers: not in place)
  for (int j = j0; j < n; ++j) {
    h->prof_iter = (j % pstride) == pstride / 2;
    const int bidx = (j + n - 2) % (n - 1);
    double* vj = h->d_V + (int64_t)j * h->ldv;
    // the sweep (gated; bytes are accounted after the run from the device's sweep log: the host does not know which ran)
    {
      QtwFuse fz;
      fz.gate = gate;
      // pass 1's last block adds the blocks' runs itself (k_final_rows_t's order: same bits): one gated launch less per step.
      // (tune[18] == 3: the separate second-stage kernel, A/B)
      const bool fold = h->tune[18] != 3;
      if (fold) {
        fz.ticket = reinterpret_cast<unsigned*>(h->d_omi + omega_state_ints(n));
        fz.c_out = h->d_c;
      }
      h->qplan.variant = 0;
      {
        Scope sc(h, LZ_K_QTW, 0, 0);
        LZ_HIP(h, launch_qtw(h->d_V, h->ldv, h->rows_pad, j + 1, j, rcur, h->d_nrm2, h->d_beta + bidx, h->qplan, h->d_part, 1, h->stream, &fz));
        LZ_TRY(check_launch(h, "qtw(gated)"));
      }
      if (!fold) {
        Scope sc(h, LZ_K_FINAL, 0, 0);
        launch_final_rows(h->d_part, j + 1, h->qplan.P, h->d_c, h->stream, true, gate);
        LZ_TRY(check_launch(h, "final_rows(gated)"));
      }
      LZ_TRY(comm_allreduce(h, h->d_c, j + 1));  // (N > 1: issued every step - the host cannot skip a collective the device may need)
      {
        Scope sc(h, LZ_K_UPDATE, 0, 0);
        launch_update(h->d_V, h->ldv, h->rows_pad, j + 1, j, h->d_c, nullptr, h->d_beta + bidx, 0, h->stream, 0, -1, 0, 0, 0, 0, 0, gate);
        LZ_TRY(check_launch(h, "update(gated)"));
      }
    }
    if (fuse_scale) {
      SpmvScale ss;
      ss.r = rcur;
      ss.nrm2 = h->d_nrm2;
      ss.vj = vj;
      ss.beta_slot = h->d_beta + bidx;
      ss.gate = gate;
      int npa = 0;
      {
        // (the scale pass's 16M bytes ride here: BASELINE.md's accounting of the step is unchanged.  With a row-class coded matrix the
        // launch moves 25 bytes per row - r once, y, V[j], the class byte - so only V[j]'s 8 are added: counting r twice would be a
        // third of the total there)
        Scope sc(h, LZ_K_SPMV, spmv_bytes(h, true) + (h->csr.ell_coded ? 8.0 : 16.0) * M, spmv_flops(h) + M);
        npa = launch_spmv_ell(h->csr, vj, rnext, vj, h->d_part, h->stream, &ss);
        LZ_TRY(check_launch(h, "spmv(ell, scale fused)"));
      }
      {
        Scope sc(h, LZ_K_FINAL, 0, 0);
        launch_final_sum(h->d_part, npa, h->d_alpha + j, h->stream);
        LZ_TRY(check_launch(h, "final_sum(alpha)"));
      }
      std::swap(rcur, rnext);
    } else {
      {
        Scope sc(h, LZ_K_QTW, 16.0 * M, M);
        launch_scale_store(vj, rcur, h->d_nrm2, h->d_beta + bidx, h->rows_pad, h->stream, gate);
        LZ_TRY(check_launch(h, "scale_store(gated)"));
      }
      LZ_TRY(step_spmv(h, j));  // r = A V[j] into h->d_r (== rcur), alpha_j
    }
    // at j = 0 the reference subtracts beta * V[-1], the still-zero last row: a no-op
    LZ_TRY(three_term_and_decide(j, j > 0 ? j - 1 : -1, h->d_alpha + j, h->d_beta + bidx, rcur, j + 1 < n, j + 1));
  }
  if (rcur != h->d_r) std::swap(h->d_r, h->d_r2);  // the residual entering step n is what lz_get_residual hands out
  return LZ_OK;
}

// ---- partial re-orthogonalisation with ONE all-reduce per step (LZ_FLAG_REORTH_PARTIAL | LZ_FLAG_ONE_REDUCE; round 5) --------
// The device-decided loop above issues three collectives per step on a partition (alpha, ||r||^2 and - used or not - the
// coefficient vector): at the N = 8 rank share a step without a sweep is ~26 us of kernels, so three latency-bound all-reduces
// would make the selective arm scale negatively.  Here a step costs ONE all-reduce + ONE exchange, sweep or no sweep, exactly like
// run_loop_onereduce: the single buffer [p_0..p_{m-1}, r''.r'' | q_0..q_{m-1}, u.u, u.r'', alpha] carries alpha's partial, the three
// self terms and the sweep's dots (zero in a step without a sweep); the gate is read by every kernel from device memory and is the
// same on every rank because it is derived from reduced sums only (k_partial_onered_post: a one-step look-ahead of Simon's
// recurrence, since the exact test of omega_{j,:} needs the very sums this all-reduce delivers).
// Per step:  [pass 1, two columns - gated]  [second stage - gated]  ALL-REDUCE  [post: alpha, c, ||r||^2, omega, next gate]
//            [r = r'' - alpha u]  [update - gated | scale V[j] = r / beta - gated the other way]  EXCHANGE  [SpMV + alpha partial]
//            [r'' = A v_j - beta v_{j-1} + the three self terms' partials]  [their sums + alpha's into the next buffer]
// Two buffers alternate (the update of step j still reads buffer j & 1 while the post kernel clears the other).  No host
// synchronisation inside the loop over RCCL.  The three-sum ||r||^2 cancels like the full one-reduce loop's: same guard, same
// repeat of the solve (on the three-collective device loop) when it fires.
int run_loop_partial_onereduce(lz_handle h, int n) {
  const double M = (double)h->rows;
  if (h->om_n < n + 4) {  // (omega_onered_ints(n) <= omega_state_ints(n + 4) + 1)
    LZ_TRY(dev_alloc(h, h->d_om, omega_state_doubles(n + 4)));
    LZ_TRY(dev_alloc(h, h->d_omi, omega_state_ints(n + 4) + 1));
    h->om_n = n + 4;
  }
  const double kappa = h->tune[20] > 0 ? (double)h->tune[20] : 4.0;
  const size_t bstride = (size_t)2 * qtw_ldp(n + 2) + 8;
  {
    Scope sc(h, LZ_K_FINAL, 0, 0);
    launch_partial_onered_post(nullptr, nullptr, 0, 0, 0, nullptr, nullptr, nullptr, -1, n, h->d_om, h->d_omi, kappa, h->stream);
    LZ_TRY(check_launch(h, "partial_onered_post(init)"));
  }
  unsigned* ticket = reinterpret_cast<unsigned*>(h->d_omi + omega_onered_ints(n));
  LZ_HIP(h, hipMemsetAsync(ticket, 0, sizeof(unsigned), h->stream));
  const bool lean = h->tune[18] != 3;  // (knob 18 = 3: every folded stage as a kernel of its own - the first form of this loop, kept as the A/B arm)
  LZ_TRY(step_spmv(h, 0, h->d_c + onered_slot(0), false));  // warm-up: r'' = A v0 (Lanczos.py:108), alpha0 partial
  const int pstride = h->tune[7] > 1 ? h->tune[7] : 1;
  const size_t self_off = onered_part_off(h);  // the self terms' partials live behind the SpMV's alpha partials in d_part
  for (int j = 0; j < n; ++j) {
    h->prof_iter = (j % pstride) == pstride / 2;
    const int bidx = (j + n - 2) % (n - 1);
    const int m = j, urow = j > 0 ? j - 1 : 0, ldp = onered_ldp(m);
    const bool last = j == n - 1;
    double* buf = h->d_c + (size_t)(j & 1) * bstride;
    double* bufn = h->d_c + (size_t)((j + 1) & 1) * bstride;
    const int* gate = h->d_omi + (j & 1);
    double* u = h->d_V + (int64_t)urow * h->ldv;
    double* vj = h->d_V + (int64_t)j * h->ldv;
    h->qplan.variant = 0;
    {
      QtwFuse fz;
      fz.gate = gate;
      if (lean) {  // pass 1's last block adds the blocks' runs itself (k_final_rows_t's order)
        fz.ticket = ticket;
        fz.c_out = buf;
      }
      Scope sc(h, LZ_K_QTW, 0, 0);  // (bytes of the launches that really ran: accounted after the run from the device's sweep log)
      LZ_HIP(h, launch_qtw(h->d_V, h->ldv, h->rows_pad, m, urow, h->d_r, nullptr, nullptr, h->qplan, h->d_part + (lean ? self_off : 0), 3, h->stream, &fz));
      LZ_TRY(check_launch(h, "qtw(two columns, gated)"));
    }
    if (!lean) {
      Scope sc(h, LZ_K_FINAL, 0, 0);
      launch_final_rows_t(h->d_part, h->qplan.G, 2 * ldp, ldp + m + 2, buf, h->stream, gate);
      LZ_TRY(check_launch(h, "final_rows(gated)"));
    }
    LZ_TRY(comm_allreduce(h, buf, onered_slot(m) + 1));  // THE collective of this iteration
    {
      Scope sc(h, LZ_K_FINAL, 0, 0);
      launch_partial_onered_post(buf, last ? nullptr : bufn, last ? 0 : onered_slot(j + 1) + 1, m, ldp, h->d_alpha + urow, h->d_nrm2, h->d_alpha, j, n,
                                 h->d_om, h->d_omi, kappa, h->stream);
      LZ_TRY(check_launch(h, "partial_onered_post"));
    }
    if (lean) {
      // ONE launch forms w = (r'' - alpha u) / beta and either sweeps (V[j] = 2 w - sum c_i V_i) or stores V[j] = w
      Scope sc(h, LZ_K_UPDATE, 32.0 * M, 4.0 * M);  // (a sweep's basis rows: accounted after the run)
      launch_update(h->d_V, h->ldv, h->rows_pad, j + 1, j, buf, h->d_r, h->d_beta + bidx, 0, h->stream, 0, -1, 1, 0, 0, 0, 0, gate, u, h->d_alpha + urow);
      LZ_TRY(check_launch(h, "update | scale (gated)"));
    } else {
      {
        Scope sc(h, LZ_K_THREE, 24.0 * M, 2.0 * M);
        launch_three_term(h->d_r, u, nullptr, h->d_alpha + urow, nullptr, h->rows_pad, h->d_part, h->stream);  // r = r'' - alpha u
        LZ_TRY(check_launch(h, "three_term(alpha)"));
      }
      {
        Scope sc(h, LZ_K_UPDATE, 0, 0);
        launch_update(h->d_V, h->ldv, h->rows_pad, j + 1, j, buf, h->d_r, h->d_beta + bidx, 0, h->stream, 0, -1, 1, 0, 0, 0, 0, gate);
        LZ_TRY(check_launch(h, "update(gated)"));
      }
      {
        Scope sc(h, LZ_K_QTW, 16.0 * M, M);
        launch_scale_store(vj, h->d_r, h->d_nrm2, h->d_beta + bidx, h->rows_pad, h->stream, gate);
        LZ_TRY(check_launch(h, "scale_store(gated)"));
      }
    }
    if (last) {
      LZ_TRY(step_spmv(h, j, h->d_alpha + j, true));  // the last alpha has no pass to ride on
      break;
    }
    // r'' = A V[j] - beta V[j-1] (at j = 0 the reference's V[-1] is the zero row) + the self terms of the next step's ||r||^2;
    // the four second-stage sums (alpha's partial and the three self terms) leave in one launch
    const int mn = j + 1, ldpn = onered_ldp(mn);
    int npa = 0, G3 = 0;
    if (lean)
      LZ_TRY(step_spmv(h, j, nullptr, false, &npa));
    else
      LZ_TRY(step_spmv(h, j, bufn + onered_slot(mn), false));
    double* spart = h->d_part + (lean ? self_off : 0);
    {
      Scope sc(h, LZ_K_THREE, (j > 0 ? 40.0 : 16.0) * M, (j > 0 ? 8.0 : 6.0) * M);
      G3 = launch_three_term_self(h->d_r, vj, j > 0 ? h->d_V + (int64_t)(j - 1) * h->ldv : nullptr, h->d_beta + bidx, h->rows_pad, spart, h->stream);
      LZ_TRY(check_launch(h, "three_term(beta) + self terms"));
    }
    FinalMulti fm;
    for (int q = 0; q < 3; ++q) {
      fm.part[q] = spart + (size_t)q * G3;
      fm.n[q] = G3;
    }
    fm.out[0] = bufn + mn;             // r''.r''
    fm.out[1] = bufn + ldpn + mn + 1;  // u.r''
    fm.out[2] = bufn + ldpn + mn;      // u.u
    fm.part[3] = h->d_part, fm.n[3] = npa, fm.out[3] = bufn + onered_slot(mn);  // alpha's partial (k_final_sum's grouping)
    Scope sc(h, LZ_K_FINAL, 0, 0);
    launch_final_sum_multi(fm, lean ? 4 : 3, h->stream);
    LZ_TRY(check_launch(h, "final_sum(alpha, self terms)"));
  }
  return LZ_OK;
}

// after the final synchronisation of lz_run: the device's sweep log -> lz_last_sweeps and the byte / flop accounting of the
// gated launches (pass 1: 8 j M + 16 M bytes, pass 2 the same; in a swept step the scale kernel / fused scale did no work)
void account_partial_device(lz_handle h, int n, const std::vector<int>& log, int* sweeps_out, size_t log0 = 2) {
  const double M = (double)h->rows;
  const int pstride = h->tune[7] > 1 ? h->tune[7] : 1;
  int sweeps = 0;
  for (int j = 0; j < n; ++j) {
    if (!log[log0 + j]) continue;
    ++sweeps;
    // (pass 1's read of r and write of V[j], 16 M bytes, are on the books already: the scale pass is accounted in every step)
    const double flops = 2.0 * (j + 1) * M;
    const bool timed = (h->flags & LZ_FLAG_PROFILE) != 0 && (j % pstride) == pstride / 2;
    for (int cls : {LZ_K_QTW, LZ_K_UPDATE}) {
      const double bytes = 8.0 * j * M + (cls == LZ_K_UPDATE ? 16.0 * M : 0.0);
      h->acc.bytes[cls] += bytes;
      h->acc.flops[cls] += flops;
      if (timed) h->acc.timed_bytes[cls] += bytes;
    }
  }
  *sweeps_out = sweeps;
}

// Breakdown report.  The reference divides by beta blindly (Lanczos.py:113): an exhausted Krylov
// space gives it a residual of rounding noise (or an exact zero and then inf/NaN), and it carries on.  So does this
// run - the coefficients are delivered exactly as computed - but the status says so: a beta at or below 64 eps times
// the scale of T (max |alpha|, |beta|), or any non-finite coefficient.  beta[n-2] is also where step j = 0 parks its
// norm before step n-1 overwrites it, so every entry of beta_out has been a divisor.
int breakdown_status(lz_handle h, int n, const double* alpha_out, const double* beta_out) {
  double tscale = 0.0;
  for (int j = 0; j < n; ++j) {
    if (std::isfinite(alpha_out[j])) tscale = std::max(tscale, std::fabs(alpha_out[j]));
    if (j < n - 1 && std::isfinite(beta_out[j])) tscale = std::max(tscale, std::fabs(beta_out[j]));
  }
  const double tiny = 64.0 * 2.220446049250313e-16 * tscale;
  for (int j = 0; j < n; ++j) {
    const bool bad_a = !std::isfinite(alpha_out[j]);
    const bool bad_b = j < n - 1 && !(std::isfinite(beta_out[j]) && beta_out[j] > tiny);
    if (bad_a || bad_b) {
      char msg[200];
      if (bad_b && std::isfinite(beta_out[j]))
        snprintf(msg, sizeof msg, "lz_run: Lanczos breakdown - beta[%d] = %.3e <= 64 eps * %.3e: the Krylov space is exhausted, later vectors are rounding noise",
                 j, beta_out[j], tscale);
      else
        snprintf(msg, sizeof msg, "lz_run: Lanczos breakdown - %s[%d] is not finite (a residual norm reached zero)", bad_b ? "beta" : "alpha", j);
      h->err = msg;
      return LZ_WARN_BREAKDOWN;
    }
  }
  return LZ_OK;
}

}  // namespace api
}  // namespace lz

extern "C" {

// ---- early allocation of the big buffers ---------------------------------------------------------------------------------
int lz_reserve(lz_handle h, int64_t rows_local, int n, int with_ritz) {
  if (!h) return LZ_ERR_ARG;
  if (rows_local <= 0 || n < 1) return fail(nullptr, LZ_ERR_ARG, "lz_reserve: bad sizes");
  if (hipSetDevice(h->dev) != hipSuccess) return LZ_ERR_HIP;  // (h->err belongs to the thread that drives the handle: not written here)
  const int64_t rows_pad = round_up(rows_local, kPadDoubles);
  const size_t vsz = (size_t)n * (size_t)skew_stride(h, rows_pad);
  const size_t ysz = y_doubles(rows_local, n);
  // One buffer at a time: decide under the lock, allocate / free OUTSIDE it (a device allocation may take seconds and a free
  // synchronises the device: lz_run's basis_alloc and lz_ritz_vectors take the same lock to adopt a buffer), publish under the lock.
  // with_ritz: 0 the basis only, 1 both, 2 the Ritz vectors only (the helper thread's second call used to re-check the
  // basis and could strand a second 8 n M bytes once basis_alloc had adopted the first).
  auto reserve_one = [&](double*& slot, size_t& slot_count, size_t want) {
    double* stale = nullptr;
    {
      std::lock_guard<std::mutex> lk(h->res_mu);
      if (slot && slot_count >= want) return;
      stale = slot;
      slot = nullptr;
      slot_count = 0;
    }
    if (stale) (void)big_free(stale);
    size_t free_b = 0, total_b = 0;
    // leave room for the matrix, its layouts and the work vectors: reserve only what leaves a quarter of the device free
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || want * sizeof(double) + total_b / 4 > free_b) return;
    void* p = nullptr;
    if (big_alloc(h->dev, &p, want * sizeof(double)) != hipSuccess) {
      (void)hipGetLastError();
      return;  // not an error: basis_alloc / lz_ritz_vectors allocate (and report) themselves
    }
    double* loser = nullptr;
    {
      std::lock_guard<std::mutex> lk(h->res_mu);
      if (slot) {
        loser = static_cast<double*>(p);  // (another reserve call got there first)
      } else {
        slot = static_cast<double*>(p);
        slot_count = want;
      }
    }
    if (loser) (void)big_free(loser);
  };
  if (with_ritz != 2) reserve_one(h->res_V, h->res_V_count, vsz);
  if (with_ritz) reserve_one(h->res_Y, h->res_Y_count, ysz);
  return LZ_OK;
}

// ---- the run -----------------------------------------------------------------------
int lz_run(lz_handle h, int n, const double* v0_local, double* alpha_out, double* beta_out) {
  if (!h) return LZ_ERR_ARG;
  if (!v0_local || !alpha_out || !beta_out) return fail(h, LZ_ERR_ARG, "lz_run: NULL buffer");
  if (n < 2) return fail(h, LZ_ERR_ARG, "lz_run: n must be >= 2 (the reference's beta array has n-1 entries)");
  if (n > h->Mg) return fail(h, LZ_ERR_ARG, "lz_run: n cannot be larger than M");
  const bool dbg = getenv("LZ_DEBUG_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  LZ_TRY(basis_alloc(h, n, 1));
  h->halo_inflight_j = -1;
  h->y_n = 0;  // the Ritz vectors of an earlier run are not this run's: fetches answer LZ_ERR_STATE until lz_ritz_vectors is called again
  const double t1 = now();
  LZ_TRY(upload(h, h->d_V, v0_local, (size_t)h->rows * sizeof(double)));
  const double t2 = now();
  LZ_HIP(h, hipEventRecord(h->run_a, h->stream));
  h->host_syncs = 0;
  const Loop loop = choose_loop(h, n);
  const bool one_reduce = loop == LOOP_ONE_REDUCE || loop == LOOP_PARTIAL_ONE_REDUCE;
  int sweeps = n;
  h->last_engine = (int)loop;
  switch (loop) {
#ifdef LZ_KBENCH
    case LOOP_SMALL_ENGINE:
    case LOOP_SMALL_STEP: {
      bool ran = false;
      LZ_TRY(run_small_engine(h, n, v0_local, loop == LOOP_SMALL_STEP, &ran));
      if (!ran) {  // the engine refused (placement / barrier timeout): nothing is lost, the plain loop repeats the run
        h->last_engine = LOOP_SIX;
        LZ_TRY(run_loop_six(h, n, &sweeps));
      }
      break;
    }
#endif
    case LOOP_FUSED_SMALL: LZ_TRY(run_loop_fused_small(h, n)); break;
    case LOOP_THREE_TERM_FUSED: LZ_TRY(run_loop_three_term_fused(h, n)); break;
    case LOOP_ONE_REDUCE: LZ_TRY(run_loop_onereduce(h, n)); break;
    case LOOP_PARTIAL_DEVICE: LZ_TRY(run_loop_partial_device(h, n)); break;
    case LOOP_PARTIAL_ONE_REDUCE: LZ_TRY(run_loop_partial_onereduce(h, n)); break;
    default: LZ_TRY(run_loop_six(h, n, &sweeps)); break;
  }
  h->last_sweeps = sweeps;
  h->r_state = (h->last_engine == LOOP_SIX || h->last_engine == LOOP_PARTIAL_DEVICE) ? 1
               : (h->last_engine == LOOP_FUSED_SMALL || h->last_engine == LOOP_THREE_TERM_FUSED) ? 2 : 0;
  h->prof_iter = true;
  LZ_HIP(h, hipEventRecord(h->run_b, h->stream));
  const double t3 = now();
  h->run_timed = true;
  LZ_HIP(h, hipMemcpyAsync(alpha_out, h->d_alpha, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  LZ_HIP(h, hipMemcpyAsync(beta_out, h->d_beta, (size_t)(n - 1) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  double onered_bad = 0.0;
  if (one_reduce) LZ_HIP(h, hipMemcpyAsync(&onered_bad, h->d_nrm2 + 1, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  std::vector<int> sweep_log;
  if (loop == LOOP_PARTIAL_DEVICE || loop == LOOP_PARTIAL_ONE_REDUCE) {
    sweep_log.resize(loop == LOOP_PARTIAL_DEVICE ? omega_state_ints(n) : omega_onered_ints(n));
    LZ_HIP(h, hipMemcpyAsync(sweep_log.data(), h->d_omi, sweep_log.size() * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  }
  LZ_HIP(h, hipStreamSynchronize(h->stream));
  h->last_misses = 0;
  h->sweep_log.assign((size_t)n, 1);  // which steps ran the sweep (lz_last_sweep_log): all of them unless a partial loop says otherwise
  if (loop == LOOP_SIX && (h->flags & LZ_FLAG_REORTH_PARTIAL)) h->sweep_log.clear();  // (the host-decided loop keeps no per-step record)
  if (loop == LOOP_PARTIAL_DEVICE) {
    account_partial_device(h, n, sweep_log, &h->last_sweeps);
    for (int j = 0; j < n; ++j) h->sweep_log[(size_t)j] = sweep_log[(size_t)2 + j] != 0;
  }
  if (loop == LOOP_PARTIAL_ONE_REDUCE) {
    account_partial_device(h, n, sweep_log, &h->last_sweeps, 4);
    h->last_misses = sweep_log[3];
    for (int j = 0; j < n; ++j) h->sweep_log[(size_t)j] = sweep_log[(size_t)4 + j] != 0;
  }
  if (one_reduce && onered_bad != 0.0) {
    // cancellation guard of the one-reduce loop (k_onereduce_prepare): |r|^2 = r''.r'' - 2 alpha u.r'' + alpha^2 u.u lost too
    // many digits at some step (|alpha| >> beta).  Every rank sees the same reduced sums, so every rank takes this branch:
    // the solve is repeated on the default loop (two all-reduces per iteration), whose coefficients hold the bar.
    h->run_timed = false;
    const int keep = h->flags;
    h->flags &= ~LZ_FLAG_ONE_REDUCE;
    const int rc = lz_run(h, n, v0_local, alpha_out, beta_out);
    h->flags = keep;
    h->last_engine = LOOP_ONE_REDUCE_REPEATED;
    return rc;
  }
  if (dbg)
    fprintf(stderr, "[lz_run] alloc+memset %.3f ms, v0 upload %.3f ms, enqueue loop %.3f ms, drain+D2H %.3f ms\n", t1 - t0, t2 - t1,
            t3 - t2, now() - t3);
  {
    float ms = 0.f;
    LZ_HIP(h, hipEventElapsedTime(&ms, h->run_a, h->run_b));
    h->acc.total_ms += ms;
    h->run_timed = false;
  }
  return breakdown_status(h, n, alpha_out, beta_out);
}

/* ---- checkpoint / resume ("Optional: dump (alpha, beta, j, V[:j])") ---------------------------------- */
int lz_get_residual(lz_handle h, double* r_local) {
  if (!h || !r_local) return LZ_ERR_ARG;
  LZ_TRY(require_basis(h, 0));
  LZ_HIP(h, hipSetDevice(h->dev));
  const int n = h->n;
  if (h->r_state == 2) {
    // the three- / five-launch loops leave y = A v_{n-1} behind: their three-term recurrence rides in the NEXT step's pass 1.
    // Form r = (y - alpha_{n-1} v_{n-1}) - beta_{n-2} v_{n-2} now, with the same kernel and expression (Lanczos.py:119).
    Scope sc(h, LZ_K_THREE, 32.0 * (double)h->rows, 6.0 * (double)h->rows);
    launch_three_term(h->d_r, h->d_V + (int64_t)(n - 1) * h->ldv, n >= 2 ? h->d_V + (int64_t)(n - 2) * h->ldv : nullptr, h->d_alpha + (n - 1),
                      h->d_beta + (n >= 2 ? n - 2 : 0), h->rows_pad, h->d_part, h->stream);
    LZ_TRY(check_launch(h, "three_term(residual)"));
    h->r_state = 1;
  }
  if (h->r_state != 1)
    return fail(h, LZ_ERR_STATE, "lz_get_residual: the last run left no residual (run lz_run first; not after the one-reduce loop or lz_run_two_sided)");
  LZ_HIP(h, hipMemcpyAsync(r_local, h->d_r, (size_t)h->rows * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  LZ_HIP(h, hipStreamSynchronize(h->stream));
  return LZ_OK;
}

int lz_get_omega_state(lz_handle h, double* out, int64_t count) {
  if (!h || !out) return LZ_ERR_ARG;
  if (h->last_engine != LOOP_PARTIAL_DEVICE || !h->d_om || h->om_run_n < 2)
    return fail(h, LZ_ERR_STATE, "lz_get_omega_state: the last run was not the device-decided partial re-orthogonalisation loop (engine 7)");
  if (count != (int64_t)omega_state_doubles(h->om_run_n)) return fail(h, LZ_ERR_ARG, "lz_get_omega_state: count must be 2 + (n + 2) + 3 (n + 1) for the n of the last run");
  LZ_HIP(h, hipSetDevice(h->dev));
  LZ_HIP(h, hipMemcpyAsync(out, h->d_om, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  LZ_HIP(h, hipStreamSynchronize(h->stream));
  return LZ_OK;
}

int lz_run_resume_partial(lz_handle h, int n, int j0, const double* V_rows, int64_t ldv_in, const double* r_local, const double* alpha_in,
                          const double* beta_in, const double* omega_state, double* alpha_out, double* beta_out) {
  if (!h) return LZ_ERR_ARG;
  if (!V_rows || !r_local || !alpha_in || !beta_in || !omega_state || !alpha_out || !beta_out) return fail(h, LZ_ERR_ARG, "lz_run_resume_partial: NULL buffer");
  if (j0 < 2 || n <= j0) return fail(h, LZ_ERR_ARG, "lz_run_resume_partial: need 2 <= j0 < n (j0 completed steps, n in total)");
  if (n > h->Mg) return fail(h, LZ_ERR_ARG, "lz_run_resume_partial: n cannot be larger than M");
  if (ldv_in < h->rows) return fail(h, LZ_ERR_ARG, "lz_run_resume_partial: ldv_in < rows_local");
  if (!(h->flags & LZ_FLAG_REORTH_PARTIAL) || (h->flags & LZ_FLAG_ONE_REDUCE))
    return fail(h, LZ_ERR_STATE, "lz_run_resume_partial: needs LZ_FLAG_REORTH_PARTIAL without LZ_FLAG_ONE_REDUCE (the device-decided loop, engine 7)");
  LZ_TRY(basis_alloc(h, n, j0));  // (all j0 uploaded rows cleared first: see lz_run_resume)
  if (choose_loop(h, n) != LOOP_PARTIAL_DEVICE)
    return fail(h, LZ_ERR_STATE, "lz_run_resume_partial: these options / knobs do not select the device-decided partial loop");
  h->halo_inflight_j = -1;
  h->y_n = 0;
  LZ_TRY(upload2d(h, h->d_V, (size_t)h->ldv * sizeof(double), V_rows, (size_t)ldv_in * sizeof(double), (size_t)h->rows * sizeof(double), (size_t)j0));
  LZ_TRY(upload(h, h->d_r, r_local, (size_t)h->rows * sizeof(double)));
  LZ_HIP(h, hipMemcpyAsync(h->d_alpha, alpha_in, (size_t)j0 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  LZ_HIP(h, hipMemcpyAsync(h->d_beta, beta_in, (size_t)(j0 - 1) * sizeof(double), hipMemcpyHostToDevice, h->stream));
  LZ_HIP(h, hipEventRecord(h->run_a, h->stream));
  h->host_syncs = 0;
  h->last_engine = LOOP_PARTIAL_DEVICE;
  LZ_TRY(run_loop_partial_device(h, n, j0, omega_state));
  h->r_state = 1;
  h->prof_iter = true;
  LZ_HIP(h, hipEventRecord(h->run_b, h->stream));
  LZ_HIP(h, hipMemcpyAsync(alpha_out, h->d_alpha, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  LZ_HIP(h, hipMemcpyAsync(beta_out, h->d_beta, (size_t)(n - 1) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  std::vector<int> sweep_log(omega_state_ints(n));
  LZ_HIP(h, hipMemcpyAsync(sweep_log.data(), h->d_omi, sweep_log.size() * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  LZ_HIP(h, hipStreamSynchronize(h->stream));
  h->last_misses = 0;
  h->sweep_log.assign((size_t)n, 0);  // (the steps of the first leg are not on this record)
  account_partial_device(h, n, sweep_log, &h->last_sweeps);
  for (int j = j0; j < n; ++j) h->sweep_log[(size_t)j] = sweep_log[(size_t)2 + j] != 0;
  float ms = 0.f;
  LZ_HIP(h, hipEventElapsedTime(&ms, h->run_a, h->run_b));
  h->acc.total_ms += ms;
  return breakdown_status(h, n, alpha_out, beta_out);
}

int lz_run_resume(lz_handle h, int n, int j0, const double* V_rows, int64_t ldv_in, const double* r_local, const double* alpha_in,
                  const double* beta_in, double* alpha_out, double* beta_out) {
  if (!h) return LZ_ERR_ARG;
  if (!V_rows || !r_local || !alpha_in || !beta_in || !alpha_out || !beta_out) return fail(h, LZ_ERR_ARG, "lz_run_resume: NULL buffer");
  if (j0 < 1 || n <= j0) return fail(h, LZ_ERR_ARG, "lz_run_resume: need 1 <= j0 < n (j0 completed steps, n in total)");
  if (n > h->Mg) return fail(h, LZ_ERR_ARG, "lz_run_resume: n cannot be larger than M");
  if (ldv_in < h->rows) return fail(h, LZ_ERR_ARG, "lz_run_resume: ldv_in < rows_local");
  if (h->flags & (LZ_FLAG_REORTH_PARTIAL | LZ_FLAG_ONE_REDUCE))
    return fail(h, LZ_ERR_STATE, "lz_run_resume: not with partial re-orthogonalisation (lz_run_resume_partial continues the device-decided loop from its omega state) or the one-reduce loop");
  // The j0 rows about to be uploaded are cleared whole: the upload writes `rows` columns of each, the kernels stream rows_pad of them,
  // and a recycled allocation's padding would enter ||r||^2 (a run's own kernels write the padding of every row they produce: zeros).
  LZ_TRY(basis_alloc(h, n, j0));
  h->halo_inflight_j = -1;
  h->y_n = 0;
  LZ_TRY(upload2d(h, h->d_V, (size_t)h->ldv * sizeof(double), V_rows, (size_t)ldv_in * sizeof(double), (size_t)h->rows * sizeof(double), (size_t)j0));
  LZ_TRY(upload(h, h->d_r, r_local, (size_t)h->rows * sizeof(double)));
  LZ_HIP(h, hipMemcpyAsync(h->d_alpha, alpha_in, (size_t)j0 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (j0 > 1) LZ_HIP(h, hipMemcpyAsync(h->d_beta, beta_in, (size_t)(j0 - 1) * sizeof(double), hipMemcpyHostToDevice, h->stream));
  LZ_HIP(h, hipEventRecord(h->run_a, h->stream));
  int sweeps = n - j0;
  h->last_engine = LOOP_SIX;  // every loop structure gives the same bits (tests/test_gpu_small.py): the plain one takes a start step
  LZ_TRY(run_loop_six(h, n, &sweeps, j0));
  h->last_sweeps = sweeps;
  h->last_misses = 0;
  h->sweep_log.assign((size_t)n, 1);
  h->r_state = 1;
  h->prof_iter = true;
  LZ_HIP(h, hipEventRecord(h->run_b, h->stream));
  LZ_HIP(h, hipMemcpyAsync(alpha_out, h->d_alpha, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  LZ_HIP(h, hipMemcpyAsync(beta_out, h->d_beta, (size_t)(n - 1) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  LZ_HIP(h, hipStreamSynchronize(h->stream));
  float ms = 0.f;
  LZ_HIP(h, hipEventElapsedTime(&ms, h->run_a, h->run_b));
  h->acc.total_ms += ms;
  return breakdown_status(h, n, alpha_out, beta_out);
}

}  // extern "C"
