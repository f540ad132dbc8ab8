// Matrix setup of the C ABI: validation + upload of CSR / dense operators, SpMV plan (row blocks, two-phase layout), stencil
// assembly on the device, halo / all-gather plans, lz_spmv_host.
#include "lz_context.h"

using namespace lz;
using namespace lz::api;

namespace lz {
namespace api {

void build_rowblocks(const int32_t* rowptr, int64_t rows, int rows_cap, int nnz_cap, std::vector<int32_t>& blk) {
  blk.clear();
  blk.push_back(0);
  int64_t r = 0;
  while (r < rows) {
    int64_t e = r;
    const int64_t k0 = rowptr[r];
    while (e < rows && e - r < rows_cap && (int64_t)rowptr[e + 1] - k0 <= nnz_cap) ++e;
    if (e == r) e = r + 1;  // a single row longer than the LDS tile: block of its own
    blk.push_back((int32_t)e);
    r = e;
  }
}

// shared tail of lz_set_csr / lz_build_stencil3d: row blocks for the CSR-stream kernel + bookkeeping
int fill_csr_meta(lz_handle h, CsrDev& A, const int32_t* rowptr_host, int64_t rows_local, int64_t ncols_ext, int64_t nnz, int fixed_k,
                  int max_nnz);

int finish_csr(lz_handle h, const int32_t* rowptr_host, int64_t M_global, int64_t row0, int64_t rows_local, int64_t ncols_ext,
               int64_t nnz, int fixed_k, int max_nnz) {
  LZ_TRY(fill_csr_meta(h, h->csr, rowptr_host, rows_local, ncols_ext, nnz, fixed_k, max_nnz));
  h->T_declared = false;  // a new H invalidates H^T and the two-sided bases
  h->has_T = false;
  h->bi_n = 0;
  h->Mg = M_global;
  h->row0 = row0;
  h->rows = rows_local;
  h->ncols_ext = ncols_ext;
  h->rows_pad = round_up(rows_local, kPadDoubles);
  h->ldv = skew_stride(h, h->rows_pad);
  h->xmode = 0;
  h->kind = 1;
  return LZ_OK;
}

int fill_csr_meta(lz_handle h, CsrDev& A, const int32_t* rowptr_host, int64_t rows_local, int64_t ncols_ext, int64_t nnz, int fixed_k,
                  int max_nnz) {
  std::vector<int32_t> blk;
  int rows_cap = h->tune[2] > 0 ? h->tune[2] : 512;
  // One batch of the CSR-stream kernel covers 2048 entries (256 lanes x 4 steps x 2): for long rows (27-point
  // stencils) a tile of exactly one batch is fastest (profiles/r01/ab_spmv_27pt_tile.json); short ragged rows keep 4096.
  int nnz_cap = h->tune[4] > 0 ? h->tune[4] : ((double)nnz / (double)rows_local >= 12.0 ? 2048 : 4096);
  if (nnz_cap > 16384) nnz_cap = 16384;
  build_rowblocks(rowptr_host, rows_local, rows_cap, nnz_cap, blk);
  A.blk_nnz_cap = nnz_cap;
  A.fixed_rb = h->tune[5] > 0 ? h->tune[5] : 512;
  LZ_TRY(dev_alloc(h, A.rowblk, blk.size()));
  LZ_HIP(h, hipMemcpy(A.rowblk, blk.data(), blk.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  A.n_rowblk = (int)blk.size() - 1;
  A.rows = rows_local;
  A.ncols = ncols_ext;
  A.nnz = nnz;
  A.fixed_k = fixed_k;
  A.max_row_nnz = max_nnz;
  A.avg_row_nnz = (double)nnz / (double)rows_local;
  // Matrices without column locality (random graphs): the SpMV is bound by cache-missing 8-byte gathers, so it runs as
  // the column-blocked two-phase kernel pair that gathers out of LDS only (lz_spmv_pb.hip; 2.5x on config C3, same
  // bits).  Auto: the vector is larger than the L2s can hold (>= 2^20 columns), rows are not a fixed-K stencil, and more
  // than a quarter of the entries sit further than 2^18 columns from the diagonal.  tune[14]: 1 = never, 2 = always
  // (tests run it on small matrices).
  pb_free(A.pb);
  // Fixed-K rows (stencils): the ELL-ordered second copy (lz_spmv.hip, k_spmv_ell) - lanes own whole rows, coalesced loads
  // and gathers, no LDS staging.  Measured against the CSR-order kernel k_spmv_fixed on the headline, C2 and two 3-D grids
  // (profiles/r04/ab_spmv_ell.json): the same time to within 2 % either way - both sit at the rate this part streams a
  // 90 % read / 10 % write mix - so the plain SpMV keeps the CSR-order kernel and no second copy is made.  The ELL copy is what
  // the device-resident partial re-orthogonalisation loop needs for its fused r / beta (a lane owns whole rows): that loop
  // builds it on first use.  tune[17]: 0 auto (as just said), 1 never (not even for the partial loop), 2 ELL for every SpMV,
  // one row per lane and trip, 3 ELL with two adjacent rows per lane.
  ell_free(A);
  // 5 / 7 entries per row: the plain SpMV takes the ELL copy only on request (measured equal to the CSR-order kernel).  27-point
  // rows have no CSR-order fixed-K kernel - they ran the generic CSR-stream kernel - and there ELL wins: 247.9 vs 256.3 us on the
  // reference's largest run (deuteron N = 160: 0.710 vs 0.687; the partial loop 223 vs 232 ms with the fused r / beta), so it is
  // their default (gpurun r4q; +12 bytes per entry of device memory).
  // Round 5, row-class coding (k_spmv_ell, CODED): where the rows of a fixed-K matrix fall into <= 256 classes up to translation - any
  // stencil on a regular grid - the matrix stream of the ELL-order kernel is one byte per row (+ the values unless the coefficients are
  // constant too), found and verified on the device right here (~1 ms at 1e7 rows).  Such a copy costs (almost) no memory and moves
  // 0.21 (0.76 with streamed values) of the CSR bytes: it is every SpMV's default.  tune[17]: 2 / 3 keep the uncoded copy (A/B), 4 codes
  // the offsets only.
  A.ell_default = h->tune[17] >= 2 || (h->tune[17] == 0 && fixed_k == 27);
  const double t_ell = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
  if (fixed_k == 5 || fixed_k == 7 || fixed_k == 27) {
    if (h->tune[17] == 0 || h->tune[17] == 4) {
      LZ_HIP(h, ell_build(A, 0, h->stream, h->tune[17] == 4 ? 2 : 1, A.ell_default));
      if (A.ell_coded) A.ell_default = true;
      A.cls_group = h->tune[23];  // 0: two adjacent rows per lane; 1 / 3: one row per lane and trip, one / two units per workgroup (A/B)
    } else if (A.ell_default) {
      LZ_HIP(h, ell_build(A, h->tune[17] == 3 ? 1 : 0, h->stream));
    }
    if (getenv("LZ_DEBUG_TIMING"))
      fprintf(stderr, "[lz_set_csr] row classes / ELL copy: %.3f ms (coding %d, %d classes)\n",
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() - t_ell, A.ell_coded, A.ell_ncls);
  }
  const bool want = h->tune[14] == 2 || (h->tune[14] == 0 && fixed_k == 0 && ncols_ext >= ((int64_t)1 << 20) && A.far_frac > 0.25);
  if (want) {
    const hipError_t pe = pb_build(A, rowptr_host, &A.pb, h->stream, h->tune[10], h->tune[22]);
    A.host_colidx = nullptr;  // (the caller's arrays are only valid during this call)
    A.host_vals = nullptr;
    LZ_HIP(h, pe);
  }
  A.host_colidx = nullptr;
  A.host_vals = nullptr;
  return LZ_OK;
}

// validate + upload one CSR matrix into A (arrays padded by 2 entries: the kernels read pairs)
int upload_csr(lz_handle h, CsrDev& A, const char* who, int64_t rows, int64_t ncols, int64_t nnz, const int32_t* rowptr,
               const int32_t* colidx, const double* vals, int* fixed_k_out, int* max_nnz_out) {
  if (rowptr[0] != 0 || rowptr[rows] != nnz) return fail(h, LZ_ERR_ARG, std::string(who) + ": rowptr[0] != 0 or rowptr[rows] != nnz");
  const bool dbg = getenv("LZ_DEBUG_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  // one sweep over the rows and their entries, split over host threads (round 3: two single-thread loops, 0.35 s at the
  // headline's 5e7 entries): row lengths monotone, their maximum, whether all are equal, column range, share of far entries
  struct Part {
    int max_nnz = 0;
    bool same = true, bad_ptr = false, bad_col = false;
    int64_t far = 0;
  } parts[kMaxHostThreads];
  const int64_t k_first = rows > 0 ? (int64_t)rowptr[1] - rowptr[0] : 0;
  parallel_ranges(rows, 1 << 16, [&](int t, int64_t lo, int64_t hi) {
    Part p;  // a local: the threads' slots of `parts` share cache lines
    struct Publish {
      Part& dst;
      const Part& src;
      ~Publish() { dst = src; }
    } publish{parts[t], p};
    for (int64_t i = lo; i < hi; ++i) {
      const int64_t a = rowptr[i], b = rowptr[i + 1], d = b - a;
      if (d < 0 || a < 0 || b > nnz) {
        p.bad_ptr = true;
        return;
      }
      if (d > p.max_nnz) p.max_nnz = (int)d;
      if (d != k_first) p.same = false;
      for (int64_t k = a; k < b; ++k) {
        const int64_t c = colidx[k];
        if (c < 0 || c >= ncols) {
          p.bad_col = true;
          return;
        }
        p.far += (c > i ? c - i : i - c) > ((int64_t)1 << 18);
      }
    }
  });
  int max_nnz = 0;
  int fixed_k = (int)k_first;
  int64_t far = 0;
  for (const Part& p : parts) {
    if (p.bad_ptr) return fail(h, LZ_ERR_ARG, std::string(who) + ": rowptr not monotone");
    if (p.bad_col) return fail(h, LZ_ERR_ARG, std::string(who) + ": column index out of range");
    if (p.max_nnz > max_nnz) max_nnz = p.max_nnz;
    if (!p.same) fixed_k = 0;
    far += p.far;
  }
  A.far_frac = nnz > 0 ? (double)far / (double)nnz : 0.0;
  const double t1 = now();
  pb_free(A.pb);
  ell_free(A);
  if (fixed_k > 64) fixed_k = 0;
  LZ_TRY(dev_alloc(h, A.rowptr, (size_t)rows + 1));
  LZ_TRY(dev_alloc(h, A.colidx, (size_t)nnz + 2));
  LZ_TRY(dev_alloc(h, A.vals, (size_t)nnz + 2));
  LZ_HIP(h, hipMemsetAsync(A.colidx + nnz, 0, 2 * sizeof(int32_t), h->stream));  // (the kernels read pairs: two pad entries)
  LZ_HIP(h, hipMemsetAsync(A.vals + nnz, 0, 2 * sizeof(double), h->stream));
  LZ_TRY(upload(h, A.rowptr, rowptr, ((size_t)rows + 1) * sizeof(int32_t)));
  if (nnz > 0) {
    LZ_TRY(upload(h, A.colidx, colidx, (size_t)nnz * sizeof(int32_t)));
    LZ_TRY(upload(h, A.vals, vals, (size_t)nnz * sizeof(double)));
  }
  const double t2 = now();
  LZ_HIP(h, hipStreamSynchronize(h->stream));
  if (dbg)
    fprintf(stderr, "[%s] validation sweep %.3f ms, device alloc %.3f ms, H2D of %.1f MB %.3f ms\n", who, t1 - t0, t2 - t1,
            (12.0 * nnz + 4.0 * rows) / 1e6, now() - t2);
  *fixed_k_out = fixed_k;
  *max_nnz_out = max_nnz;
  A.host_colidx = colidx;  // for pb_build (fill_csr_meta, same API call): diagonal split, fp32-exact value check
  A.host_vals = vals;
  return LZ_OK;
}

}  // namespace api
}  // namespace lz

extern "C" {

// ---- matrix ----------------------------------------------------------------
int lz_set_csr(lz_handle h, int64_t M_global, int64_t row0, int64_t rows_local, int64_t ncols_ext, int64_t nnz,
               const int32_t* rowptr, const int32_t* colidx, const double* vals) {
  if (!h) return LZ_ERR_ARG;
  if (M_global <= 0 || rows_local <= 0 || row0 < 0 || row0 + rows_local > M_global || nnz < 0 || !rowptr ||
      (nnz > 0 && (!colidx || !vals)))
    return fail(h, LZ_ERR_ARG, "lz_set_csr: bad sizes or NULL arrays");
  if (rows_local >= (int64_t)1 << 31 || nnz >= (int64_t)1 << 31 || ncols_ext >= (int64_t)1 << 31)
    return fail(h, LZ_ERR_ARG, "lz_set_csr: sizes exceed int32 CSR indexing");
  if (ncols_ext < rows_local) return fail(h, LZ_ERR_ARG, "lz_set_csr: ncols_ext < rows_local");
  LZ_HIP(h, hipSetDevice(h->dev));
  LZ_HIP(h, hipStreamSynchronize(h->stream));
  h->kind = 0;
  LZ_TRY(dev_free(h, h->d_dense));
  LZ_TRY(dev_free(h, h->d_V));  // a new matrix invalidates the basis
  h->n = 0;
  int fixed_k = 0, max_nnz = 0;
  LZ_TRY(upload_csr(h, h->csr, "lz_set_csr", rows_local, ncols_ext, nnz, rowptr, colidx, vals, &fixed_k, &max_nnz));
  return finish_csr(h, rowptr, M_global, row0, rows_local, ncols_ext, nnz, fixed_k, max_nnz);
}

int lz_build_stencil3d_block(lz_handle h, int Nx, int Ny, int Nz, int points, double T_factor, const double* weights4,
                             int potential_kind, const double* potential, int negate_T, int64_t row0, int64_t rows_local,
                             int nranges, const int64_t* ghost_start, const int64_t* ghost_len) {
  if (!h) return LZ_ERR_ARG;
  if (Nx < 3 || Ny < 3 || Nz < 3 || (points != 7 && points != 27) || !weights4)
    return fail(h, LZ_ERR_ARG, "lz_build_stencil3d_block: need Nx, Ny, Nz >= 3, points in {7, 27}, 4 weights");
  const int64_t M = (int64_t)Nx * Ny * Nz;
  if (row0 < 0 || rows_local <= 0 || row0 + rows_local > M) return fail(h, LZ_ERR_ARG, "lz_build_stencil3d_block: bad row block");
  if (potential_kind < 0 || potential_kind > 2 || (potential_kind != 0 && !potential))
    return fail(h, LZ_ERR_ARG, "lz_build_stencil3d_block: potential_kind in {0, 1, 2}; kinds 1 and 2 need the array / the 8 parameters");
  if (nranges < 0 || nranges > 16 || (nranges > 0 && (!ghost_start || !ghost_len)))
    return fail(h, LZ_ERR_ARG, "lz_build_stencil3d_block: at most 16 ghost ranges");
  const bool whole = rows_local == M;
  if (!whole && h->world == 1) return fail(h, LZ_ERR_STATE, "lz_build_stencil3d_block: a row block needs a multi-rank handle (lz_comm_init_*)");
  const int64_t nnz = rows_local * points;
  if (nnz >= (int64_t)1 << 31 || M >= (int64_t)1 << 31) return fail(h, LZ_ERR_ARG, "lz_build_stencil3d_block: sizes exceed int32 CSR indexing");
  StencilArgs a;
  memset(&a, 0, sizeof a);
  a.Nx = Nx;
  a.Ny = Ny;
  a.Nz = Nz;
  a.negate = negate_T;
  a.pot_kind = potential_kind;
  a.renumber = whole ? 0 : 1;
  a.nranges = nranges;
  a.row0 = row0;
  a.rows_local = rows_local;
  a.tf = T_factor;
  for (int q = 0; q < 4; ++q) a.w[q] = weights4[q];
  if (potential_kind == 2)
    for (int q = 0; q < 8; ++q) a.par[q] = potential[q];
  const int64_t rows_pad = round_up(rows_local, kPadDoubles);
  int64_t ext = rows_pad, nghost = 0;
  for (int q = 0; q < nranges; ++q) {
    if (ghost_len[q] <= 0 || ghost_start[q] < 0 || ghost_start[q] + ghost_len[q] > M ||
        (ghost_start[q] < row0 + rows_local && ghost_start[q] + ghost_len[q] > row0))
      return fail(h, LZ_ERR_ARG, "lz_build_stencil3d_block: ghost range outside the grid or overlapping the owned rows");
    a.gstart[q] = ghost_start[q];
    a.glen[q] = ghost_len[q];
    a.gext[q] = ext;
    ext += ghost_len[q];
    nghost += ghost_len[q];
  }
  LZ_HIP(h, hipSetDevice(h->dev));
  LZ_HIP(h, hipStreamSynchronize(h->stream));
  h->kind = 0;
  LZ_TRY(dev_free(h, h->d_dense));
  LZ_TRY(dev_free(h, h->d_V));
  h->n = 0;
  CsrDev& A = h->csr;
  pb_free(A.pb);
  LZ_TRY(dev_alloc(h, A.rowptr, (size_t)rows_local + 1));
  LZ_TRY(dev_alloc(h, A.colidx, (size_t)nnz + 2));
  LZ_TRY(dev_alloc(h, A.vals, (size_t)nnz + 2));
  LZ_HIP(h, hipMemsetAsync(A.colidx + nnz, 0, 2 * sizeof(int32_t), h->stream));
  LZ_HIP(h, hipMemsetAsync(A.vals + nnz, 0, 2 * sizeof(double), h->stream));
  double* dpot = nullptr;
  if (potential_kind == 1) {
    LZ_TRY(dev_alloc(h, dpot, (size_t)rows_local));
    LZ_HIP(h, hipMemcpyAsync(dpot, potential, (size_t)rows_local * sizeof(double), hipMemcpyHostToDevice, h->stream));
  }
  launch_build_stencil3d(a, points, dpot, A.rowptr, A.colidx, A.vals, h->stream);
  int rc = check_launch(h, "build_stencil3d");
  hipError_t e = hipStreamSynchronize(h->stream);
  if (dpot) big_free(dpot);
  if (rc != LZ_OK) return rc;
  if (e != hipSuccess) return fail(h, LZ_ERR_HIP, std::string("lz_build_stencil3d_block: ") + hipGetErrorString(e));
  std::vector<int32_t> rowptr((size_t)rows_local + 1);
  for (int64_t i = 0; i <= rows_local; ++i) rowptr[(size_t)i] = (int32_t)(i * points);
  A.far_frac = 0.0;
  return finish_csr(h, rowptr.data(), M, row0, rows_local, whole ? M : rows_pad + nghost, nnz, points, points);
}

int lz_build_stencil3d(lz_handle h, int N, int points, double T_factor, const double* weights4, const double* potential,
                       int negate_T) {
  if (!h) return LZ_ERR_ARG;
  if (h->world > 1) return fail(h, LZ_ERR_STATE, "lz_build_stencil3d: whole matrix on one rank; use lz_build_stencil3d_block for a row partition");
  if (N < 3) return fail(h, LZ_ERR_ARG, "lz_build_stencil3d: need N >= 3, points in {7, 27}, 4 weights");
  return lz_build_stencil3d_block(h, N, N, N, points, T_factor, weights4, potential ? 1 : 0, potential, negate_T, 0, (int64_t)N * N * N, 0,
                                  nullptr, nullptr);
}

int lz_csr_info(lz_handle h, int64_t* rows, int64_t* nnz) {
  if (!h || !rows || !nnz) return LZ_ERR_ARG;
  if (h->kind != 1) return fail(h, LZ_ERR_STATE, "lz_csr_info: no CSR matrix set");
  *rows = h->csr.rows;
  *nnz = h->csr.nnz;
  return LZ_OK;
}

int lz_spmv_plan(lz_handle h, int* plan) {
  if (!h || !plan) return LZ_ERR_ARG;
  if (h->kind == 0) return fail(h, LZ_ERR_STATE, "lz_spmv_plan: no matrix set");
  if (h->kind == 2) {
    *plan = 4;
    return LZ_OK;
  }
  const CsrDev& A = h->csr;
  if (h->flags & LZ_FLAG_SPMV_SCALAR) *plan = 0;
  else if (A.pb && !(h->flags & LZ_FLAG_SPMV_STREAM)) *plan = 3;
  else if (!(h->flags & LZ_FLAG_SPMV_STREAM) && (A.fixed_k == 5 || A.fixed_k == 7 || (A.ell_default && ell_usable(A, h->flags)))) *plan = 2;
  else *plan = 1;
  return LZ_OK;
}

int lz_spmv_coding(lz_handle h, int* coding, int* classes) {
  if (!h || !coding || !classes) return LZ_ERR_ARG;
  if (h->kind == 0) return fail(h, LZ_ERR_STATE, "lz_spmv_coding: no matrix set");
  const CsrDev& A = h->csr;
  const bool used = h->kind == 1 && A.ell_coded && ell_usable(A, h->flags);
  *coding = used ? A.ell_coded : 0;
  *classes = used ? A.ell_ncls : 0;
  return LZ_OK;
}

int lz_get_csr(lz_handle h, int32_t* rowptr, int32_t* colidx, double* vals) {
  if (!h || !rowptr || !colidx || !vals) return LZ_ERR_ARG;
  if (h->kind != 1) return fail(h, LZ_ERR_STATE, "lz_get_csr: no CSR matrix set");
  LZ_HIP(h, hipSetDevice(h->dev));
  const CsrDev& A = h->csr;
  LZ_HIP(h, hipMemcpy(rowptr, A.rowptr, ((size_t)A.rows + 1) * sizeof(int32_t), hipMemcpyDeviceToHost));
  LZ_HIP(h, hipMemcpy(colidx, A.colidx, (size_t)A.nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
  LZ_HIP(h, hipMemcpy(vals, A.vals, (size_t)A.nnz * sizeof(double), hipMemcpyDeviceToHost));
  return LZ_OK;
}

int lz_set_dense_block(lz_handle h, int64_t M_global, int64_t row0, int64_t rows_local, int64_t ncols_ext, const double* A) {
  if (!h) return LZ_ERR_ARG;
  if (M_global <= 0 || rows_local <= 0 || row0 < 0 || row0 + rows_local > M_global || ncols_ext < M_global || !A)
    return fail(h, LZ_ERR_ARG, "lz_set_dense_block: bad sizes or NULL matrix");
  if (h->world == 1 && (rows_local != M_global || ncols_ext != M_global))
    return fail(h, LZ_ERR_ARG, "lz_set_dense_block: a single rank owns the whole square matrix");
  LZ_HIP(h, hipSetDevice(h->dev));
  LZ_HIP(h, hipStreamSynchronize(h->stream));
  h->kind = 0;
  LZ_TRY(dev_free(h, h->d_V));
  h->n = 0;
  const int64_t lda = (ncols_ext + 1) & ~(int64_t)1;
  LZ_TRY(dev_alloc(h, h->d_dense, (size_t)rows_local * lda + 2));
  if (lda != ncols_ext) LZ_HIP(h, hipMemsetAsync(h->d_dense, 0, ((size_t)rows_local * lda + 2) * sizeof(double), h->stream));
  LZ_TRY(upload2d(h, h->d_dense, (size_t)lda * sizeof(double), A, (size_t)ncols_ext * sizeof(double), (size_t)ncols_ext * sizeof(double),
                  (size_t)rows_local));
  LZ_HIP(h, hipStreamSynchronize(h->stream));
  h->dense_lda = lda;
  h->Mg = M_global;
  h->row0 = row0;
  h->rows = rows_local;
  h->ncols_ext = ncols_ext;
  h->rows_pad = round_up(rows_local, kPadDoubles);
  h->ldv = skew_stride(h, h->rows_pad);
  h->xmode = 0;
  h->kind = 2;
  h->T_declared = false;
  h->bi_n = 0;
  return LZ_OK;
}

int lz_set_dense(lz_handle h, int64_t M, const double* A) {
  if (!h) return LZ_ERR_ARG;
  if (h->world > 1) return fail(h, LZ_ERR_STATE, "lz_set_dense: one rank, whole matrix; use lz_set_dense_block + lz_set_allgather for a row partition");
  return lz_set_dense_block(h, M, 0, M, M, A);
}

int lz_set_halo(lz_handle h, int npeers, const int32_t* peers, const int64_t* send_counts, const int32_t* send_idx,
                const int64_t* recv_counts) {
  if (!h) return LZ_ERR_ARG;
  if (h->kind != 1) return fail(h, LZ_ERR_STATE, "lz_set_halo: call lz_set_csr first");
  if (npeers < 0 || (npeers > 0 && (!peers || !send_counts || !recv_counts))) return fail(h, LZ_ERR_ARG, "lz_set_halo: NULL arrays");
  h->peers.assign(peers, peers + npeers);
  h->scount.assign(send_counts, send_counts + npeers);
  h->rcount.assign(recv_counts, recv_counts + npeers);
  h->soff.assign(npeers, 0);
  h->roff.assign(npeers, 0);
  int64_t ts = 0, tr = 0;
  for (int p = 0; p < npeers; ++p) {
    const bool self_ok = h->tune[6] != 0;  // test knob: a rank may exchange with itself (1-rank RCCL send/recv test)
    if (peers[p] < 0 || peers[p] >= h->world || (peers[p] == h->rank && !self_ok) || send_counts[p] < 0 || recv_counts[p] < 0)
      return fail(h, LZ_ERR_ARG, "lz_set_halo: bad peer or count");
    h->soff[p] = ts;
    h->roff[p] = tr;
    ts += send_counts[p];
    tr += recv_counts[p];
  }
  if (h->rows_pad + tr != h->ncols_ext)
    return fail(h, LZ_ERR_ARG, "lz_set_halo: ncols_ext must equal lz_padded_rows(rows_local) + total receive count");
  for (int64_t k = 0; k < ts; ++k)
    if (!send_idx || send_idx[k] < 0 || send_idx[k] >= h->rows) return fail(h, LZ_ERR_ARG, "lz_set_halo: send index out of range");
  LZ_HIP(h, hipSetDevice(h->dev));
  LZ_TRY(dev_alloc(h, h->d_send_idx, (size_t)ts));
  LZ_TRY(dev_alloc(h, h->d_sendbuf, (size_t)ts));
  if (ts > 0) LZ_HIP(h, hipMemcpy(h->d_send_idx, send_idx, (size_t)ts * sizeof(int32_t), hipMemcpyHostToDevice));
  h->sstart.assign(npeers, -1);
  h->all_contig = npeers > 0;
  for (int p = 0; p < npeers; ++p) {
    bool contig = true;
    for (int64_t k = 1; k < send_counts[p]; ++k)
      if (send_idx[h->soff[p] + k] != send_idx[h->soff[p] + k - 1] + 1) {
        contig = false;
        break;
      }
    if (contig && send_counts[p] > 0) h->sstart[p] = send_idx[h->soff[p]];
    if (!contig) h->all_contig = false;
  }
  h->bnd_ranges.clear();
  h->int_ranges.clear();
  if (h->all_contig) {
    std::vector<std::pair<int64_t, int64_t>> rg;
    for (int p = 0; p < npeers; ++p)
      if (send_counts[p] > 0) rg.push_back({h->sstart[p] / 2, (h->sstart[p] + send_counts[p] + 1) / 2});  // double2 positions
    std::sort(rg.begin(), rg.end());
    for (auto& r : rg) {
      if (!h->bnd_ranges.empty() && r.first <= h->bnd_ranges.back().second)
        h->bnd_ranges.back().second = std::max(h->bnd_ranges.back().second, r.second);
      else
        h->bnd_ranges.push_back(r);
    }
    int64_t cur = 0;
    const int64_t n2 = h->rows_pad / 2;
    for (auto& r : h->bnd_ranges) {
      if (r.first > cur) h->int_ranges.push_back({cur, r.first});
      cur = r.second;
    }
    if (cur < n2) h->int_ranges.push_back({cur, n2});
    if (!h->cstream) {
      LZ_HIP(h, hipStreamCreateWithFlags(&h->cstream, hipStreamNonBlocking));
      LZ_HIP(h, hipEventCreateWithFlags(&h->e_bnd, hipEventDisableTiming));
      LZ_HIP(h, hipEventCreateWithFlags(&h->e_halo, hipEventDisableTiming));
    }
  }
  h->total_send = ts;
  h->total_recv = tr;
  h->ldv = skew_stride(h, h->rows_pad + round_up(tr, kPadDoubles));
  h->xmode = 1;
  LZ_TRY(dev_free(h, h->d_V));
  h->n = 0;
  return LZ_OK;
}

int lz_set_allgather(lz_handle h, int64_t chunk) {
  if (!h) return LZ_ERR_ARG;
  if (h->kind == 0) return fail(h, LZ_ERR_STATE, "lz_set_allgather: call lz_set_csr / lz_set_dense_block first");
  if (chunk < h->rows_pad || chunk % kPadDoubles != 0 || chunk * h->world != h->ncols_ext)
    return fail(h, LZ_ERR_ARG, "lz_set_allgather: chunk must be a multiple of 32, >= padded rows, and world*chunk == ncols_ext");
  LZ_HIP(h, hipSetDevice(h->dev));
  LZ_TRY(dev_alloc(h, h->d_xfull, (size_t)(chunk * h->world)));
  LZ_HIP(h, hipMemset(h->d_xfull, 0, (size_t)(chunk * h->world) * sizeof(double)));
  h->ag_chunk = chunk;
  h->ldv = skew_stride(h, chunk);
  h->xmode = 2;
  LZ_TRY(dev_free(h, h->d_V));
  h->n = 0;
  return LZ_OK;
}

int lz_spmv_host(lz_handle h, const double* x, double* y) {
  if (!h || !x || !y) return LZ_ERR_ARG;
  if (h->kind == 0) return fail(h, LZ_ERR_STATE, "no matrix set");
  if (h->world > 1) return fail(h, LZ_ERR_STATE, "lz_spmv_host is single-rank only");
  LZ_HIP(h, hipSetDevice(h->dev));
  const size_t nx = (size_t)round_up(h->ncols_ext, kPadDoubles) + 2 * (size_t)h->rows_pad;
  LZ_TRY(dev_alloc(h, h->d_xtmp, nx));
  double* dx = h->d_xtmp;
  double* dy = h->d_xtmp + round_up(h->ncols_ext, kPadDoubles);
  LZ_TRY(ensure_part(h, std::max<size_t>((size_t)h->csr.n_rowblk + 64, (size_t)(h->rows / 4 + 64))));  // >= the two-phase kernel's row blocks
  LZ_HIP(h, hipMemcpyAsync(dx, x, (size_t)h->ncols_ext * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (h->kind == 1)
    launch_spmv_csr(h->csr, dx, dy, dx, h->d_part, h->flags, h->stream);
  else
    launch_gemv_dense(h->d_dense, h->rows, h->ncols_ext, h->dense_lda, dx, dx, dy, h->d_part, h->stream);
  LZ_TRY(check_launch(h, "spmv"));
  LZ_HIP(h, hipMemcpyAsync(y, dy, (size_t)h->rows * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  LZ_HIP(h, hipStreamSynchronize(h->stream));
  return LZ_OK;
}

}  // extern "C"
