// Result side of the C ABI: the basis (lz_get_basis*), the Ritz back-transform (resident or chunked), the device Gram matrix and
// quality sums behind get_H_eigs / print_good_eigs.  Kernels: lz_gemm.hip.
#include "lz_context.h"

using namespace lz;
using namespace lz::api;

extern "C" {

int lz_get_basis(lz_handle h, double* V_out, int64_t ld) {
  if (!h || !V_out) return LZ_ERR_ARG;
  LZ_TRY(require_basis(h, 0));
  if (ld < h->rows) return fail(h, LZ_ERR_ARG, "lz_get_basis: ld < rows_local");
  LZ_HIP(h, hipSetDevice(h->dev));
  LZ_HIP(h, xfer_d2h(h->dev, h->stream, h->xfer, V_out, (size_t)ld * sizeof(double), h->d_V, (size_t)h->ldv * sizeof(double),
                     (size_t)h->rows * sizeof(double), (size_t)h->n));
  return LZ_OK;
}

int lz_get_basis_block(lz_handle h, int64_t row0, int64_t nrows, double* V_out, int64_t ld) {
  if (!h || !V_out) return LZ_ERR_ARG;
  LZ_TRY(require_basis(h, 0));
  if (row0 < 0 || nrows < 1 || row0 + nrows > h->rows || ld < nrows) return fail(h, LZ_ERR_ARG, "lz_get_basis_block: bad row range or ld < nrows");
  LZ_HIP(h, hipSetDevice(h->dev));
  LZ_HIP(h, xfer_d2h(h->dev, h->stream, h->xfer, V_out, (size_t)ld * sizeof(double), h->d_V + row0, (size_t)h->ldv * sizeof(double),
                     (size_t)nrows * sizeof(double), (size_t)h->n));
  return LZ_OK;
}

}  // extern "C"

namespace {

// ---- Ritz back-transform: resident or chunked -----------------------------------------------------------------------
// rows [r0, r0 + nr) of Y = V^T-layout x S into `dst` (row-major, leading dimension n).  r0 is a multiple of 16 (the
// S-stationary kernel moves whole 16-row tiles of V with 16-byte LDS-DMA pieces); dst needs round_up(nr, 16) + 16 rows.
int ritz_rows_into(lz_handle h, int64_t r0, int64_t nr, double* dst) {
  const int n = h->y_n;
  Scope sc(h, LZ_K_RITZ, 16.0 * n * (double)nr + 8.0 * n * n, 2.0 * (double)nr * n * n);
  LZ_HIP(h, launch_ritz_gemm(h->d_V + r0, h->ldv, nr, n, h->d_S, h->s_npad, dst, n, h->stream, h->tune[9],
                             reinterpret_cast<unsigned long long*>(h->d_rclk)));
  return check_launch(h, "ritz_gemm");
}

// columns [c0, c0 + nc) of Y for ALL rows into `dst` (rows x ldy): the chunked mode's way to hand whole Ritz vectors to
// the quality sums (A y_i needs every row of y_i) without ever holding all n of them
int ritz_cols_into(lz_handle h, int c0, int nc, double* dst, int64_t ldy) {
  const int n = h->y_n;
  Scope sc(h, LZ_K_RITZ, 8.0 * n * (double)h->y_rows + 8.0 * nc * (double)h->y_rows, 2.0 * (double)h->y_rows * n * nc);
  launch_ritz_gemm_cols(h->d_V, h->ldv, h->y_rows, n, h->d_S + c0, h->s_npad, nc, dst, ldy, h->stream);
  return check_launch(h, "ritz_gemm(columns)");
}

}  // namespace

extern "C" {

int lz_ritz_vectors(lz_handle h, const double* S, double* Y_out) {
  if (!h || !S) return LZ_ERR_ARG;
  LZ_TRY(require_basis(h, 0));
  LZ_HIP(h, hipSetDevice(h->dev));
  const int n = h->n;
  const int npad = (int)round_up(n, 16);
  std::vector<double> Sp((size_t)npad * npad, 0.0);
  for (int k = 0; k < n; ++k) memcpy(&Sp[(size_t)k * npad], S + (size_t)k * n, (size_t)n * sizeof(double));
  if (!h->d_S || h->s_npad != npad) {
    LZ_TRY(dev_alloc(h, h->d_S, Sp.size() + 64));
    h->s_npad = npad;
  }
  if (!h->d_rclk) LZ_TRY(dev_alloc(h, h->d_rclk, 8 + 256));
  LZ_HIP(h, hipMemsetAsync(h->d_rclk, 0, (8 + 256) * sizeof(uint64_t), h->stream));
  LZ_HIP(h, hipMemsetAsync(h->d_rclk + 4, 0xFF, sizeof(uint64_t), h->stream));  // [4], [6]: min over waves of their entry tick
  LZ_HIP(h, hipMemsetAsync(h->d_rclk + 6, 0xFF, sizeof(uint64_t), h->stream));
  LZ_HIP(h, hipMemcpyAsync(h->d_S, Sp.data(), Sp.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  LZ_HIP(h, hipStreamSynchronize(h->stream));  // Sp is a local
  // Resident when all of Y fits beside the basis (with 1 GiB to spare for the Gram partials and the runtime), else chunked.
  // tune[16] > 0 forces the chunked mode with that many rows per chunk (tests).
  const size_t full = y_doubles(h->rows, n);
  bool chunked = h->tune[16] > 0;
  size_t free_b = 0, total_b = 0;
  {
    std::lock_guard<std::mutex> lk(h->res_mu);
    if (h->res_Y && !chunked && h->res_Y_count >= full && !(h->d_Y && !h->y_chunked && h->y_cap >= (int64_t)full)) {
      big_free(h->d_Y);  // (a smaller or chunked buffer of an earlier call)
      h->d_Y = h->res_Y;
      h->y_cap = (int64_t)h->res_Y_count;
      h->y_chunked = false;
      h->res_Y = nullptr;
      h->res_Y_count = 0;
    }
  }
  if (!chunked && !(h->d_Y && !h->y_chunked && h->y_cap >= (int64_t)full)) {
    LZ_TRY(dev_free(h, h->d_Y));
    h->y_cap = 0;
    LZ_HIP(h, hipMemGetInfo(&free_b, &total_b));
    // what has to stay free beside Y: lz_ritz_gram's scratch (K-slice partials + G; 0.25 GB at n = 200, 4 GB at n = 1000 on the
    // split-K path) plus 512 MB for the runtime and the quality sums
    const size_t gram_need = (std::max<size_t>((gram_scratch_doubles(n) + (size_t)n * n - 1) / ((size_t)n * n), 512) + 2) * (size_t)n * n * sizeof(double);
    chunked = full * sizeof(double) + gram_need + ((size_t)512 << 20) > free_b;
  }
  h->y_rows = h->rows;
  h->y_n = n;
  if (!chunked) {
    if (!h->d_Y || h->y_cap < (int64_t)full) {
      LZ_TRY(dev_alloc(h, h->d_Y, full));
      h->y_cap = (int64_t)full;
    }
    h->y_chunked = false;
    h->y_chunk = h->rows;
    LZ_TRY(ritz_rows_into(h, 0, h->rows, h->d_Y));
    if (Y_out) {
      const size_t bytes = (size_t)h->rows * n * sizeof(double);
      LZ_HIP(h, xfer_d2h(h->dev, h->stream, h->xfer, Y_out, bytes, h->d_Y, bytes, bytes, 1));
    } else {
      LZ_HIP(h, hipStreamSynchronize(h->stream));
    }
    return LZ_OK;
  }
  // chunked: a bounded buffer (at most 4 GiB, at most a quarter of what is free), whole 16-row tiles
  int64_t chunk = h->tune[16] > 0 ? h->tune[16] : 0;
  if (chunk == 0) {
    LZ_TRY(dev_free(h, h->d_Y));
    h->y_cap = 0;
    LZ_HIP(h, hipMemGetInfo(&free_b, &total_b));
    const size_t budget = std::min<size_t>((size_t)4 << 30, free_b / 4);
    chunk = (int64_t)(budget / ((size_t)n * sizeof(double)));
    if (chunk < 4096) return fail(h, LZ_ERR_NOMEM, "lz_ritz_vectors: no device memory left for even a 4096-row chunk of Ritz vectors");
  }
  chunk = std::min<int64_t>(round_up(chunk, 16), round_up(h->rows, 16));
  const size_t need = y_doubles(chunk, n);
  if (!h->d_Y || h->y_cap < (int64_t)need) {
    LZ_TRY(dev_alloc(h, h->d_Y, need));
    h->y_cap = (int64_t)need;
  }
  h->y_chunked = true;
  h->y_chunk = chunk;
  if (Y_out) return lz_get_ritz_rows(h, 0, h->rows, Y_out);
  return LZ_OK;
}

int lz_get_ritz_rows(lz_handle h, int64_t row0, int64_t nrows, double* Y_out) {
  if (!h || !Y_out) return LZ_ERR_ARG;
  if (!h->d_Y || h->y_n < 1) return fail(h, LZ_ERR_STATE, "lz_get_ritz_rows: call lz_ritz_vectors first");
  if (row0 < 0 || nrows < 0 || row0 + nrows > h->y_rows) return fail(h, LZ_ERR_ARG, "lz_get_ritz_rows: row range outside [0, rows_local)");
  LZ_HIP(h, hipSetDevice(h->dev));
  const int n = h->y_n;
  if (!h->y_chunked) {
    const size_t bytes = (size_t)nrows * n * sizeof(double);
    LZ_HIP(h, xfer_d2h(h->dev, h->stream, h->xfer, Y_out, bytes, h->d_Y + (size_t)row0 * n, bytes, bytes, nrows > 0 ? 1 : 0));
    return LZ_OK;
  }
  if (!h->d_V || h->n != n || h->rows != h->y_rows) return fail(h, LZ_ERR_STATE, "lz_get_ritz_rows: the basis of the run is gone");
  for (int64_t r = row0 & ~(int64_t)15; r < row0 + nrows; r += h->y_chunk) {
    // only the 16-row tiles that cover the requested window are re-formed (a 32-row window of C4 used to cost a 4 GiB chunk)
    const int64_t nr = std::min<int64_t>(std::min<int64_t>(h->y_chunk, round_up(row0 + nrows - r, 16)), h->y_rows - r);
    LZ_TRY(ritz_rows_into(h, r, nr, h->d_Y));
    const int64_t a = std::max(r, row0), b = std::min(r + nr, row0 + nrows);
    const size_t bytes = (size_t)(b - a) * n * sizeof(double);
    LZ_HIP(h, xfer_d2h(h->dev, h->stream, h->xfer, Y_out + (size_t)(a - row0) * n, bytes, h->d_Y + (size_t)(a - r) * n, bytes, bytes, 1));
  }
  return LZ_OK;
}

int lz_get_ritz_vectors(lz_handle h, double* Y_out) {
  if (!h || !Y_out) return LZ_ERR_ARG;
  if (!h->d_Y || h->y_n < 1) return fail(h, LZ_ERR_STATE, "lz_get_ritz_vectors: call lz_ritz_vectors first");
  return lz_get_ritz_rows(h, 0, h->y_rows, Y_out);
}

int lz_ritz_info(lz_handle h, int64_t* chunk_rows, double* clock4) {
  if (!h) return LZ_ERR_ARG;
  if (!h->d_Y || h->y_n < 1) return fail(h, LZ_ERR_STATE, "lz_ritz_info: call lz_ritz_vectors first");
  LZ_HIP(h, hipSetDevice(h->dev));
  if (chunk_rows) *chunk_rows = h->y_chunked ? h->y_chunk : 0;
  if (clock4) {
    uint64_t c[8] = {0};
    if (h->d_rclk) {
      LZ_HIP(h, hipMemcpyAsync(c, h->d_rclk, sizeof c, hipMemcpyDeviceToHost, h->stream));
      LZ_HIP(h, hipStreamSynchronize(h->stream));
    }
    // [0] shader cycles, [1] ticks of the constant 100 MHz counter, [2] 16-row tiles, [3] MFMAs per tile and SIMD (x 64 = issue floor)
    clock4[0] = c[1] ? 100.0 * (double)c[0] / (double)c[1] : 0.0;  // shader clock in MHz while the kernel ran
    clock4[1] = c[2] ? (double)c[0] / (double)c[2] : 0.0;          // shader cycles per 16-row tile
    // S-in-LDS kernels: the waves of a SIMD are not in lockstep (the oldest wins the issue arbitration and finishes early), so
    // the honest figure is workgroup 0's whole span (first wave in .. last wave out, S staging included) at the measured clock
    if (c[7] > c[6] && c[6] != 0 && c[6] != ~0ull && c[2]) clock4[1] = (double)(c[7] - c[6]) * clock4[0] / 100.0 / (double)c[2];
    clock4[2] = (double)c[3] / 4.0 * 64.0;                         // MFMA issue floor per tile: (MFMAs per tile / 4 SIMDs) x 64 cycles; c[3] holds 4x the per-SIMD count
    clock4[3] = (double)c[2];
    if (getenv("LZ_DEBUG_TIMING") && c[5] > c[4] && c[4] != ~0ull) {
      fprintf(stderr, "[lz_ritz_info] kernel-internal span (first wave in .. last wave out) %.1f us; wave 0's tile loop %.1f us\n",
              (double)(c[5] - c[4]) * 0.01, (double)c[1] * 0.01);
      std::vector<uint64_t> wg(256);
      if (hipMemcpy(wg.data(), h->d_rclk + 8, 256 * sizeof(uint64_t), hipMemcpyDeviceToHost) == hipSuccess) {
        fprintf(stderr, "[lz_ritz_info] workgroup exit times (us after the first wave in), by workgroup id:");
        for (int i = 0; i < 256; ++i) {
          if (i % 16 == 0) fprintf(stderr, "\n   ");
          fprintf(stderr, " %6.1f", wg[i] > c[4] ? (double)(wg[i] - c[4]) * 0.01 : -1.0);
        }
        fprintf(stderr, "\n");
      }
    }
  }
  return LZ_OK;
}

int lz_ritz_gram(lz_handle h, double* gram_out) {
  if (!h || !gram_out) return LZ_ERR_ARG;
  if (!h->d_Y || h->y_n < 1) return fail(h, LZ_ERR_STATE, "lz_ritz_gram: call lz_ritz_vectors first");
  LZ_HIP(h, hipSetDevice(h->dev));
  const int n = h->y_n;
  const int nz_max = 512;
  const int64_t nchunks = h->y_chunked ? (h->y_rows + h->y_chunk - 1) / h->y_chunk : 1;
  if (h->y_chunked && (!h->d_V || h->n != n || h->rows != h->y_rows)) return fail(h, LZ_ERR_STATE, "lz_ritz_gram: the basis of the run is gone");
  // scratch: the K-slice partials of one chunk (the symmetric kernel's or the split-K TN GEMM's), one n x n slice per chunk
  // (added in chunk order at the end), G.  Kept in the handle: a 160-250 MB hipMalloc + hipFree per call cost milliseconds.
  // (rounded UP to whole n x n slices: the grouped form's unit table rides behind the symmetric kernel's partial slices)
  const size_t slices = std::max<size_t>((gram_scratch_doubles(n) + (size_t)n * n - 1) / ((size_t)n * n), (size_t)nz_max);
  const size_t need = (slices + (size_t)nchunks + 1) * (size_t)n * n;
  if (h->gram_cap < need) {
    LZ_TRY(dev_alloc(h, h->d_gram, need));
    h->gram_cap = need;
  }
  if (!h->d_gclk) {
    LZ_TRY(dev_alloc(h, h->d_gclk, 4));
  }
  LZ_HIP(h, hipMemsetAsync(h->d_gclk, 0, 4 * sizeof(uint64_t), h->stream));
  double* part = h->d_gram;
  double* cpart = part + slices * n * n;
  double* dG = cpart + (size_t)nchunks * n * n;
  int rc = LZ_OK;
  h->gram_sym_last = false;
  for (int64_t q = 0; q < nchunks && rc == LZ_OK; ++q) {
    const int64_t r = q * h->y_chunk, nr = std::min<int64_t>(h->y_chunk, h->y_rows - r);
    if (h->y_chunked) rc = ritz_rows_into(h, r, nr, h->d_Y);
    if (rc != LZ_OK) break;
    // flops on the books: the symmetric half, n (n + 1) per row (the full product is 2 n^2; the kernel computes the upper tiles)
    Scope sc(h, LZ_K_RITZ, 8.0 * n * (double)nr, (double)nr * n * (n + 1.0));
    if (h->tune[19] != 1 && launch_gram_sym(h->d_Y, n, nr, n, part, cpart + (size_t)q * n * n, h->stream, reinterpret_cast<unsigned long long*>(h->d_gclk), h->tune[21], h->tune[19])) {
      h->gram_sym_last = true;
    } else {
      const int nz = launch_gram(h->d_Y, n, nr, n, part, nz_max, h->stream);
      launch_sum_slices(part, nz, (int64_t)n * n, cpart + (size_t)q * n * n, h->stream);
    }
    rc = check_launch(h, "gram");
  }
  if (rc == LZ_OK) {
    launch_sum_slices(cpart, (int)nchunks, (int64_t)n * n, dG, h->stream);
    rc = check_launch(h, "gram(sum)");
  }
  if (rc == LZ_OK) rc = comm_allreduce(h, dG, (int64_t)n * n);
  hipError_t e = hipSuccess;
  if (rc == LZ_OK) e = hipMemcpyAsync(gram_out, dG, (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (rc != LZ_OK) return rc;
  if (e != hipSuccess) return fail(h, LZ_ERR_HIP, std::string("lz_ritz_gram: ") + hipGetErrorString(e));
  return LZ_OK;
}

int lz_gram_info(lz_handle h, double* info4) {
  if (!h || !info4) return LZ_ERR_ARG;
  info4[0] = info4[1] = info4[2] = info4[3] = 0.0;
  if (!h->d_gclk || !h->gram_sym_last) return LZ_OK;
  LZ_HIP(h, hipSetDevice(h->dev));
  uint64_t c[4] = {0};
  LZ_HIP(h, hipMemcpyAsync(c, h->d_gclk, sizeof c, hipMemcpyDeviceToHost, h->stream));
  LZ_HIP(h, hipStreamSynchronize(h->stream));
  info4[0] = c[1] ? 100.0 * (double)c[0] / (double)c[1] : 0.0;  // shader clock in MHz while workgroup 0 ran
  info4[1] = c[2] ? (double)c[0] / (double)c[2] : 0.0;          // shader cycles per k-step (4 rows of Y) of its wave 0
  info4[2] = 64.0 * (double)c[3];                               // MFMA issue floor of that: MFMAs per k-step and SIMD x 64 cycles
  info4[3] = (double)c[2];
  return LZ_OK;
}

int lz_ritz_quality(lz_handle h, double* out) {
  if (!h || !out) return LZ_ERR_ARG;
  if (!h->d_Y || h->y_n < 1) return fail(h, LZ_ERR_STATE, "lz_ritz_quality: call lz_ritz_vectors first");
  if (h->kind == 0) return fail(h, LZ_ERR_STATE, "lz_ritz_quality: no matrix set");
  LZ_HIP(h, hipSetDevice(h->dev));
  const int n = h->y_n;
  // Chunked mode: whole Ritz vectors are formed a batch of columns at a time (Yb = V^T-layout x S[:, c0:c0+nb], all rows)
  // and handed to the same kernels with ldy = nb.
  double* Yb = nullptr;
  int nb = n;
  int64_t ldy = n;
  if (h->y_chunked) {
    if (!h->d_V || h->n != n || h->rows != h->y_rows) return fail(h, LZ_ERR_STATE, "lz_ritz_quality: the basis of the run is gone");
    size_t free_b = 0, total_b = 0;
    LZ_HIP(h, hipMemGetInfo(&free_b, &total_b));
    const size_t per_col = (size_t)(round_up(h->y_rows, 16) + 16) * sizeof(double);
    int64_t fit = (int64_t)((free_b > ((size_t)1 << 30) ? free_b - ((size_t)1 << 30) : 0) / 2 / per_col);
    if (h->tune[16] > 0) fit = 16;  // test knob: the smallest batch
    nb = (int)std::min<int64_t>(round_up(n, 16), fit / 16 * 16);
    if (nb < 16) return fail(h, LZ_ERR_NOMEM, "lz_ritz_quality: no device memory left for a 16-column batch of Ritz vectors");
    ldy = nb;
    LZ_TRY(dev_alloc(h, Yb, (size_t)(round_up(h->y_rows, 16) + 16) * nb + 64));
  }
  std::vector<double> sums(2 * (size_t)n);
  int rc = LZ_OK;
  hipError_t e = hipSuccess;
  if (h->world > 1 || h->tune[6] || h->kind == 2) {
    // Row-block partition (and dense matrices on any number of ranks: the fused kernel below walks CSR rows): z = A y_i needs
    // the neighbours' entries of y_i, so every Ritz vector takes the path a Lanczos
    // vector takes - copied into basis row 0 (saved and restored), exchanged (halo or all-gather), multiplied by the
    // SpMV kernel, whose epilogue already delivers y_i . z; ||z||^2 from the three-term kernel with zero coefficients.
    // One all-reduce of the 2 n sums at the end.
    if (!h->d_V || h->n < 1 || h->y_rows != h->rows) rc = fail(h, LZ_ERR_STATE, "lz_ritz_quality: the basis of the run is gone");
    if (rc == LZ_OK && (size_t)2 * n > (size_t)2 * qtw_ldp(h->n + 2) + 8) rc = fail(h, LZ_ERR_STATE, "lz_ritz_quality: coefficient buffer too small");
    if (rc != LZ_OK) {
      big_free(Yb);
      return rc;
    }
    double* v0 = h->d_V;
    double* save = nullptr;  // basis row 0 is borrowed; in chunked mode the batches are formed from the INTACT basis first
    e = hipMemcpyAsync(h->d_r2, v0, (size_t)h->ldv * sizeof(double), hipMemcpyDeviceToDevice, h->stream);
    (void)save;
    if (e == hipSuccess) e = hipMemsetAsync(h->d_nrm2, 0, 2 * sizeof(double), h->stream);
    h->halo_inflight_j = -1;
    for (int c0 = 0; c0 < n && rc == LZ_OK && e == hipSuccess; c0 += nb) {
      const int nc = std::min(nb, n - c0);
      const double* Ysrc = h->d_Y;
      if (h->y_chunked) {
        e = hipMemcpyAsync(v0, h->d_r2, (size_t)h->ldv * sizeof(double), hipMemcpyDeviceToDevice, h->stream);  // the batch GEMM reads basis row 0
        if (e != hipSuccess) break;
        rc = ritz_cols_into(h, c0, nc, Yb, ldy);
        Ysrc = Yb;
      }
      for (int i = 0; i < nc && rc == LZ_OK; ++i) {
        launch_extract_column(Ysrc, ldy, h->y_chunked ? i : c0 + i, h->rows, h->rows_pad, v0, h->stream);
        rc = step_spmv(h, 0, h->d_c + c0 + i, false);
        if (rc != LZ_OK) break;
        const int np = launch_three_term(h->d_r, v0, nullptr, h->d_nrm2, h->d_nrm2, h->rows_pad, h->d_part, h->stream);
        launch_final_sum(h->d_part, np, h->d_c + n + c0 + i, h->stream);
        rc = check_launch(h, "ritz_quality(row-block)");
      }
    }
    hipError_t e2 = hipMemcpyAsync(v0, h->d_r2, (size_t)h->ldv * sizeof(double), hipMemcpyDeviceToDevice, h->stream);  // basis row 0 back
    if (e == hipSuccess) e = e2;
    if (rc == LZ_OK) rc = comm_allreduce(h, h->d_c, 2 * n);
    if (rc == LZ_OK && e == hipSuccess) e = hipMemcpyAsync(sums.data(), h->d_c, sums.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    big_free(Yb);
    if (rc != LZ_OK) return rc;
    if (e != hipSuccess) return fail(h, LZ_ERR_HIP, std::string("lz_ritz_quality: ") + hipGetErrorString(e));
    for (int i = 0; i < n; ++i) out[i] = sums[i] * sums[i] / sums[n + i];
    return LZ_OK;
  }
  const size_t nblk = (size_t)((h->rows + 2047) / 2048);
  double* part = nullptr;
  rc = dev_alloc(h, part, (nblk + 1) * 2 * (size_t)nb);
  if (rc != LZ_OK) {
    big_free(Yb);
    return rc;
  }
  double* dSums = part + nblk * 2 * (size_t)nb;
  for (int c0 = 0; c0 < n && rc == LZ_OK && e == hipSuccess; c0 += nb) {
    const int nc = std::min(nb, n - c0);
    const double* Ysrc = h->d_Y;
    if (h->y_chunked) {
      rc = ritz_cols_into(h, c0, nc, Yb, ldy);
      Ysrc = Yb;
      if (rc != LZ_OK) break;
    }
    {
      Scope sc(h, LZ_K_RITZ, 12.0 * h->csr.nnz + 8.0 * nc * (double)h->rows, 2.0 * (double)h->csr.nnz * nc);
      const int nblocks = launch_ritz_quality(h->csr, Ysrc, ldy, nc, part, h->stream);
      launch_sum_slices(part, nblocks, 2 * (int64_t)nc, dSums, h->stream);
      rc = check_launch(h, "ritz_quality");
    }
    std::vector<double> two(2 * (size_t)nc);
    if (rc == LZ_OK) e = hipMemcpyAsync(two.data(), dSums, two.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    for (int i = 0; i < nc; ++i) {
      sums[(size_t)c0 + i] = two[(size_t)i];
      sums[(size_t)n + c0 + i] = two[(size_t)nc + i];
    }
  }
  big_free(part);
  big_free(Yb);
  if (rc != LZ_OK) return rc;
  if (e != hipSuccess) return fail(h, LZ_ERR_HIP, std::string("lz_ritz_quality: ") + hipGetErrorString(e));
  for (int i = 0; i < n; ++i) out[i] = sums[i] * sums[i] / sums[n + i];
  return LZ_OK;
}

}  // extern "C"
