// Internal declarations shared by the kernel translation units and the C ABI.
// gfx950 (MI355X / CDNA4) only: wave = 64 lanes, 256 CUs in 8 XCDs.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "lanczos_hip.h"

namespace lz {

constexpr int kWave = 64;
constexpr int kTPB = 256;          // threads per block for all streaming kernels
constexpr int kPadDoubles = 32;    // vectors are padded to 256-byte multiples
constexpr int kNumCU = 256;
constexpr int kNumXCD = 8;

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

int xfer_threads();
// run fn(t, lo, hi) over [0, count) on a few host threads (the validation sweeps over all nnz of lz_set_csr)
template <class F>
inline void parallel_ranges(int64_t count, int64_t min_per_thread, F fn) {
  const unsigned hc = std::thread::hardware_concurrency();
  int T = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(16, hc ? hc / 2 : 1), count / std::max<int64_t>(min_per_thread, 1)));
  if (xfer_threads() == 0) T = 1;  // LZ_XFER_THREADS=0: no helper threads anywhere
  std::vector<std::thread> pool;
  const int64_t per = (count + T - 1) / T;
  int started = 0;
  try {
    for (int t = 1; t < T; ++t) {
      pool.emplace_back(fn, t, std::min(count, t * per), std::min(count, (t + 1) * per));
      ++started;
    }
  } catch (const std::system_error&) {
  }
  fn(0, (int64_t)0, std::min(count, per));
  for (int t = started + 1; t < T; ++t) fn(t, std::min(count, t * per), std::min(count, (t + 1) * per));  // threads that could not be had
  for (auto& th : pool) th.join();
}
constexpr int kMaxHostThreads = 16;

// ---- CSR SpMV ----------------------------------------------------------
struct CsrDev {
  int64_t rows = 0, ncols = 0, nnz = 0;
  int32_t* rowptr = nullptr;
  int32_t* colidx = nullptr;
  double* vals = nullptr;
  int fixed_k = 0;            // > 0: every row has exactly fixed_k entries
  int32_t* rowblk = nullptr;  // CSR-stream row blocks: rows [rowblk[b], rowblk[b+1])
  int n_rowblk = 0;
  int fixed_rb = 512;         // rows per block of the fixed-K kernel
  int blk_nnz_cap = 4096;     // products per row block (LDS tile of the CSR-stream kernel)
  int max_row_nnz = 0;
  double avg_row_nnz = 0;
  double far_frac = 0;        // share of entries whose column is > 2^18 away from their row: no L2 reuse of x to speak of
  struct PbDev* pb = nullptr; // column-blocked two-phase layout (lz_spmv_pb.hip), built when the matrix has no column locality
  // ELL-ordered copy of a fixed-K matrix (lz_spmv.hip, k_spmv_ell): blocks of ell_rb rows, entry k of every row of a block
  // contiguous - a lane owns whole rows, its loads and the stencil's gathers are coalesced, no LDS transposition.
  int32_t* ell_c = nullptr;
  double* ell_v = nullptr;
  int ell_rb = 0;             // rows per block of the ELL copy (0: none)
  int ell_variant = 0;        // 0: one row per lane and trip (rows t, t + 256, ...); 1: two adjacent rows per lane (16-byte loads)
  bool ell_default = false;   // true: every SpMV takes the ELL copy (knob 17 >= 2, or a row-class coded one); false: only the partial loop's fused SpMV does
  // row-class coding of the ELL copy (round 5; k_spmv_ell, CODED): one byte per row names its class; a class is the row's K offsets
  // col - row (ell_coded >= 1: ell_c is not kept) and, for constant coefficients, its K values (ell_coded == 2: ell_v is not kept either)
  uint8_t* ell_cls = nullptr;
  int32_t* cls_off = nullptr; // [256][K]
  double* cls_val = nullptr;  // [256][K]
  double* cls_diag = nullptr; // ell_coded == 3: the diagonal's values, one per row (the class holds the offsets and the OTHER values)
  int ell_coded = 0;
  int ell_ncls = 0;
  int cls_group = 0;          // knob 23: 0 k_spmv_cls2 (two adjacent rows per lane); 1 / 3: k_spmv_cls with one / two units per workgroup (A/B)
  const int32_t* host_colidx = nullptr;  // the caller's arrays, valid ONLY inside lz_set_csr / lz_set_csr_transpose (pb_build reads them)
  const double* host_vals = nullptr;
};

// ---- column-blocked two-phase SpMV (lz_spmv_pb.hip): gathers out of LDS only; y bit-identical to the CSR-stream kernel
hipError_t pb_build(const CsrDev& A, const int32_t* rowptr_host, PbDev** out, hipStream_t s, int cap_knob = 0, int groups = 0);  // *out == nullptr: not applicable; groups > 1: the A/B arm of knob 22
void pb_free(PbDev*& pb);
int pb_num_partials(const PbDev* pb);
void pb_layout_info(const PbDev* pb, int64_t* np, int* in_bytes_x2, int* diag);
int launch_spmv_pb(const CsrDev& A, const PbDev* pb, const double* x, double* y, const double* x_own, double* part, hipStream_t s);

// launch wrappers (all asynchronous on `s`)
// y = A x (rows), part[b] = sum_{rows of block b} x_own[i] * y[i]; returns number of partials written
int launch_spmv_csr(const CsrDev& A, const double* x, double* y, const double* x_own, double* part, int flags,
                    hipStream_t s);
// ELL copy of a fixed-K matrix (K in {5, 7, 27}); variant as CsrDev::ell_variant.  ell_free releases it.
hipError_t ell_build(CsrDev& A, int variant, hipStream_t s, int coding = 0, bool plain_fallback = true);
void ell_free(CsrDev& A);
bool ell_usable(const CsrDev& A, int flags);
// The device-resident partial re-orthogonalisation loop's SpMV: when gate[0] == 0 (no sweep ran on this vector) the kernel
// forms v_j = r / sqrt(nrm2[0]) ITSELF wherever it reads an entry of x (IEEE division: the same bits wherever it is formed),
// stores the rows it owns to vj and beta to beta_slot; when gate[0] != 0 the sweep kernels have written vj and it is a plain
// y = A vj.  y must not alias r.  Returns the number of alpha partials.
struct EllCode {
  const double* diag = nullptr;  // coded 3: the value of the entry at offset 0 comes from here
  const uint8_t* cls = nullptr;
  const int32_t* off = nullptr;
  const double* val = nullptr;
  int ncls = 0;
};
struct SpmvScale {
  const double* r = nullptr;
  const double* nrm2 = nullptr;
  double* vj = nullptr;
  double* beta_slot = nullptr;
  const int* gate = nullptr;
};
int launch_spmv_ell(const CsrDev& A, const double* x, double* y, const double* x_own, double* part, hipStream_t s,
                    const SpmvScale* sc = nullptr);
struct StencilArgs {  // by-value kernel argument of the stencil assembly
  int Nx, Ny, Nz, negate, pot_kind, renumber, nranges;
  int64_t row0, rows_local;
  double tf, w[4], par[8];
  int64_t gstart[16], glen[16], gext[16];  // ghost ranges: global start, length, first extended-local index
};
void launch_build_stencil3d(const StencilArgs& a, int points, const double* pot, int32_t* rowptr, int32_t* colidx, double* vals,
                            hipStream_t s);
int launch_gemv_dense(const double* A, int64_t M, int64_t cols, int64_t lda, const double* x, const double* x_own, double* y,
                      double* part, hipStream_t s);

// out[0] = sum(part[0..n)) (fixed order)
void launch_final_sum(const double* part, int n, double* out, hipStream_t s);
// c[i] = sum_b part[i*G + b]; transposed (4x4x4 MFMA kernel): c[i] = sum_b part[b*qtw_ldp(nrows) + i]
inline int qtw_ldp(int nrows) { return (nrows + 15) & ~15; }
void launch_final_rows(const double* part, int nrows, int G, double* c, hipStream_t s, bool transposed = false, const int* gate = nullptr);
void launch_final_rows_t(const double* part, int G, int ldp, int nout, double* c, hipStream_t s, const int* gate = nullptr);
// up to four independent k_final_sum's in one launch: out[q][0] = sum(part[q][0 .. n[q])) in k_final_sum's grouping
struct FinalMulti {
  const double* part[4];
  int n[4];
  double* out[4];
};
void launch_final_sum_multi(const FinalMulti& fm, int count, hipStream_t s);

struct QtwPlan {
  int64_t L = 0;    // elements of w owned by one block (multiple of 512)
  int G = 0;        // number of blocks
  int P = 0;        // partials per basis row (G for the VALU kernel, 4G for the MFMA kernel)
  bool mfma = false;
  int family = 2;   // 2 = 4x4x4 MFMA (default), 1 = 16x16x4 MFMA, 0 = VALU
  int variant = 0;  // A/B knob (unroll / rows per tile)
};
QtwPlan plan_qtw(int64_t len, int flags, const int* tune, int nrows_max);
// pass 1 of the re-orthogonalisation (+ optional v_j = r / sqrt(nrm2)):
//   part[i*G + b] = sum_{m in block b} V[i][m] * V[j][m],  i in [0, nrows)
// Fused small-problem mode of pass 1 (mode 4): the kernel's prologue finishes the previous step itself - alpha from the
// SpMV's block partials (k_final_sum's grouping), r = (y - alpha v_prev) - beta v_prev2 on its own slice - before it
// stages and dots r.  Three launches per Lanczos step instead of six, same bits.
struct QtwFuse {
  const double* apart = nullptr;  // alpha partials of the SpMV that produced y
  int np = 0;
  int jprev = 0, jprev2 = -1;     // basis rows of v_prev (the vector just multiplied) and v_prev2 (-1: none)
  const double* beta_prev = nullptr;
  double* alpha_out = nullptr;
  double* r_out = nullptr;        // where r goes (NOT y itself: with the row split several blocks read the same slice of y)
  const int* gate = nullptr;      // any mode: the kernel returns at once when gate[0] == 0 (device-resident partial re-orthogonalisation)
  unsigned* ticket = nullptr;     // mode 1: the last block to finish also adds up all blocks' runs into c_out (k_final_rows_t's order)
  double* c_out = nullptr;
};
// returns the error of the per-kernel LDS-limit raise (hipFuncSetAttribute), if that was needed and failed
hipError_t launch_qtw(double* V, int64_t ldv, int64_t len, int nrows, int j, const double* r, const double* nrm2,
                      double* beta_slot, const QtwPlan& plan, double* part, int mode, hipStream_t s, const QtwFuse* fuse = nullptr);
// pass 2: V[j] = 2 V[j] - sum_{i<nrows} c[i] V[i] (sequential, unfused: bitwise NumPy order)
// raw_c (fused mode only): c holds the reduced sums, beta = sqrt(c[j]) is formed and stored by the kernel itself
// raw_c == 2 (fused small-problem mode): c points at pass 1's block partials, cG runs of cldp doubles; the kernel adds
// them in k_final_rows_t's order itself
void launch_update(double* V, int64_t ldv, int64_t len, int nrows, int j, const double* c, const double* r_fused,
                   double* beta, int variant, hipStream_t s, int64_t pos_lo = 0, int64_t pos_hi = -1, int raw_c = 0,
                   int64_t pos_lo_b = 0, int64_t pos_hi_b = 0,  // second range: only with the small-range (face) kernel
                   int cG = 0, int cldp = 0, const int* gate = nullptr,
                   // one-reduce partial loop (r_fused, raw_c == 1, gate): w = (r_fused - asub[0] usub) / beta formed in the kernel; gate[0] == 0:
                   // V[j] = w (no sweep) instead of returning
                   const double* usub = nullptr, const double* asub = nullptr);
// gate != nullptr: runs only when gate[0] == 0 (the step without a sweep)
void launch_scale_store(double* vj, const double* r, const double* nrm2, double* beta_slot, int64_t len, hipStream_t s, const int* gate = nullptr);
// ---- device-resident partial re-orthogonalisation (Simon's omega-recurrence in a one-block kernel) ----
// State: st[0] = ||A|| estimate, st[1] = force_next, st[2 .. 2 + n + 2) = hb (hb[k] = the norm that formed V[k]), then three
// rows of n + 1 doubles (omega_{j,:} lives in row j % 3); ist[0] = the gate of the coming step (1: sweep), ist[1] = number of
// sweeps so far, ist[2 + j] = whether step j swept.
inline size_t omega_state_doubles(int n) { return (size_t)2 + (n + 2) + 3 * (size_t)(n + 1); }
inline size_t omega_state_ints(int n) { return (size_t)2 + n + 1; }
// Prepares the decision for step jn (called after step jn - 1 has left alpha[jn-1] and ||r||^2; jn == 0: after the warm-up).
// part != nullptr: first nrm2[0] = sum(part[0..np)) in k_final_sum's order (single rank: one launch for both).
void launch_omega(const double* part, int np, double* nrm2, const double* alpha, int jn, int n, double* st, int* ist, hipStream_t s,
                  double* c_clear = nullptr);  // c_clear: jn + 1 coefficients zeroed when step jn does not sweep (partitioned runs)
// The post-reduce kernel of the one-reduce partial loop (k_partial_onered_post, lz_reorth.hip): finishes the all-reduced buffer,
// advances the omega-recurrence, takes the look-ahead sweep decision of step j + 1, clears the next step's buffer.  j < 0: initialises
// the gates (ist: omega_onered_ints(n) ints).
inline size_t omega_onered_ints(int n) { return (size_t)4 + n + 2; }
void launch_partial_onered_post(double* buf, double* bufn, int nzero_next, int m, int ldp, double* alpha_slot, double* nrm2, const double* alpha,
                                int j, int n, double* st, int* ist, double kappa, hipStream_t s);
// r = r - beta vm (vm may be nullptr: r unchanged) + block partials of [r.r, u.r, u.u] at part[b], part[G + b], part[2 G + b]; returns G
int launch_three_term_self(double* r, const double* u, const double* vm, const double* beta, int64_t len, double* part, hipStream_t s);
void launch_fused_prepare(double* c, int j, double* beta_slot, hipStream_t s);
void launch_onereduce_prepare(double* buf, int m, int ldp, double* alpha_slot, double* bad, hipStream_t s);
// r = (r - alpha v_j) - beta v_jm1 ; part[b] = partial ||r||^2 ; returns number of partials
int launch_three_term(double* r, const double* vj, const double* vjm1, const double* alpha, const double* beta,
                      int64_t len, double* part, hipStream_t s);
// gather x[idx[k]] -> buf[k]
void launch_gather(const double* x, const int32_t* idx, int64_t n, double* buf, hipStream_t s);

// ---- two-sided Lanczos (lz_twosided.hip) ----
// One link of the bi-orthogonalisation chain for both sides: optional (x, y) = (xs / f[0], ys / f[1] * f[2]) [first],
// optional x -= Sp[0]/Sp[1] * ap, y -= Sp[2]/Sp[3] * bp [pend], store, then dots: 0 -> S = [x.a, a.a, y.b, b.b] (epi 0);
// 1 -> x.y (epi 1: fo = {sqrt|.|, sqrt|.|, sign}); 2 -> [x.x, y.y] (epi 2: fo = {sqrt, sqrt, 1}); 3 -> none.
void launch_bi(int first, int pend, int dots, double* x, double* y, const double* xs, const double* ys, const double* f,
               const double* ap, const double* bp, const double* Sp, const double* a, const double* b, int64_t len, double* part,
               int epi, double* S, double* fo, double* o0, double* o1, unsigned* ticket, hipStream_t s, const double* pend_part = nullptr,
               bool defer_fold = false);
// r -= c0 u, s -= c1 v (sub); dots 0: o0 = (da.r + db.s)/2; 1: o0 = sqrt|r.s|, o1 = r.s/o0, fo = {o0, o1, 1}; 2: o0 = da.r
void launch_bi_two_term(int sub, int dots, double* r, double* sv, const double* u, const double* v, const double* c0, const double* c1,
                        const double* da, const double* db, int64_t len, double* part, double* fo, double* o0, double* o1,
                        unsigned* ticket, hipStream_t s);  // ticket != nullptr: the last block folds the partials (one launch)
// ---- result publication (lz_xfer.hip): large device -> host copies through a ring of pinned staging buffers + host copy threads
struct Xfer;
void xfer_free(Xfer*& x);
hipError_t xfer_d2h(int dev, hipStream_t stream, Xfer*& state, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width,
                    size_t height);  // returns with the destination complete (stream synchronised)
int xfer_threads();

void launch_bi_mem_safe(double* xrow, const double* B, int64_t ldv, int n, int j, int64_t len, double* coef, hipStream_t s);
int bi_partials_needed();  // per partial buffer; the handle keeps two (deferred folds ping-pong between them)

// Y(rows x n, row-major, ldy) = sum_k V[k][m] * S[k][i]   (FP64 MFMA)
// variant 0: the S-stationary kernel (S in registers, V tiles through LDS) for 49 <= n <= 200 when there are enough row
// tiles for a persistent grid, else one workgroup per 128 rows, one wave per SIMD owning 32 rows x all columns; 1: the
// latter always.  (The retired arms - persistent waves, S staged through LDS - live in the kernel-bench build.)
// clk (optional, 8 words): in-kernel clock record of the S-stationary kernel, see lz_ritz_info.
// Returns the error of the per-kernel LDS-limit raise (hipFuncSetAttribute) if that was needed and failed.
hipError_t launch_ritz_gemm(const double* V, int64_t ldv, int64_t rows, int n, const double* Spad, int npad, double* Y,
                            int64_t ldy, hipStream_t s, int variant = 0, unsigned long long* clk = nullptr);
void launch_ritz_gemm_cols(const double* V, int64_t ldv, int64_t rows, int kcount, const double* B, int ldb, int ncols, double* Y,
                           int64_t ldy, hipStream_t s);
// G = Y^T Y as K-chunk partials (n x n each); returns the number of chunks (<= nz_max)
int launch_gram(const double* Y, int64_t ldy, int64_t rows, int n, double* part, int nz_max, hipStream_t s);
void launch_sum_slices(const double* part, int nz, int64_t count, double* out, hipStream_t s);
// accumulator-stationary symmetric Gram kernel (n >= 33, >= 4096 rows; more than 208 columns: in groups of 11 column tiles):
// out = Y^T Y complete; false: not covered
constexpr int kGramMaxSlices = 768;
size_t gram_scratch_doubles(int n);
bool launch_gram_sym(const double* Y, int64_t ldy, int64_t rows, int n, double* part, double* out, hipStream_t s,
                     unsigned long long* clk = nullptr, int nz_override = 0,  // nz_override > 0: that many K slices (A/B knob 21)
                     int variant = 0);  // 0: operands staged through LDS (even n), 2: the register-ring kernels (knob 19 = 2)
// per-block [s1 (n), s2 (n)] partials of the Ritz-vector quality sums; returns the number of blocks
int launch_ritz_quality(const CsrDev& A, const double* Y, int64_t ldy, int n, double* part, hipStream_t s);
// x[r] = Y[r * ldy + col] for r < rows, 0 for rows <= r < rows_pad
void launch_extract_column(const double* Y, int64_t ldy, int col, int64_t rows, int64_t rows_pad, double* x, hipStream_t s);

// ---- small-problem engine (lz_small.hip): the whole run as one cooperative kernel
struct SmallArgs {
  int kind;  // 1 CSR, 2 dense
  const int32_t* rowptr;
  const int32_t* colidx;
  const double* vals;
  const int32_t* rowblk;  // CSR-stream row blocks (nparts + 1); nullptr: fixed blocks of 512 rows (fixed-K kernel)
  int nparts;             // alpha partials of the multi-kernel path: row blocks (CSR) or ceil(rows / 4) (dense)
  const double* dense;
  int64_t lda;
  int rows, rows_pad, n;
  int64_t ldv;
  double* V;
  const double* x0;  // scratch copy of the start vector (rows_pad doubles)
  double* y;     // SpMV output (the handle's r vector)
  double* drow;  // rows: V[j]_i * (A V[j])_i
  double* pc;    // n + 1: raw re-orthogonalisation sums [V_i . r (i < j), r . r]
  double* alpha;
  double* beta;
  unsigned* bar;  // 4 zeroed words: [0] device-scope barrier, [1] XCD-local barrier, [2] status (0 ok, 2 placement, 3 barrier timeout)
  unsigned* xcc;  // one word per participating block: the XCD it found itself on (+ 1)
  unsigned long long zero;  // 0 (a run-time value: the addend of the L2-atomic "loads", see ld_sh)
};

constexpr int kSmallMaxPadRows = 1280;
int small_grid(int rows_pad);
hipError_t launch_small_run(const SmallArgs& a, int nb, bool local, hipStream_t s);
hipError_t launch_small_step(const SmallArgs& a, int mode, int j, int nb, hipStream_t s);

}  // namespace lz
