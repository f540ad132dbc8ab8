/*
 * lanczos_hip.h - C ABI of liblanczos_hip.so (MI355X / gfx950 Lanczos hot path).
 *
 * The reference (jgslunde/Lanczos) has no FFI layer: its only interface is the
 * Python class surface of Python/Regular/Lanczos.py:11-337 and
 * Python/Irregular/IrrLanczos.py:12-554, whose "GPU backend" is a run of CuPy
 * library calls.  Each entry point below therefore cites the CuPy/NumPy call
 * sites of the reference that it replaces (paths relative to /root/reference).
 * The Python host mirror of the class surface lives in lanczos_amd/ and reaches
 * this library through ctypes only (no torch types cross this boundary).
 *
 * Conventions
 *   - every function returns an int status: 0 = LZ_OK, negative = error;
 *     lz_last_error(h) returns a human-readable message for the last failure.
 *   - the caller owns all host buffers (pointer + explicit sizes); the library
 *     owns all device memory behind the opaque handle.
 *   - fp64 values, int32 CSR indices, row-major everywhere.
 *   - one handle = one GPU = one rank.  A handle is not thread-safe.
 *   - distributed runs: the basis, r and the matrix are row-block partitioned;
 *     "local" sizes refer to this rank's rows.
 */
#ifndef LANCZOS_HIP_H
#define LANCZOS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lz_context* lz_handle;

enum lz_status {
  LZ_OK = 0,
  LZ_WARN_BREAKDOWN = 1, /* lz_run finished, but a residual norm beta fell to <= 64 eps * max(|alpha|, |beta|) or a
                            coefficient is not finite: the Krylov space is exhausted.  The reference divides blindly
                            (Lanczos.py:113) and carries on with rounding noise / inf / NaN; the coefficients are
                            delivered exactly as computed, this status is the only difference. */
  LZ_ERR_ARG = -1,       /* bad argument / shape */
  LZ_ERR_HIP = -2,       /* HIP runtime error */
  LZ_ERR_COMM = -3,      /* RCCL / host-collective error */
  LZ_ERR_STATE = -4,     /* call order violated (e.g. run before set_csr) */
  LZ_ERR_NOMEM = -5,     /* device allocation failed */
  LZ_ERR_NODEVICE = -6   /* no usable GPU */
};

/* flags for lz_run / lz_set_options */
enum lz_flags {
  LZ_FLAG_NONE = 0,
  LZ_FLAG_PROFILE = 1,        /* bracket every hot kernel with hipEvents (lz_get_timings) */
  LZ_FLAG_QTW_MFMA = 2,       /* kernel-bench build only (retired A/B arm, 20 % slower): the v_mfma_f64_16x16x4_f64 Q^T w
                                 kernel; the product library refuses it (default: the 4x4x4 MFMA kernel)              */
  LZ_FLAG_QTW_VALU = 4,       /* the VALU + wave-shuffle Q^T w kernel (also the automatic fallback beyond ~5000 basis rows) */
  LZ_FLAG_SPMV_SCALAR = 8,    /* force the plain one-thread-per-row CSR kernel            */
  LZ_FLAG_FUSED_NORM = 16,    /* ||r||^2 rides with the Q^T r sums: pass 1 dots the raw residual, the update kernel forms
                                 beta, c_i = (V_i.r)/beta and V[j] = r/beta (one pass over V[j] and, at N > 1, one all-reduce
                                 less).  The Python layers set it by default; 0 = scale first, then dot (reference order) */
  LZ_FLAG_SPMV_STREAM = 32,   /* force the generic CSR-stream kernel (no fixed-K fast path) */
  LZ_FLAG_OVERLAP_HALO = 128, /* multi-rank, contiguous (stencil) halos: update the faces of V[j] first, exchange them on a
                                 second stream while the interior is updated; the SpMV waits on an event (opt-in)    */
  LZ_FLAG_ONE_REDUCE = 256,   /* opt-in, multi-rank runs: ONE all-reduce per iteration.  Pass 1 dots the basis against two
                                 columns at once on the matrix cores - r'' = A v_j - beta v_{j-1} and v_j - so that alpha_j, the
                                 coefficients V_i.(r'' - alpha_j v_j) = V_i.r'' - alpha_j V_i.v_j and |r|^2 = r''.r'' - 2 alpha_j
                                 v_j.r'' + alpha_j^2 v_j.v_j all come out of one reduced buffer (implies LZ_FLAG_FUSED_NORM).
                                 Same mathematics, different rounding (the three-term update subtracts beta v_{j-1} first, |r|^2
                                 is formed from three sums whose relative error is ~eps (alpha^2 + beta^2) / beta^2): not
                                 bit-identical to the default, and within 1e-10 only while |alpha| is not >> beta.  A guard
                                 watches exactly that: when |r|^2 falls below 1e-4 of r''.r'' (or is not positive) at any
                                 step, lz_run REPEATS the solve on the default loop and lz_last_engine reports 5 - the
                                 delivered coefficients then are the default loop's, bit for bit.
                                 With LZ_FLAG_REORTH_PARTIAL (round 5, lz_last_engine 8): the device-decided partial loop with
                                 ONE all-reduce + one exchange per iteration, sweep or no sweep (instead of three + one): the
                                 one reduced buffer carries alpha's partial, the three self terms of |r|^2 and - in a step
                                 that sweeps - the basis dots; the sweep decision is a one-step look-ahead of Simon's
                                 recurrence taken from reduced sums only, so every rank agrees (lz_last_sweep_misses counts
                                 the vectors the look-ahead should have swept and did not; they are swept one step late).
                                 Same guard; a repeat runs the three-collective partial loop (engine 5).              */
  LZ_FLAG_REORTH_PARTIAL = 64 /* opt-in: partial re-orthogonalisation (Simon 1984).  The reference sweeps the whole basis
                                 every step; with this flag the sweep (same kernels, same arithmetic) runs only when the
                                 omega-recurrence estimate of the loss of orthogonality exceeds sqrt(eps), on that and the
                                 next step.  The recurrence and the decision live on the device (no host synchronisation
                                 inside lz_run: lz_last_host_syncs).  The basis stays semi-orthogonal (<= sqrt(eps)), which keeps T - and so the
                                 Ritz values - within O(eps ||A||) of the full-sweep run; V itself differs at that level. */
};

/* kernel classes reported by lz_get_timings */
enum lz_kernel_class {
  LZ_K_SPMV = 0,     /* r = A v_j with fused alpha partials             */
  LZ_K_QTW = 1,      /* v_j = r/beta; c = Q^T v_j (reorth pass 1)        */
  LZ_K_UPDATE = 2,   /* v_j = 2 v_j - Q c        (reorth pass 2)        */
  LZ_K_THREE = 3,    /* r = r - alpha v_j - beta v_{j-1}; ||r||^2        */
  LZ_K_FINAL = 4,    /* tiny second-stage reductions                    */
  LZ_K_COMM = 5,     /* all-reduce / halo exchange / all-gather         */
  LZ_K_RITZ = 6,     /* Y = V^T-layout GEMM (FP64 MFMA)                 */
  LZ_K_COUNT = 7
};

typedef struct lz_timings {
  double ms[LZ_K_COUNT];             /* summed device time of the TIMED launches per class (hipEvent)   */
  double timed_bytes[LZ_K_COUNT];    /* ALGORITHMIC bytes of the timed launches    */
  int64_t timed_launches[LZ_K_COUNT];/* launches bracketed by events (all of them, or 1 iteration in     */
                                     /* `stride` when lz_set_tuning(h, 7, stride) samples the profiling) */
  double bytes[LZ_K_COUNT];          /* algorithmic bytes of ALL launches per class                      */
  double flops[LZ_K_COUNT];          /* algorithmic flops of all launches per class                      */
  int64_t launches[LZ_K_COUNT];      /* number of launches per class                                     */
  double total_ms;                   /* device time of the whole lz_run(s) (events)                      */
} lz_timings;

/* ---- library / device ------------------------------------------------- */
int lz_version(void);
int lz_device_count(int* count);
/* Replaces: `import cupy as np` backend switch, Lanczos.py:85-88. */
int lz_create(lz_handle* out, int device_id);
int lz_destroy(lz_handle h);
const char* lz_last_error(lz_handle h); /* h may be NULL: last error of lz_create */
int lz_set_options(lz_handle h, int flags);
/* Tuning knobs; they take effect at the next lz_set_csr / lz_basis_alloc / lz_run / lz_ritz_vectors, and results never depend
 * on them beyond summation order.
 *    0  Q^T w slice length per block            1  Q^T w kernel variant (unroll / rows per tile)
 *    2 / 4  CSR-stream rows / entries per block  5  fixed-K rows per block
 *    6  issue the collectives even when world == 1 (tests of the RCCL calls with a 1-rank communicator)
 *    7  profile only every value-th iteration of lz_run
 *    8  update-kernel variant (1 cached loads, 2 one position per lane, 3/4/5 slice owner with 8/4/1 positions per lane)
 *    9  Ritz back-transform kernel: 0 auto (33..128 columns: S resident in LDS, 32-row tiles; 129..200: S-stationary, S in
 *       registers; else one workgroup per 128 rows), 1 the latter always
 *   10  irregular SpMV: entries per row block   12  1 = no row-stride skew
 *   13  1 = NaN-poison a fresh basis allocation before the required parts are cleared (test knob)
 *   14  irregular SpMV plan (0 auto: the column-blocked two-phase kernels for matrices without column locality, 1 never, 2 always)
 *   15  loop structure (0 auto: three launches per step for small problems, five up to 4e6 rows per rank, else six; 1 six always)
 *   16  rows per chunk of the CHUNKED Ritz mode (0 auto: chunked only when Y does not fit beside the basis; > 0 forces it: tests)
 *   11  two-sided Gram-Schmidt links (0 / 1: streaming kernel + fold kernel per link)
 *   17  fixed-K (stencil) SpMV layout: 0 auto (CSR-order kernel with products staged through LDS; the ELL-ordered second copy -
 *       a lane owns whole rows - is built and used only by the partial re-orthogonalisation loop's fused SpMV; 27 entries per
 *       row, which have no CSR-order fixed-K kernel: ELL for every SpMV, 3 % faster than CSR-stream), 1 never ELL,
 *       2 ELL for every SpMV, one row per lane and trip, 3 ELL, two adjacent rows per lane (16-byte loads).
 *       Round 5: auto first looks for ROW CLASSES (lz_spmv_coding) - a coded ELL copy is every SpMV's default; 2 / 3 keep the uncoded
 *       copy (A/B), 4 codes the offsets only even where the values repeat as well (A/B)
 *   19  Gram matrix of the Ritz vectors: 0 auto (accumulator-stationary symmetric kernel, operands staged through LDS once per workgroup),
 *       1 the split-K TN GEMM always, 2 the symmetric kernel with per-wave register rings (round 4; also what an odd n runs)
 *   18  partial re-orthogonalisation loop: 0 auto (device-resident decisions, lz_last_engine 7), 1 the host-decided loop
 *       (two scalars read back per step; same bits), 2 device-resident but with the separate scale pass (no fused r / beta),
 *       3 device-resident with pass 1's second-stage sums as a kernel of their own (default: pass 1's last block adds them)
 *   23  fully row-class coded SpMV (lz_spmv_coding == 2): 0 auto (two ADJACENT rows per lane: 16-byte gathers, half the vector-memory
 *       instructions), 1 / 3 one row per lane and trip with one / two 512-row units per workgroup (A/B; same bits in every form - the
 *       alpha partials are per unit and per the one-row-per-lane lane mapping)
 *   22  irregular (two-phase) SpMV: interleave its two phases over this many groups of row blocks (A/B arm of round 5: the product
 *       stream of a group could stay in the Infinity Cache between the phases; measured 8-110 % slower - DESIGN.md section 4; kernel-bench
 *       build only; 0 / 1 = off)
 *   21  Gram matrix: number of K slices (workgroups per unit) of the symmetric kernel; 0 auto (whole residency rounds of 256)
 *   20  one-reduce partial loop (LZ_FLAG_REORTH_PARTIAL | LZ_FLAG_ONE_REDUCE): safety factor kappa of the look-ahead sweep
 *       decision (a sweep is due when kappa * max |predicted omega| > sqrt(eps); 0 = the default, 4)
 * The product library returns LZ_ERR_ARG for everything that lives only in the kernel-bench build (make KBENCH=1 ->
 * liblanczos_kbench.so, loaded by tools/ and by the tests of those arms; sources: lz_small.hip, lz_*_kbench.h): the A/B arms
 * retired in round 3 because they measured slower - the one-kernel and one-launch-per-step engines
 * (15 = 2, 3, 5), the persistent / LDS-staged / 16-row-tile Ritz GEMMs (9 = 2, 3, 4, 6), the ticket / deferred-fold two-sided links
 * (11 = 2, 3), the two-phase SpMV's row-block-group interleaving (22 >= 2), and LZ_FLAG_QTW_MFMA (lz_set_options).  The timing-only ablation arms of rounds 1-4 (kernel variants that computed
 * wrong results on purpose: knob 1 >= 20, knob 3, knob 9 >= 10) were deleted from both builds in round 5. */
int lz_set_tuning(lz_handle h, int index, int value);
/* "hip=<path of the libamdhip64 this library is bound to>;rccl=<path of the librccl it dlopened, or empty>".
 * RCCL is always taken from the directory of that HIP runtime (LZ_RCCL_PATH overrides): see DESIGN.md section 6 and LAB_NOTEBOOK.md section 5. */
int lz_runtime_info(char* buf, size_t buflen);
int lz_device_synchronize(lz_handle h);
/* free / total device memory of the handle's GPU as the runtime reports it (hipMemGetInfo): the figure the resident-versus-chunked
 * decision of lz_ritz_vectors is taken on; tests use it to see that lz_destroy gives everything back */
int lz_device_memory(lz_handle h, int64_t* free_bytes, int64_t* total_bytes);
int lz_device_name(lz_handle h, char* buf, size_t buflen);
/* vectors are padded to 256-byte multiples on the device; in halo mode the ghost
 * entries of the extended local vector start at index lz_padded_rows(rows_local). */
int64_t lz_padded_rows(int64_t rows);

/* ---- distributed setup (optional; default is a single rank) ------------
 * The reference has no communication layer; the
 * partition below is this build's own design: rows are split in contiguous
 * blocks, alpha/beta/c are summed with an all-reduce, and the SpMV input is
 * exchanged either as neighbour halos (send/recv lists) or as an all-gather. */
int lz_comm_load(void);                           /* dlopen RCCL now (optional; lz_comm_* do it lazily) */
int lz_comm_unique_id(void* id, size_t id_bytes); /* >= 128 bytes; rank 0 calls, host broadcasts */
int lz_comm_init_rccl(lz_handle h, int world, int rank, const void* id, size_t id_bytes);
/* host-staged collectives (tests / fallback): the library copies device data to
 * the given host buffer, calls back, and copies the result to the device. */
typedef int (*lz_host_allreduce_fn)(void* user, double* buf, int64_t count);
/* peers/send_counts/recv_counts have npeers entries; sendbuf/recvbuf are the
 * concatenated per-peer segments in peer order. */
typedef int (*lz_host_exchange_fn)(void* user, int npeers, const int32_t* peers, const double* sendbuf,
                                   const int64_t* send_counts, double* recvbuf, const int64_t* recv_counts);
/* allgather: sendbuf (count doubles) -> recvbuf (world*count doubles) */
typedef int (*lz_host_allgather_fn)(void* user, const double* sendbuf, double* recvbuf, int64_t count);
int lz_comm_init_host(lz_handle h, int world, int rank, lz_host_allreduce_fn ar, lz_host_exchange_fn ex,
                      lz_host_allgather_fn ag, void* user);

/* ---- matrix ----------------------------------------------------------- *
 * Replaces: cupyx.scipy.sparse.csr_matrix(H, dtype=float64), Lanczos.py:88
 * (csc_matrix in IrrLanczos.py:205; a symmetric CSC is the same arrays).
 * Single rank: rows_local == M_global, ncols_ext == M_global.
 * Multi rank : this rank's row block; colidx index the EXTENDED local vector
 *   [ owned rows (rows_local) | ghost entries ... ] of length ncols_ext
 *   (halo mode, see lz_set_halo) or the padded global vector (all-gather mode,
 *   see lz_set_allgather). */
int lz_set_csr(lz_handle h, int64_t M_global, int64_t row0, int64_t rows_local, int64_t ncols_ext, int64_t nnz,
               const int32_t* rowptr, const int32_t* colidx, const double* vals);
/* Dense row-major A (M x M), single rank only.  Replaces the dense->CSR
 * conversion the reference's GPU path performs for ndarray input
 * (1Dbox.py:27 -> Lanczos.py:88) with a real dense GEMV. */
int lz_set_dense(lz_handle h, int64_t M, const double* A);
/* Row block [row0, row0 + rows_local) of a dense symmetric A split over ranks (the north star's dense multi-GPU case;
 * nothing to mirror in the single-process reference).  A is rows_local x ncols_ext row-major with the columns laid out
 * like the all-gathered vector: rank q's entries at [q * chunk, q * chunk + rows_q), zero columns in the padding
 * (ncols_ext = world * chunk; follow with lz_set_allgather(h, chunk)).  One rank: rows_local = ncols_ext = M_global. */
int lz_set_dense_block(lz_handle h, int64_t M_global, int64_t row0, int64_t rows_local, int64_t ncols_ext, const double* A);
/* Assemble the reference's regular-grid Hamiltonian directly in device CSR (replaces the
 * Python-loop COO build of Python/Regular/Hamiltonian.py:45-128 plus `H = -T + V; H.sort_indices()` of
 * 3Ddeuteron.py:80-81): periodic N^3 grid, flat index x + y N + z N^2, `points` = 7 or 27, weights4 =
 * {centre, face, edge, corner} as the host computed them (7-point uses the first two).  Entries are
 * T_factor * w (negated if negate_T), plus potential[row] (N^3 host doubles, may be NULL) on the diagonal;
 * columns sorted.  The matrix becomes the handle's operator, exactly as after lz_set_csr. */
int lz_build_stencil3d(lz_handle h, int N, int points, double T_factor, const double* weights4, const double* potential,
                       int negate_T);
/* The same assembly for a Nx x Ny x Nz grid and for ONE RANK'S ROW BLOCK [row0, row0 + rows_local) of it (the 8-GPU form of
 * BASELINE config C4 assembles its 1.25e7-row slab in place; nothing to mirror in the single-process reference).
 * Columns are renumbered into the extended local vector of halo mode: owned rows first, then the ghost entries, given
 * as nranges <= 16 contiguous GLOBAL index ranges in the order they occupy the ghost tail (what lanczos_amd.partition
 * plans for a stencil slab; follow with lz_set_halo).  rows_local == Nx Ny Nz: whole matrix, global columns.
 * potential_kind 0: none; 1: `potential` = rows_local host doubles (this rank's diagonal); 2: `potential` = 8 parameters
 * {eCore, rCore, eWell, rWell, power, Lx, Ly, Lz} of the deuteron hard core + well of 3Ddeuteron.py:51-61,
 * eCore exp(-(r/rCore)^power) - eWell exp(-(r/rWell)^power), evaluated ON THE DEVICE at np.linspace(-L/2, L/2, N)
 * coordinates (replaces the point-by-point host loop of Hamiltonian.py:38-43; device exp/pow differ from NumPy's in the
 * last bits, so the bit-exact-vs-reference builder keeps kind 1). */
int lz_build_stencil3d_block(lz_handle h, int Nx, int Ny, int Nz, int points, double T_factor, const double* weights4,
                             int potential_kind, const double* potential, int negate_T, int64_t row0, int64_t rows_local,
                             int nranges, const int64_t* ghost_start, const int64_t* ghost_len);
int lz_csr_info(lz_handle h, int64_t* rows, int64_t* nnz);
/* which SpMV kernel the current matrix + options select: 0 scalar CSR (LZ_FLAG_SPMV_SCALAR), 1 CSR-stream, 2 fixed-K
 * (stencils), 3 column-blocked two-phase (matrices without column locality, lz_spmv_pb.hip), 4 dense GEMV */
int lz_spmv_plan(lz_handle h, int* plan);
/* Row-class coding of a fixed-K (stencil) matrix (round 5): where the rows of the matrix fall into <= 256 classes up to translation -
 * the K offsets col - row, and for constant coefficients the K values too - the SpMV streams ONE BYTE per row (the class; + 8 bytes per
 * entry when only the offsets repeat) instead of 12 bytes per entry; found and verified on the device at lz_set_csr, same products in the
 * same order (same bits).  coding: 0 none, 1 offsets by class (values streamed), 2 offsets and values by class, 3 offsets and the
 * values OFF the diagonal by class with the diagonal's values streamed (8 more bytes per row: a constant-coefficient stencil plus a
 * potential - the reference's Hamiltonians); classes: how many. */
int lz_spmv_coding(lz_handle h, int* coding, int* classes);
/* download the handle's CSR matrix (sizes from lz_csr_info) */
int lz_get_csr(lz_handle h, int32_t* rowptr, int32_t* colidx, double* vals);
/* halo plan: for peer p (npeers of them) send x[send_idx[..]] (local row
 * indices, send_counts[p] of them, concatenated) and receive recv_counts[p]
 * doubles into the ghost region, in peer order. */
int lz_set_halo(lz_handle h, int npeers, const int32_t* peers, const int64_t* send_counts, const int32_t* send_idx,
                const int64_t* recv_counts);
/* all-gather plan: every rank owns `chunk` padded rows; x_full has world*chunk entries. */
int lz_set_allgather(lz_handle h, int64_t chunk);

/* Optional: allocate the device buffers of the coming run EARLY - the (n, rows_local) Krylov basis (with_ritz 0 or 1) and, where
 * it fits, the (rows_local, n) Ritz vectors (with_ritz 1; with_ritz 2: the Ritz vectors ONLY - what the helper thread asks for
 * while the solve already runs on the basis).  A hipMalloc of 16 GB costs 0.2 ms most of the time and 0.1 - 4 s now and then on
 * this platform (since round 5 buffers of this size are a reserved virtual range backed by physical chunks, which has not been
 * seen to stall: tools/probes/alloc_pattern_probe.hip); the
 * class mirror issues this call from a helper thread while it draws the start vector, hashes, validates and uploads the
 * matrix (Lanczos.py:85-104 does the same work in sequence), so lz_run / lz_ritz_vectors find their buffers ready.  It is the
 * ONE entry point that may run concurrently with another call on the same handle (lz_set_options / lz_set_tuning / lz_set_csr /
 * lz_set_dense / lz_build_stencil3d*): it touches only its own fields.  It must have returned before lz_run.  Never fails
 * for lack of memory (lz_run then allocates, and reports, itself); reserves nothing unless a quarter of the device stays free.
 * Single rank (rows_local = M). */
int lz_reserve(lz_handle h, int64_t rows_local, int n, int with_ritz);

/* ---- the Lanczos run --------------------------------------------------- *
 * Replaces Lanczos.py:104-119 (== IrrLanczos.py:222-238): basis allocation,
 * warm-up step and the Krylov loop with full re-orthogonalisation, with the
 * reference CPU branch's arithmetic (reorthogonalize, Lanczos.py:247-249).
 * v0_local: this rank's rows of the NORMALISED start vector (Lanczos.py:93-100
 * is done by the host mirror with NumPy's legacy RNG).  alpha_out[n],
 * beta_out[n-1] (n >= 2) receive the recurrence coefficients (replicated on all
 * ranks); beta follows the reference's indexing (beta[j-1] set at step j). */
int lz_run(lz_handle h, int n, const double* v0_local, double* alpha_out, double* beta_out);
/* ---- checkpoint / resume (SURVEY.md section 5 hook; nothing to mirror: the reference keeps no solver state on disk) ----
 * After lz_run(h, n1, ...) the state (V[0..n1), r, alpha[0..n1), beta[0..n1-1)) is everything the recurrence needs to go on at
 * step n1: lz_get_basis / lz_get_residual + the coefficient arrays are a checkpoint.  lz_get_residual: r = H v_{n1-1} -
 * alpha_{n1-1} v_{n1-1} - beta_{n1-2} v_{n1-2} (Lanczos.py:119 after the last step), this rank's rows.
 * lz_run_resume continues such a run to n > j0 steps in total: V_rows is (j0, rows_local) row-major with leading dimension
 * ldv_in, alpha_in has j0 and beta_in j0 - 1 entries.  Steps j0 .. n-1 run on the plain six-launch loop (every loop
 * structure produces the same bits), so alpha_out[n], beta_out[n-1] and the basis equal those of an uninterrupted
 * lz_run(h, n, ...) BIT FOR BIT (same options).  Not with LZ_FLAG_REORTH_PARTIAL / LZ_FLAG_ONE_REDUCE (LZ_ERR_STATE). */
int lz_get_residual(lz_handle h, double* r_local);
/* The same for LZ_FLAG_REORTH_PARTIAL's device-decided loop (engine 7; round 5): its checkpoint also carries the omega-recurrence
 * state - lz_get_omega_state after a run of j0 steps: 2 + (j0 + 2) + 3 (j0 + 1) doubles [||A|| estimate, "sweep the next vector too",
 * the norms hb[k] that formed V[k], three rows of omega] - and lz_run_resume_partial continues from it (j0 >= 2): the decision of step j0
 * is taken from the refreshed ||r||^2 exactly as the uninterrupted run took it, so coefficients, basis and sweep schedule of the
 * continued run equal an uninterrupted n-step run bit for bit.  lz_last_sweeps / lz_last_sweep_log then cover the steps j0 .. n-1. */
int lz_get_omega_state(lz_handle h, double* out, int64_t count);
int lz_run_resume_partial(lz_handle h, int n, int j0, const double* V_rows, int64_t ldv_in, const double* r_local, const double* alpha_in,
                          const double* beta_in, const double* omega_state, double* alpha_out, double* beta_out);
int lz_run_resume(lz_handle h, int n, int j0, const double* V_rows, int64_t ldv_in, const double* r_local, const double* alpha_in,
                  const double* beta_in, double* alpha_out, double* beta_out);
/* Krylov basis, row-major (n, rows_local): basis vector j is row j
 * (replaces cp.asnumpy(V.T), Lanczos.py:136; the mirror exposes the transposed view).
 * Result publication: device -> host copies of 192 MB and more (this call, lz_get_basis_block, lz_ritz_vectors with Y_out,
 * lz_get_ritz_vectors / lz_get_ritz_rows) go through a ring of pinned staging buffers (6 x 32 MB per handle, created at the
 * first such copy) emptied by host copy threads: 40-45 GB/s into fresh pageable NumPy memory instead of the 15-21 GB/s of a
 * plain hipMemcpy (tools/publish_probe.py).  Environment LZ_XFER_THREADS = number of copy threads (default 6; 0 = plain copy). */
int lz_get_basis(lz_handle h, double* V_out, int64_t ld);
/* the entries [row0, row0 + nrows) of every basis vector: (n, nrows) row-major with leading dimension ld >= nrows (a window
 * of V when the whole (n, rows_local) array is too large to move: BASELINE config C4, 160 GB) */
int lz_get_basis_block(lz_handle h, int64_t row0, int64_t nrows, double* V_out, int64_t ld);
/* Ritz back-transform Y = V_cols * S  (Lanczos.py:153-156: n GEMVs np.dot(V, S[:, i])):
 * S is (n, n) row-major (columns = eigenvectors of H_eff), Y_out is (rows_local, n) row-major or NULL (Y stays on the
 * device for the checks and is fetched later, whole or by rows).  Never fails for lack of room for Y: see lz_ritz_info. */
int lz_ritz_vectors(lz_handle h, const double* S, double* Y_out);
/* copy the Y of the last lz_ritz_vectors call to the host: (rows_local, n) row-major */
int lz_get_ritz_vectors(lz_handle h, double* Y_out);
/* rows [row0, row0 + nrows) of that Y: (nrows, n) row-major.  This is how H_eigvecs (Lanczos.py:60-66) is served when the
 * whole (M, n) array fits neither beside the basis on the device nor on the host. */
int lz_get_ritz_rows(lz_handle h, int64_t row0, int64_t nrows, double* Y_out);
/* How the last lz_ritz_vectors call is held.  *chunk_rows = 0: Y is resident on the device; > 0: CHUNKED mode - a second
 * rows_local x n array does not fit beside the basis (BASELINE config C4 on one GPU: 160 GB + 160 GB), so the device keeps
 * S and a buffer of that many rows, and lz_get_ritz_rows / lz_get_ritz_vectors / lz_ritz_gram / lz_ritz_quality re-form the
 * rows (or column batches) they need from the basis: a 16-row tile of Y depends only on the same 16 columns of V.
 * lz_set_tuning(h, 16, rows) forces the chunked mode (tests).
 * clock4 (may be NULL; zeros unless one of the persistent kernels ran - n <= 200 with enough rows): {shader clock in MHz while
 * the kernel ran (s_memtime cycles / s_memrealtime ticks of the constant 100 MHz counter over workgroup 0's trip), shader
 * cycles workgroup 0 needed per 16-row tile of one of its waves, the MFMA issue floor for that in cycles (MFMAs per tile
 * and SIMD x 64), 16-row tiles walked by that wave}. */
int lz_ritz_info(lz_handle h, int64_t* chunk_rows, double* clock4);
/* Device-side versions of the two checks get_H_eigs runs on Y (Lanczos.py:157-158,
 * 288-323): column norms (n) and the (n, n) Gram matrix Y^T Y, computed on the
 * device-resident Y of the last lz_ritz_vectors call (summed over ranks). */
int lz_ritz_gram(lz_handle h, double* gram_out);
/* In-kernel clock record of the last lz_ritz_gram when it ran the accumulator-stationary symmetric kernel (48 <= n <= 208, at
 * least 4096 rows; zeros otherwise): {shader clock in MHz while workgroup 0 ran, shader cycles its wave 0 needed per k-step
 * (4 rows of Y), the MFMA issue floor of that (MFMAs per k-step and SIMD x 64 cycles), k-steps walked}. */
int lz_gram_info(lz_handle h, double* info4);
/* y_i = A * Y[:, i] residual check used by print_good_eigs (Lanczos.py:166-185):
 * out[i] = (A y_i . y_i)^2 / (||A y_i||^2), for all n columns.  One rank: one fused kernel over the CSR matrix.
 * Row-block partition (world > 1): collective; every Ritz vector is exchanged and multiplied like a Lanczos vector
 * (CSR or dense), the 2 n sums travel in one all-reduce, every rank receives the same n values. */
int lz_ritz_quality(lz_handle h, double* out);
int lz_get_timings(lz_handle h, lz_timings* out);
/* The collectives behind lz_timings.launches[LZ_K_COMM] of the interval the LAST lz_get_timings closed, by kind: all-reduces
 * (alpha / ||r||^2 / coefficient sums) and SpMV-input exchanges (halo send/recv group or all-gather). */
int lz_comm_counts(lz_handle h, int64_t* allreduces, int64_t* exchanges);
/* number of steps of the last lz_run that ran the re-orthogonalisation sweep (== n without LZ_FLAG_REORTH_PARTIAL) */
int lz_last_sweeps(lz_handle h, int* sweeps);
/* log[j] = 1 if step j of the last lz_run ran the sweep, j < n <= its number of steps (all 1 for the full-sweep loops; the device's own
 * record for the device-decided partial loops, engines 7 and 8; LZ_ERR_ARG after the host-decided loop of knob 18 = 1, which keeps none).
 * tests/ replay Simon's recurrence (and the look-ahead of engine 8) on the host from alpha / beta and hold the device to it. */
int lz_last_sweep_log(lz_handle h, int* log, int n);
/* LZ_FLAG_REORTH_PARTIAL | LZ_FLAG_ONE_REDUCE only (else 0): vectors of the last lz_run whose exact omega-recurrence estimate
 * exceeded sqrt(eps) although the one-step look-ahead gate had not swept them (lz_set_tuning(h, 20, kappa) sets the look-ahead's
 * safety factor, default 4). */
int lz_last_sweep_misses(lz_handle h, int* misses);
/* How the last lz_run was executed (choose_loop, lz_loops.hip).  0: six launches per step (also: the host-decided partial
 * re-orthogonalisation loop, every kernel A/B arm, more than 4e6 rows per rank).  7: LZ_FLAG_REORTH_PARTIAL with the decision on
 * the device (round 4, the default of that flag): the sweep kernels of every step are enqueued and return at once when the
 * device-side omega-recurrence says no sweep is due; with an ELL-ordered stencil matrix the SpMV forms r / beta itself.  2: the fused-launch path of small problems (a vector of at most eight
 * pass-1 slices, one rank, fused-norm mode): the second-stage reductions and the three-term recurrence ride in the prologue
 * of their consumer kernels - three launches per step, bit-identical results.  3: up to 4e6 rows per rank in fused-norm mode
 * with the full sweep: the three-term recurrence rides in the prologue of the next step's pass 1 (five launches per step,
 * bit-identical; not with LZ_FLAG_OVERLAP_HALO).  lz_set_tuning(h, 15, 1) selects 0 in both cases.  6: LZ_FLAG_ONE_REDUCE.
 * 8: LZ_FLAG_ONE_REDUCE | LZ_FLAG_REORTH_PARTIAL (one all-reduce per step, look-ahead sweep decision on the device).
 * 5: a one-reduce run whose cancellation guard fired and that was repeated on the default loop.  1 / 4: the one-kernel and
 * one-launch-per-step engines of the kernel-bench build (retired from the product library in round 3: bit-identical, not
 * faster - LAB_NOTEBOOK.md section 4). */
int lz_last_engine(lz_handle h, int* engine);
/* Host <-> device synchronisations lz_run made between its first and its last launch (the final wait for alpha / beta is not
 * counted).  0 for every loop on one rank and over RCCL - including, since round 4, the partial re-orthogonalisation mode
 * (engine 7: Simon's omega-recurrence and the sweep decision live on the device; lz_set_tuning(h, 18, 1) selects the former
 * host-decided loop, engine 0, which reads two scalars back per step).  The host-staged collective backend (tests) counts two
 * per collective. */
int lz_last_host_syncs(lz_handle h, int64_t* syncs);

/* ---- single steps (unit parity tests drive the kernels one by one) ------ */
/* allocate a zeroed basis of n rows + r (what lz_run does first, Lanczos.py:104-107) */
int lz_basis_alloc(lz_handle h, int n);
int lz_basis_set_row(lz_handle h, int j, const double* row_local); /* host -> V[j] */
/* rows j0 .. j0 + count - 1 in one strided copy; rows_local: count rows of rows_local doubles, ld doubles apart (the (n, M) array
 * the static Lanczos.reorthogonalize(V, j) is handed, Lanczos.py:233) */
int lz_basis_set_rows(lz_handle h, int j0, int count, const double* rows_local, int64_t ld);
int lz_basis_get_row(lz_handle h, int j, double* row_local);       /* V[j] -> host */
int lz_r_set(lz_handle h, const double* r_local);
int lz_r_get(lz_handle h, double* r_local);
/* r = A V[j]; *dot_out = V[j] . r   (Lanczos.py:116-118) */
int lz_step_spmv(lz_handle h, int j, double* dot_out);
/* V[j] = r / ||r|| (if scale != 0; *beta_out = ||r||) then the reference's
 * reorthogonalize(V, j) over rows [0, nrows): c = V V[j]; V[j] = 2 V[j] - c^T V
 * (Lanczos.py:112-115, 247-249).  c_out[nrows] receives the coefficients. */
int lz_step_reorth(lz_handle h, int j, int nrows, int scale, double* beta_out, double* c_out);
/* r = r - alpha V[j] - beta V[jm1] (jm1 < 0: term skipped); *norm2_out = ||r||^2 (Lanczos.py:119,112) */
int lz_step_three_term(lz_handle h, int j, int jm1, double alpha, double beta, double* norm2_out);
/* y = A x on host vectors (length rows_local / ncols_ext handled internally; single rank) */
int lz_spmv_host(lz_handle h, const double* x, double* y);

/* ---- two-sided (bi-orthogonal) Lanczos: the Irregular copy's execute_Lanczos --------------------------------
 * Replaces Python/Irregular/IrrLanczos.py:77-187 (driver loop) and :408-441 (bireorthogonalize, default branch).
 * Single rank, CSR only.  Four (n, rows) bases live on the device: 0 = q (published as V: lz_get_basis /
 * lz_ritz_vectors read it), 1 = p, 2 = q_basis, 3 = p_basis (the orthonormalised copies the reference projects on). */

/* HT = csr_matrix(H.transpose()) (IrrLanczos.py:92/96) as sorted CSR of the same square shape as lz_set_csr's matrix.
 * rowptr == NULL: H is symmetric, H^T x runs on H itself. */
int lz_set_csr_transpose(lz_handle h, int64_t nnz, const int32_t* rowptr, const int32_t* colidx, const double* vals);
/* The whole run: q0, p0 are the start pair already scaled so that q0 . p0 = +-1 (IrrLanczos.py:104-106, host side).
 * alpha_out[n], beta_out[n-1], gamma_out[n-1] (IrrLanczos.py:119-121).  n >= 2. */
int lz_run_two_sided(lz_handle h, int n, const double* q0, const double* p0, double* alpha_out, double* beta_out,
                     double* gamma_out);
/* Step API for unit parity tests and for the static IrrLanczos.bireorthogonalize(V1, V2, q_basis, p_basis, j):
 * allocate the four bases (zeroed), move rows, run the default branch of bireorthogonalize on row j (j >= 0; with j = 0 there
 * is nothing to project on: the pair is rescaled to q.p = +-1 and seeds the two orthonormal bases, IrrLanczos.py:418-438). */
int lz_bi_alloc(lz_handle h, int n);
int lz_bi_set_row(lz_handle h, int which, int j, const double* row);
int lz_bi_get_row(lz_handle h, int which, int j, double* row);
int lz_step_bireorth(lz_handle h, int j);
/* The other branch of the same static method, bireorthogonalize(..., mem_safe=True) (IrrLanczos.py:398-407; no caller in the
 * reference): V1[j] -= sum_i (V1[j].V2[i] / V2[i].V2[i]) V2[i], then V2[j] against the rows of V1 likewise - one sweep over
 * ALL n rows with the reference's uu[j:] = 1, uv[j] = 0; bases 0 (V1) and 1 (V2) only, q_basis / p_basis untouched.  j >= 0. */
int lz_step_bireorth_mem_safe(lz_handle h, int j);

#ifdef __cplusplus
}
#endif
#endif /* LANCZOS_HIP_H */
