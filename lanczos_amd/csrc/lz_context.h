// Shared host-side state of the C ABI (lz_api.hip, lz_loops.hip, lz_matrix.hip, lz_ritz.hip, lz_twosided_api.hip): the handle,
// the error macros, the per-launch profiling scope and the helpers those files call across each other.  Internal: nothing
// here is part of include/lanczos_hip.h.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <system_error>
#include <thread>

#include "lz_internal.h"

using lz::CsrDev;
using lz::QtwPlan;
using lz::Xfer;

struct EventRec {
  int cls;
  hipEvent_t a, b;
};

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

struct lz_context {
  int dev = 0;
  hipStream_t stream = nullptr;
  std::string err;
  std::string name;
  int flags = 0;
  int tune[24] = {0};  // A/B knobs, see lz_set_tuning

  // partition
  int64_t Mg = 0, row0 = 0, rows = 0, ncols_ext = 0;
  int64_t rows_pad = 0;  // owned rows padded to 32 doubles
  int64_t ldv = 0;       // stride between basis rows (>= rows_pad, + ghost tail in halo mode)

  // matrix
  int kind = 0;  // 0 none, 1 csr, 2 dense
  CsrDev csr;
  double* d_dense = nullptr;
  int64_t dense_lda = 0;  // row stride of the device copy (even: 16-byte aligned rows)

  // basis and work vectors
  int n = 0;
  double* d_V = nullptr;
  double* d_r = nullptr;
  double* d_r2 = nullptr;     // fused small-problem path: r of the three-term recurrence (d_r then holds the SpMV output)
  double* d_alpha = nullptr;  // n
  double* d_beta = nullptr;   // n
  double* d_c = nullptr;      // n + 1
  double* d_nrm2 = nullptr;   // 2 : [0] = ||r||^2
  double* d_part = nullptr;   // partials
  size_t part_cap = 0;
  double* d_xtmp = nullptr;   // lz_spmv_host scratch
  QtwPlan qplan;

  // two-sided Lanczos (IrrLanczos.py:77-187): H^T, the three extra bases P / Qb / Pb (Q is d_V), s, gamma, scalars
  CsrDev csrT;
  bool has_T = false;      // false: H declared symmetric, H^T x runs on csr
  bool T_declared = false; // lz_set_csr_transpose was called for the current matrix
  double* d_B3 = nullptr;  // 3 * n * ldv
  int bi_n = 0;
  double* d_s = nullptr;
  double* d_gamma = nullptr;  // n + 1
  double* d_bi = nullptr;     // [0..3] raw sums S, [4..6] factors f

  // Ritz vectors: Y = V^T-layout x S.  Resident (d_Y holds all y_rows rows) when it fits beside the basis; otherwise
  // CHUNKED: d_Y is a y_chunk-row buffer, the padded S stays on the device (d_S) and every consumer (Gram matrix, row
  // fetch, quality sums) re-forms the rows it needs - a 16-row tile of Y depends only on the same 16 columns of V.
  double* d_Y = nullptr;
  int64_t y_rows = 0;
  int y_n = 0;
  double* d_S = nullptr;
  int s_npad = 0;
  uint64_t* d_rclk = nullptr;  // in-kernel clock record of the last S-stationary back-transform (lz_ritz_info)
  double* d_gram = nullptr;    // scratch of lz_ritz_gram (K-slice partials + per-chunk slices + G), kept between calls
  size_t gram_cap = 0;
  uint64_t* d_gclk = nullptr;  // in-kernel clock record of the last symmetric Gram kernel (lz_gram_info)
  bool gram_sym_last = false;
  bool y_chunked = false;
  int64_t y_chunk = 0;     // rows per chunk (multiple of 16)
  int64_t y_cap = 0;       // doubles allocated behind d_Y

  // communication
  int world = 1, rank = 0;
  int comm_kind = 0;  // 0 none, 1 rccl, 2 host callbacks
  ncclComm_t comm = nullptr;
  lz_host_allreduce_fn h_ar = nullptr;
  lz_host_exchange_fn h_ex = nullptr;
  lz_host_allgather_fn h_ag = nullptr;
  void* h_user = nullptr;
  std::vector<double> hbuf_a, hbuf_b;
  int xmode = 0;  // 0 none, 1 halo, 2 allgather
  std::vector<int32_t> peers;
  std::vector<int64_t> scount, rcount, soff, roff;
  std::vector<int64_t> sstart;  // >= 0: the peer's send list is the contiguous run x[sstart .. sstart+scount) (stencil faces)
  bool all_contig = false;
  // LZ_FLAG_OVERLAP_HALO: the boundary positions of V[j] are updated first, their halo exchange runs on `cstream`
  // while the compute stream updates the interior; the SpMV waits for `e_halo`.
  hipStream_t cstream = nullptr;
  hipEvent_t e_bnd = nullptr, e_halo = nullptr;
  int halo_inflight_j = -1;
  std::vector<std::pair<int64_t, int64_t>> bnd_ranges, int_ranges;  // double2 position ranges of a basis row
  int64_t total_send = 0, total_recv = 0;
  int64_t n_allreduce = 0, n_exchange = 0;            // collectives issued since the last lz_get_timings ...
  int64_t n_allreduce_last = 0, n_exchange_last = 0;  // ... and in the interval that call closed (lz_comm_counts)
  int32_t* d_send_idx = nullptr;
  double* d_sendbuf = nullptr;
  int64_t ag_chunk = 0;
  double* d_xfull = nullptr;

  // timing
  std::vector<EventRec> events;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> free_events;
  hipEvent_t run_a = nullptr, run_b = nullptr;
  bool run_timed = false;
  int last_sweeps = 0;
  std::vector<int> sweep_log;  // per step of the last lz_run: 1 = the re-orthogonalisation sweep ran (lz_last_sweep_log; device-decided loops only, else all 1)
  int last_misses = 0;  // one-reduce partial loop: vectors whose exact omega exceeded sqrt(eps) although the look-ahead gate had not swept them
  int last_engine = 0;  // which loop ran last (enum Loop)
  int r_state = 0;      // what d_r holds after the last run: 0 nothing usable, 1 the residual entering step n, 2 y = A v_{n-1} (three-term pending)
  Xfer* xfer = nullptr;        // staging ring of the large device -> host copies (lz_xfer.hip), created at the first one
  double* h_pinned = nullptr;  // 8 pinned doubles for the per-step scalar read-back of the host-decided partial-reorth loop (tune[18] == 1)
  double* d_om = nullptr;      // device-resident partial re-orthogonalisation: omega-recurrence state (omega_state_doubles)
  int* d_omi = nullptr;        //   ... gate of the coming step, sweep count, per-step sweep log (omega_state_ints)
  int om_n = 0;
  int om_run_n = 0;            // n of the last device-decided partial run: the layout of d_om (lz_get_omega_state)
  int64_t host_syncs = 0;      // host <-> device synchronisations between the first and the last launch of the last lz_run
  // lz_reserve (may be called from a second host thread while this one prepares the matrix): device buffers for the basis and
  // the Ritz vectors of the coming run, adopted by basis_alloc / lz_ritz_vectors.  Only these fields are touched by it.
  std::mutex res_mu;
  double* res_V = nullptr;
  size_t res_V_count = 0;
  double* res_Y = nullptr;
  size_t res_Y_count = 0;
  bool prof_iter = true;  // false while lz_run skips an iteration under profile sampling (tune[7])
  lz_timings acc;
};

namespace lz {
namespace api {

extern std::string g_create_error;
extern RcclApi g_rccl;
extern std::string g_rccl_path;
const char* load_rccl();  // nullptr, or why RCCL could not be loaded

int fail(lz_handle h, int code, const std::string& msg);

#define LZ_HIP(h, call)                                                                                   \
  do {                                                                                                    \
    hipError_t e_ = (call);                                                                               \
    if (e_ != hipSuccess)                                                                                 \
      return fail(h, e_ == hipErrorOutOfMemory ? LZ_ERR_NOMEM : LZ_ERR_HIP,                               \
                  std::string(#call) + ": " + hipGetErrorString(e_));                                     \
  } while (0)

#define LZ_NCCL(h, call)                                                                                  \
  do {                                                                                                    \
    ncclResult_t r_ = (call);                                                                             \
    if (r_ != ncclSuccess) return fail(h, LZ_ERR_COMM, std::string(#call) + ": " + g_rccl.GetErrorString(r_)); \
  } while (0)

#define LZ_TRY(expr)          \
  do {                        \
    int rc_ = (expr);         \
    if (rc_ != LZ_OK) return rc_; \
  } while (0)

int64_t skew_stride(lz_handle h, int64_t ld);
int check_launch(lz_handle h, const char* what);

// large buffers (>= kBigMinBytes: the basis, the Ritz vectors, a dense matrix): a reserved virtual range backed by physical chunks
// instead of hipMalloc, whose 16 GB calls stall for seconds now and then on this pool (lz_api.hip); big_free releases either kind
constexpr size_t kBigMinBytes = (size_t)256 << 20;
hipError_t big_alloc(int dev, void** out, size_t bytes);
hipError_t big_free(void* p);
void big_vmm_disable();  // from now on big_alloc uses hipMalloc (process-wide)

template <class T>
inline int dev_free(lz_handle h, T*& p) {
  if (p) {
    LZ_HIP(h, big_free(p));
    p = nullptr;
  }
  return LZ_OK;
}

template <class T>
inline int dev_alloc(lz_handle h, T*& p, size_t count) {
  LZ_TRY(dev_free(h, p));
  void* q = nullptr;
  LZ_HIP(h, big_alloc(h->dev, &q, std::max<size_t>(count, 1) * sizeof(T)));
  p = static_cast<T*>(q);
  return LZ_OK;
}

int upload(lz_handle h, void* dst, const void* src, size_t bytes);
int upload2d(lz_handle h, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height);

// per-launch accounting (algorithmic bytes / flops per kernel class), optional hipEvent bracket and roctx range
struct Scope {
  lz_handle h;
  int cls;
  hipEvent_t a = nullptr, b = nullptr;
  bool on;
  bool marked = false;
  Scope(lz_handle h_, int cls_, double bytes, double flops);
  ~Scope();
};
int drain_events(lz_handle h);

// collectives (RCCL or host-staged)
int comm_allreduce(lz_handle h, double* dbuf, int64_t count);
int comm_exchange_x(lz_handle h, int j, const double** x_out);

// the individual steps of the recurrence (lz_loops.hip)
int ensure_part(lz_handle h, size_t need);
double spmv_bytes(lz_handle h, bool ell = false);  // ell: the launch takes the ELL copy whatever the plain SpMV does (the partial loop's fused SpMV)
double spmv_flops(lz_handle h);
int step_spmv(lz_handle h, int j, double* alpha_dst = nullptr, bool reduce = true, int* np_out = nullptr);
int step_reorth(lz_handle h, int j, int nrows, bool scale, int beta_idx, bool in_run_loop = false);
int step_three_term(lz_handle h, int j, int jm1, const double* d_alpha, const double* d_beta, bool need_norm = true);
size_t fused_coff(lz_handle h);
size_t onered_part_off(lz_handle h);
int breakdown_status(lz_handle h, int n, const double* alpha_out, const double* beta_out);

// basis (lz_api.hip)
int basis_alloc(lz_handle h, int n, int zero_rows);
int require_basis(lz_handle h, int j);
inline size_t y_doubles(int64_t rows, int n) { return (size_t)(round_up(rows, 16) + 16) * (size_t)n + 64; }

// matrix setup (lz_matrix.hip)
int fill_csr_meta(lz_handle h, CsrDev& A, const int32_t* rowptr_host, int64_t rows_local, int64_t ncols_ext, int64_t nnz, int fixed_k,
                  int max_nnz);
int upload_csr(lz_handle h, CsrDev& A, const char* who, int64_t rows, int64_t ncols, int64_t nnz, const int32_t* rowptr,
               const int32_t* colidx, const double* vals, int* fixed_k_out, int* max_nnz_out);

}  // namespace api
}  // namespace lz
