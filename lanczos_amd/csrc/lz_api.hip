// C ABI of liblanczos_hip.so, part 1: library / device, the handle, options, communicators, the basis and the single-step entry
// points; plus the helpers every part shares (error reporting, per-launch profiling scope, collectives).  The run loops are
// in lz_loops.hip, matrix setup in lz_matrix.hip, Ritz vectors / Gram / quality in lz_ritz.hip, the two-sided variant in
// lz_twosided_api.hip.  See include/lanczos_hip.h for the contract and the reference call sites each entry point replaces.
#include <atomic>
#include <deque>
#include <unordered_map>

#include "lz_context.h"

using namespace lz;

namespace lz {
namespace api {

std::string g_create_error;


RcclApi g_rccl;

// RCCL must pair with the HIP runtime THIS library is bound to: its streams and device pointers are handed to
// ncclAllReduce & co.  A process can hold two ROCm trees (PyTorch bundles its own libamdhip64 / libhsa-runtime64 /
// librccl under torch/lib, and its NEEDED names carry no version, so they never match the system libraries' sonames):
// `dlopen("librccl.so.1")` by bare soname then returns whichever copy happens to be mapped already - in round 1 that
// was torch's RCCL (bound to torch's HIP runtime) driven with streams of the system runtime, and the process aborted in
// free() at teardown.  So: take librccl from the directory of the libamdhip64 that resolves OUR
// hipMalloc; LZ_RCCL_PATH overrides; the system path is the last resort.  RTLD_LOCAL: symbols are only reached through
// dlsym on this handle, nothing is interposed.
std::string g_rccl_path;
const char* load_rccl() {
  if (g_rccl.lib) return nullptr;
  void* lib = nullptr;
  std::vector<std::string> cand;
  if (const char* e = getenv("LZ_RCCL_PATH")) cand.push_back(e);
  Dl_info info;
  if (dladdr(reinterpret_cast<const void*>(static_cast<hipError_t (*)(void**, size_t)>(&hipMalloc)), &info) && info.dli_fname) {
    std::string dir(info.dli_fname);
    const size_t slash = dir.rfind('/');
    if (slash != std::string::npos) {
      dir.resize(slash);
      cand.push_back(dir + "/librccl.so.1");
      cand.push_back(dir + "/librccl.so");
    }
  }
  cand.push_back("/opt/rocm/lib/librccl.so.1");
  for (const auto& c : cand) {
    lib = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (lib) {
      g_rccl_path = c;
      break;
    }
  }
  if (!lib) return "cannot dlopen librccl next to the HIP runtime in use (set LZ_RCCL_PATH)";
#define LZ_SYM(field, name)                                          \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(lib, name)); \
  if (!g_rccl.field) return "missing RCCL symbol " name;
  LZ_SYM(GetUniqueId, "ncclGetUniqueId")
  LZ_SYM(CommInitRank, "ncclCommInitRank")
  LZ_SYM(CommDestroy, "ncclCommDestroy")
  LZ_SYM(AllReduce, "ncclAllReduce")
  LZ_SYM(AllGather, "ncclAllGather")
  LZ_SYM(Send, "ncclSend")
  LZ_SYM(Recv, "ncclRecv")
  LZ_SYM(GroupStart, "ncclGroupStart")
  LZ_SYM(GroupEnd, "ncclGroupEnd")
  LZ_SYM(GetErrorString, "ncclGetErrorString")
#undef LZ_SYM
  g_rccl.lib = lib;
  return nullptr;
}

int fail(lz_handle h, int code, const std::string& msg) {
  if (h)
    h->err = msg;
  else
    g_create_error = msg;
  return code;
}

// Row stride of the basis.  Every streaming kernel has several rows in flight at the SAME column offset, so a stride
// that is a multiple of a large power of two (M = 2^20, 160^3 = 2^15 * 125, ...) lands them on the same HBM channels.
// Making stride / 256 B odd spreads consecutive rows over the channel interleave; costs at most 256 B per row.
// tune[12] = 1 disables the skew (A/B).
int64_t skew_stride(lz_handle h, int64_t ld) {
  if (h->tune[12] == 1) return ld;
  return ((ld / kPadDoubles) & 1) ? ld : ld + kPadDoubles;
}

int check_launch(lz_handle h, const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(h, LZ_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
  return LZ_OK;
}

// host -> device on the handle's stream.  The runtime's own path: a resident pageable source is pinned in place and read at
// the link's rate (57 GB/s measured); a staged pipeline like lz_xfer.hip's was built and measured slower (see lz_xfer.hip).
int upload(lz_handle h, void* dst, const void* src, size_t bytes) {
  LZ_HIP(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
  return LZ_OK;
}
int upload2d(lz_handle h, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height) {
  LZ_HIP(h, hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyHostToDevice, h->stream));
  return LZ_OK;
}

namespace {
// ---- roctx ranges (opt-in: LZ_ROCTX=1) --------------------------------------
// Host-side phase markers for `rocprofv3 --marker-trace`: one range per kernel class around its launches.  The marker
// library is dlopen'ed on first use, like RCCL; nothing is linked and nothing happens without the environment variable.
struct RoctxApi {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  bool tried = false;
};
RoctxApi g_roctx;
const char* const kClassNames[LZ_K_COUNT] = {"lz:spmv", "lz:qtw", "lz:update", "lz:three_term", "lz:final", "lz:comm", "lz:ritz"};
bool roctx_on() {
  if (!g_roctx.tried) {
    g_roctx.tried = true;
    const char* e = getenv("LZ_ROCTX");
    if (e && e[0] == '1') {
      void* lib = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
      if (!lib) lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
      if (lib) {
        g_roctx.push = reinterpret_cast<int (*)(const char*)>(dlsym(lib, "roctxRangePushA"));
        g_roctx.pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
      }
    }
  }
  return g_roctx.push && g_roctx.pop;
}
}  // namespace

// ---- profiling events ---------------------------------------------------
Scope::Scope(lz_handle h_, int cls_, double bytes, double flops) : h(h_), cls(cls_) {
  if (roctx_on()) {
    g_roctx.push(kClassNames[cls]);
    marked = true;
  }
  h->acc.bytes[cls] += bytes;
  h->acc.flops[cls] += flops;
  h->acc.launches[cls] += 1;
  on = (h->flags & LZ_FLAG_PROFILE) != 0 && h->prof_iter;
  if (on) {
    h->acc.timed_bytes[cls] += bytes;
    h->acc.timed_launches[cls] += 1;
    if (!h->free_events.empty()) {
      a = h->free_events.back().first;
      b = h->free_events.back().second;
      h->free_events.pop_back();
    } else {
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
        on = false;
        return;
      }
    }
    hipEventRecord(a, h->stream);
  }
}
Scope::~Scope() {
  if (marked) g_roctx.pop();
  if (on) {
    hipEventRecord(b, h->stream);
    h->events.push_back({cls, a, b});
  }
}

// ---- large device buffers (the Krylov basis, the Ritz vectors): virtual range + physical chunks -------------------------------
// hipMalloc of the headline's 16 GB basis is what the first execute_Lanczos of an object used to wait for: 0.2 ms most of the
// time, but 0.1 - 4.3 s every few calls on this pool (tools/probes/alloc_pattern_probe.hip, profiles/r05/alloc_pattern_probe.jsonl:
// 4 of 18 hipMallocs of 16 GB stalled for 1 - 4 s, none of 11 reserve + create + map sequences took more than 3 ms, first and
// second touch of the mapped range at the same rate as hipMalloc'ed memory).  So buffers of 256 MB and more are a reserved virtual
// range backed by 2 GB physical chunks (hipMemAddressReserve / hipMemCreate / hipMemMap / hipMemSetAccess); kernels see one
// contiguous pointer.  Any failure on that path falls back to hipMalloc (LZ_NO_VMM=1 forces that).  A registry maps the base
// pointer to its chunks so that dev_free / lz_destroy release either kind through big_free.
namespace {
struct BigBuf {
  size_t bytes = 0;
  std::vector<hipMemGenericAllocationHandle_t> chunks;
  std::vector<size_t> sizes;
};
std::mutex g_big_mu;
std::unordered_map<void*, BigBuf> g_big;
std::atomic<bool> g_vmm_off{false};  // set by lz_comm_init_rccl(world > 1), see there
constexpr size_t kBigChunk = (size_t)2 << 30;
constexpr size_t kBigAlign = (size_t)2 << 20;

// Freed address ranges in quarantine: reserved again (unmapped, no memory behind them) so that no new buffer lands on them.
struct Quarantined {
  void* va;
  size_t bytes;
};
std::deque<Quarantined> g_quarantine;  // (under g_big_mu)
size_t g_quarantine_bytes = 0;
constexpr size_t kQuarantineBytes = (size_t)16 << 40;  // 16 TB of addresses (nothing behind them) are held back before the oldest are let go

void big_release(void* va, BigBuf& b) {
  size_t off = 0;
  for (size_t k = 0; k < b.chunks.size(); ++k) {
    (void)hipMemUnmap(static_cast<char*>(va) + off, b.sizes[k]);
    (void)hipMemRelease(b.chunks[k]);
    off += b.sizes[k];
  }
  // On this stack (ROCm 7.2, gfx950) the physical memory only comes back with hipMemAddressFree, and a fresh mapping that lands on
  // just-freed addresses intermittently has holes - kernels fault on pages that belong to the NEW buffer (tools/spmv_coding_probe.py:
  // 3 of 4 runs, the faulting page inside both the freed and the new buffer; never with LZ_NO_VMM=1; profiles/r05/
  // vmm_address_reuse_fault.txt).  So the range is freed - the memory returns - and at once reserved again at the same address, with
  // nothing mapped: later buffers cannot land on it.  Up to 16 TB of such addresses are held back (a thousand 16 GB bases); the oldest are let go beyond that.
  (void)hipMemAddressFree(va, b.bytes);
  static const bool reuse_ok = getenv("LZ_VMM_ADDRESS_REUSE") != nullptr;  // (the first form of this round, to reproduce the fault)
  if (reuse_ok) return;
  void* q = nullptr;
  if (hipMemAddressReserve(&q, b.bytes, kBigAlign, va, 0) == hipSuccess && q) {
    if (q != va) {  // the hint was not honoured: this reservation protects nothing
      (void)hipMemAddressFree(q, b.bytes);
      if (getenv("LZ_DEBUG_VMM")) fprintf(stderr, "[vmm] quarantine of %p (%zu bytes) not honoured: got %p\n", va, b.bytes, q);
      return;
    }
    std::vector<Quarantined> old;
    {
      std::lock_guard<std::mutex> lk(g_big_mu);
      g_quarantine.push_back({q, b.bytes});
      g_quarantine_bytes += b.bytes;
      while (g_quarantine_bytes > kQuarantineBytes && g_quarantine.size() > 1) {
        old.push_back(g_quarantine.front());
        g_quarantine_bytes -= g_quarantine.front().bytes;
        g_quarantine.pop_front();
      }
    }
    for (const Quarantined& o : old) (void)hipMemAddressFree(o.va, o.bytes);
  } else {
    (void)hipGetLastError();
    if (getenv("LZ_DEBUG_VMM")) fprintf(stderr, "[vmm] quarantine of %p (%zu bytes): reservation failed\n", va, b.bytes);
  }
}
}  // namespace

hipError_t big_alloc(int dev, void** out, size_t bytes) {
  *out = nullptr;
  static const bool no_vmm = getenv("LZ_NO_VMM") != nullptr;
  if (!no_vmm && !g_vmm_off.load() && bytes >= kBigMinBytes) {
    const size_t total = (bytes + kBigAlign - 1) / kBigAlign * kBigAlign;
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    hipMemAccessDesc acc;
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = dev;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    void* va = nullptr;
    if (hipMemAddressReserve(&va, total, kBigAlign, nullptr, 0) == hipSuccess && va) {
      BigBuf b;
      b.bytes = total;
      bool ok = true, oom = false;
      for (size_t off = 0; off < total && ok; off += kBigChunk) {
        const size_t sz = std::min(kBigChunk, total - off);
        hipMemGenericAllocationHandle_t hh;
        hipError_t e = hipMemCreate(&hh, sz, &prop, 0);
        if (e != hipSuccess) {
          ok = false;
          oom = e == hipErrorOutOfMemory;
          break;
        }
        b.chunks.push_back(hh);
        b.sizes.push_back(sz);
        if (hipMemMap(static_cast<char*>(va) + off, sz, 0, hh, 0) != hipSuccess) {
          (void)hipMemRelease(hh);
          b.chunks.pop_back();
          b.sizes.pop_back();
          ok = false;
          break;
        }
        if (hipMemSetAccess(static_cast<char*>(va) + off, sz, &acc, 1) != hipSuccess) ok = false;  // (mapped: big_release unmaps it)
      }
      if (ok) {
        std::lock_guard<std::mutex> lk(g_big_mu);
        g_big.emplace(va, std::move(b));
        *out = va;
        return hipSuccess;
      }
      big_release(va, b);
      (void)hipGetLastError();
      if (oom) return hipErrorOutOfMemory;  // hipMalloc would say the same
    } else {
      (void)hipGetLastError();
    }
  }
  return hipMalloc(out, bytes);
}

void big_vmm_disable() { g_vmm_off.store(true); }

hipError_t big_free(void* p) {
  if (!p) return hipSuccess;
  BigBuf b;
  {
    std::lock_guard<std::mutex> lk(g_big_mu);
    auto it = g_big.find(p);
    if (it == g_big.end()) return hipFree(p);
    b = std::move(it->second);
    g_big.erase(it);
  }
  // hipFree synchronises the device before it releases; an unmap does not: nothing may still run on the range
  hipError_t e = hipDeviceSynchronize();
  big_release(p, b);
  return e;
}

int drain_events(lz_handle h) {
  if (h->events.empty() && !h->run_timed) return LZ_OK;
  LZ_HIP(h, hipStreamSynchronize(h->stream));
  for (auto& e : h->events) {
    float ms = 0.f;
    LZ_HIP(h, hipEventElapsedTime(&ms, e.a, e.b));
    h->acc.ms[e.cls] += ms;
    h->free_events.push_back({e.a, e.b});
  }
  h->events.clear();
  if (h->run_timed) {
    float ms = 0.f;
    LZ_HIP(h, hipEventElapsedTime(&ms, h->run_a, h->run_b));
    h->acc.total_ms += ms;
    h->run_timed = false;
  }
  return LZ_OK;
}

// ---- collectives --------------------------------------------------------
int comm_allreduce(lz_handle h, double* dbuf, int64_t count) {
  if ((h->world <= 1 && !(h->tune[6] && h->comm_kind)) || count <= 0) return LZ_OK;
  Scope sc(h, LZ_K_COMM, 8.0 * count, 0);
  h->n_allreduce += 1;
  if (h->comm_kind == 1) {
    LZ_NCCL(h, g_rccl.AllReduce(dbuf, dbuf, (size_t)count, ncclDouble, ncclSum, h->comm, h->stream));
    return LZ_OK;
  }
  if (h->comm_kind == 2) {
    h->hbuf_a.resize((size_t)count);
    LZ_HIP(h, hipMemcpyAsync(h->hbuf_a.data(), dbuf, count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    LZ_HIP(h, hipStreamSynchronize(h->stream));
    h->host_syncs += 2;
    if (h->h_ar(h->h_user, h->hbuf_a.data(), count) != 0) return fail(h, LZ_ERR_COMM, "host all-reduce callback failed");
    LZ_HIP(h, hipMemcpyAsync(dbuf, h->hbuf_a.data(), count * sizeof(double), hipMemcpyHostToDevice, h->stream));
    LZ_HIP(h, hipStreamSynchronize(h->stream));
    return LZ_OK;
  }
  return fail(h, LZ_ERR_STATE, "world > 1 but no communicator initialised");
}

// make the SpMV input of basis row j complete on this rank; returns the x pointer to use
int comm_exchange_x(lz_handle h, int j, const double** x_out) {
  double* vj = h->d_V + (int64_t)j * h->ldv;
  *x_out = vj;
  if (h->world <= 1 && !(h->tune[6] && h->comm_kind)) return LZ_OK;
  if (h->xmode == 1) {
    if (h->peers.empty()) return LZ_OK;
    Scope sc(h, LZ_K_COMM, 8.0 * (h->total_send + h->total_recv), 0);
    h->n_exchange += 1;
    const bool direct = h->all_contig && h->comm_kind == 1;  // contiguous faces are sent straight out of V[j]
    if (!direct) {
      launch_gather(vj, h->d_send_idx, h->total_send, h->d_sendbuf, h->stream);
      LZ_TRY(check_launch(h, "gather"));
    }
    double* ghost = vj + h->rows_pad;
    if (h->comm_kind == 1) {
      LZ_NCCL(h, g_rccl.GroupStart());
      for (size_t p = 0; p < h->peers.size(); ++p) {
        const double* src = direct ? vj + h->sstart[p] : h->d_sendbuf + h->soff[p];
        if (h->scount[p] > 0)
          LZ_NCCL(h, g_rccl.Send(src, (size_t)h->scount[p], ncclDouble, h->peers[p], h->comm, h->stream));
        if (h->rcount[p] > 0)
          LZ_NCCL(h, g_rccl.Recv(ghost + h->roff[p], (size_t)h->rcount[p], ncclDouble, h->peers[p], h->comm, h->stream));
      }
      LZ_NCCL(h, g_rccl.GroupEnd());
    } else {
      h->hbuf_a.resize((size_t)std::max<int64_t>(h->total_send, 1));
      h->hbuf_b.resize((size_t)std::max<int64_t>(h->total_recv, 1));
      LZ_HIP(h, hipMemcpyAsync(h->hbuf_a.data(), h->d_sendbuf, h->total_send * sizeof(double), hipMemcpyDeviceToHost, h->stream));
      LZ_HIP(h, hipStreamSynchronize(h->stream));
      h->host_syncs += 2;
      if (h->h_ex(h->h_user, (int)h->peers.size(), h->peers.data(), h->hbuf_a.data(), h->scount.data(), h->hbuf_b.data(),
                  h->rcount.data()) != 0)
        return fail(h, LZ_ERR_COMM, "host halo-exchange callback failed");
      LZ_HIP(h, hipMemcpyAsync(ghost, h->hbuf_b.data(), h->total_recv * sizeof(double), hipMemcpyHostToDevice, h->stream));
      LZ_HIP(h, hipStreamSynchronize(h->stream));
    }
    return LZ_OK;
  }
  if (h->xmode == 2) {
    Scope sc(h, LZ_K_COMM, 8.0 * h->ag_chunk * h->world, 0);
    h->n_exchange += 1;
    if (h->comm_kind == 1) {
      LZ_NCCL(h, g_rccl.AllGather(vj, h->d_xfull, (size_t)h->ag_chunk, ncclDouble, h->comm, h->stream));
    } else {
      h->hbuf_a.resize((size_t)h->ag_chunk);
      h->hbuf_b.resize((size_t)(h->ag_chunk * h->world));
      LZ_HIP(h, hipMemcpyAsync(h->hbuf_a.data(), vj, h->ag_chunk * sizeof(double), hipMemcpyDeviceToHost, h->stream));
      LZ_HIP(h, hipStreamSynchronize(h->stream));
      h->host_syncs += 2;
      if (h->h_ag(h->h_user, h->hbuf_a.data(), h->hbuf_b.data(), h->ag_chunk) != 0)
        return fail(h, LZ_ERR_COMM, "host all-gather callback failed");
      LZ_HIP(h, hipMemcpyAsync(h->d_xfull, h->hbuf_b.data(), h->ag_chunk * h->world * sizeof(double), hipMemcpyHostToDevice, h->stream));
      LZ_HIP(h, hipStreamSynchronize(h->stream));
    }
    *x_out = h->d_xfull;
    return LZ_OK;
  }
  return fail(h, LZ_ERR_STATE, "world > 1 but neither lz_set_halo nor lz_set_allgather was called");
}

int require_basis(lz_handle h, int j) {
  if (!h->d_V) return fail(h, LZ_ERR_STATE, "no basis allocated (call lz_run or lz_basis_alloc first)");
  if (j < 0 || j >= h->n) return fail(h, LZ_ERR_ARG, "basis row index out of range");
  return LZ_OK;
}

}  // namespace api
}  // namespace lz

using namespace lz::api;

// ======================================================================= C ABI
extern "C" {

int lz_version(void) { return 100; }

int lz_device_count(int* count) {
  if (!count) return LZ_ERR_ARG;
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) {
    *count = 0;
    g_create_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e);
    return LZ_ERR_NODEVICE;
  }
  *count = c;
  return LZ_OK;
}

int lz_create(lz_handle* out, int device_id) {
  if (!out) return fail(nullptr, LZ_ERR_ARG, "out is NULL");
  *out = nullptr;
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess || c <= 0)
    return fail(nullptr, LZ_ERR_NODEVICE,
                std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
  if (device_id < 0 || device_id >= c) return fail(nullptr, LZ_ERR_ARG, "device_id out of range");
  e = hipSetDevice(device_id);
  if (e != hipSuccess) return fail(nullptr, LZ_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
  lz_context* h = new lz_context();
  h->dev = device_id;
  memset(&h->acc, 0, sizeof(h->acc));
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) h->name = std::string(prop.name) + " (" + prop.gcnArchName + ")";
  e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete h;
    return fail(nullptr, LZ_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
  }
  hipEventCreate(&h->run_a);
  hipEventCreate(&h->run_b);
  *out = h;
  return LZ_OK;
}

int lz_destroy(lz_handle h) {
  if (!h) return LZ_OK;
  hipSetDevice(h->dev);
  hipStreamSynchronize(h->stream);
  if (h->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
  big_free(h->csr.rowptr);
  big_free(h->csr.colidx);
  big_free(h->csr.vals);
  big_free(h->csr.rowblk);
  pb_free(h->csr.pb);
  pb_free(h->csrT.pb);
  ell_free(h->csr);
  ell_free(h->csrT);
  big_free(h->d_dense);
  big_free(h->d_V);
  big_free(h->d_r);
  big_free(h->d_r2);
  big_free(h->d_alpha);
  big_free(h->d_beta);
  big_free(h->d_c);
  big_free(h->d_nrm2);
  big_free(h->d_part);
  big_free(h->csrT.rowptr);
  big_free(h->csrT.colidx);
  big_free(h->csrT.vals);
  big_free(h->csrT.rowblk);
  big_free(h->d_B3);
  big_free(h->d_s);
  big_free(h->d_gamma);
  big_free(h->d_bi);
  big_free(h->d_xtmp);
  big_free(h->d_Y);
  big_free(h->d_S);
  big_free(h->d_rclk);
  big_free(h->d_gram);
  big_free(h->d_gclk);
  big_free(h->d_send_idx);
  big_free(h->d_sendbuf);
  big_free(h->d_xfull);
  big_free(h->d_om);
  big_free(h->d_omi);
  big_free(h->res_V);
  big_free(h->res_Y);
  if (h->h_pinned) hipHostFree(h->h_pinned);
  xfer_free(h->xfer);
  if (h->cstream) {
    hipStreamSynchronize(h->cstream);
    hipStreamDestroy(h->cstream);
    hipEventDestroy(h->e_bnd);
    hipEventDestroy(h->e_halo);
  }
  for (auto& e : h->events) {
    hipEventDestroy(e.a);
    hipEventDestroy(e.b);
  }
  for (auto& e : h->free_events) {
    hipEventDestroy(e.first);
    hipEventDestroy(e.second);
  }
  if (h->run_a) hipEventDestroy(h->run_a);
  if (h->run_b) hipEventDestroy(h->run_b);
  hipStreamDestroy(h->stream);
  delete h;
  return LZ_OK;
}

const char* lz_last_error(lz_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int lz_set_options(lz_handle h, int flags) {
  if (!h) return LZ_ERR_ARG;
#ifndef LZ_KBENCH
  if (flags & LZ_FLAG_QTW_MFMA)
    return fail(h, LZ_ERR_ARG, "lz_set_options: LZ_FLAG_QTW_MFMA (the 16x16x4 Q^T w arm, 20 % slower) was retired from the product library (build with KBENCH=1)");
#endif
  h->flags = flags;
  return LZ_OK;
}

int lz_set_tuning(lz_handle h, int index, int value) {
  if (!h) return LZ_ERR_ARG;
  if (index < 0 || index >= 24) return fail(h, LZ_ERR_ARG, "lz_set_tuning: knob index out of range");
  // The timing-only ablation arms (kernel variants that computed wrong results on purpose, for one-off measurements: knob 1 >= 20,
  // knob 3, knob 9 >= 10) were deleted in round 5; their results are in profiles/r01 .. r04.
  if ((index == 1 && value >= 20) || (index == 3 && value != 0) || (index == 9 && value >= 10))
    return fail(h, LZ_ERR_ARG, "lz_set_tuning: the timing-only ablation arms were removed in round 5 (their measurements are in profiles/)");
#ifndef LZ_KBENCH
  // Retired A/B arms (built, measured slower, kept bit-identity-tested in the kernel-bench build): the one-kernel /
  // one-launch-per-step engines (15 = 2, 3, 5), the persistent and LDS-staged Ritz GEMMs (9 >= 2), the ticket / deferred-fold
  // two-sided links (11 >= 2), the row-block-group interleaving of the two-phase SpMV (22 >= 2: round 5, 8-110 % slower)
  if ((index == 15 && value >= 2) || (index == 9 && value >= 2) || (index == 11 && value >= 2) || (index == 22 && value >= 2))
    return fail(h, LZ_ERR_ARG, "lz_set_tuning: this A/B arm was retired from the product library (build with KBENCH=1)");
#endif
  if (value < 0) return fail(h, LZ_ERR_ARG, "lz_set_tuning: negative value");
  h->tune[index] = value;
  return LZ_OK;
}

int lz_runtime_info(char* buf, size_t buflen) {
  if (!buf || buflen == 0) return LZ_ERR_ARG;
  Dl_info info;
  const char* hip = "";
  if (dladdr(reinterpret_cast<const void*>(static_cast<hipError_t (*)(void**, size_t)>(&hipMalloc)), &info) && info.dli_fname) hip = info.dli_fname;
  snprintf(buf, buflen, "hip=%s;rccl=%s", hip, g_rccl_path.c_str());
  return LZ_OK;
}

int lz_device_synchronize(lz_handle h) {
  if (!h) return LZ_ERR_ARG;
  LZ_HIP(h, hipSetDevice(h->dev));
  LZ_HIP(h, hipDeviceSynchronize());
  return LZ_OK;
}

int lz_device_memory(lz_handle h, int64_t* free_bytes, int64_t* total_bytes) {
  if (!h || !free_bytes || !total_bytes) return LZ_ERR_ARG;
  LZ_HIP(h, hipSetDevice(h->dev));
  size_t f = 0, t = 0;
  LZ_HIP(h, hipMemGetInfo(&f, &t));
  *free_bytes = (int64_t)f;
  *total_bytes = (int64_t)t;
  return LZ_OK;
}

int lz_device_name(lz_handle h, char* buf, size_t buflen) {
  if (!h || !buf || buflen == 0) return LZ_ERR_ARG;
  snprintf(buf, buflen, "%s", h->name.c_str());
  return LZ_OK;
}

int64_t lz_padded_rows(int64_t rows) { return round_up(rows, kPadDoubles); }

// ---- communication -------------------------------------------------------
int lz_comm_load(void) {
  const char* e = load_rccl();
  if (e) return fail(nullptr, LZ_ERR_COMM, e);
  return LZ_OK;
}

int lz_comm_unique_id(void* id, size_t id_bytes) {
  if (!id || id_bytes < sizeof(ncclUniqueId)) return fail(nullptr, LZ_ERR_ARG, "id buffer must hold at least 128 bytes");
  const char* e = load_rccl();
  if (e) return fail(nullptr, LZ_ERR_COMM, e);
  ncclUniqueId uid;
  ncclResult_t r = g_rccl.GetUniqueId(&uid);
  if (r != ncclSuccess) return fail(nullptr, LZ_ERR_COMM, std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r));
  memset(id, 0, id_bytes);
  memcpy(id, &uid, sizeof(uid));
  return LZ_OK;
}

int lz_comm_init_rccl(lz_handle h, int world, int rank, const void* id, size_t id_bytes) {
  if (!h) return LZ_ERR_ARG;
  if (world < 1 || rank < 0 || rank >= world || !id || id_bytes < sizeof(ncclUniqueId))
    return fail(h, LZ_ERR_ARG, "bad world/rank/id");
  const char* e = load_rccl();
  if (e) return fail(h, LZ_ERR_COMM, e);
  LZ_HIP(h, hipSetDevice(h->dev));
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  LZ_NCCL(h, g_rccl.CommInitRank(&h->comm, world, uid, rank));
  // A multi-rank RCCL run hands slices of the basis to ncclSend / ncclRecv / ncclAllGather as user buffers.  RCCL with N > 1 ranks has
  // never executed on this one-GPU pool, and neither has RCCL on reserve + map (VMM) ranges: such runs keep plain hipMalloc for their
  // large buffers (the occasional allocator stall costs setup time, never the timed region) until a multi-GPU node has shown otherwise.
  if (world > 1) big_vmm_disable();
  h->world = world;
  h->rank = rank;
  h->comm_kind = 1;
  return LZ_OK;
}

int lz_comm_init_host(lz_handle h, int world, int rank, lz_host_allreduce_fn ar, lz_host_exchange_fn ex,
                      lz_host_allgather_fn ag, void* user) {
  if (!h) return LZ_ERR_ARG;
  if (world < 1 || rank < 0 || rank >= world || !ar) return fail(h, LZ_ERR_ARG, "bad world/rank/callbacks");
  h->world = world;
  h->rank = rank;
  h->comm_kind = 2;
  h->h_ar = ar;
  h->h_ex = ex;
  h->h_ag = ag;
  h->h_user = user;
  return LZ_OK;
}

// ---- basis -------------------------------------------------------------------
// zero_rows: how many leading basis rows to clear.  The step API hands out an all-zero basis (the reference's
// np.zeros((n, M)), Lanczos.py:104); lz_run only needs row 0 cleared (padding + ghost tail around the uploaded v0):
// every other row is fully written before it is first read, and at j = 0 the beta * V[-1] term is skipped outright,
// so the 8nM-byte memset (3 ms of the 553 ms headline solve) is not paid per run.
}  // extern "C"
int lz::api::basis_alloc(lz_handle h, int n, int zero_rows) {
  if (!h) return LZ_ERR_ARG;
  if (h->kind == 0) return fail(h, LZ_ERR_STATE, "no matrix set (lz_set_csr / lz_set_dense)");
  if (n < 1) return fail(h, LZ_ERR_ARG, "n must be >= 1");
  if (h->world > 1 && h->xmode == 0) return fail(h, LZ_ERR_STATE, "multi-rank run needs lz_set_halo or lz_set_allgather");
  LZ_HIP(h, hipSetDevice(h->dev));
  LZ_HIP(h, hipStreamSynchronize(h->stream));
  const size_t vsz = (size_t)n * (size_t)h->ldv;
  const bool fresh = !h->d_V || h->n != n;
  if (fresh) {
    double* adopted = nullptr;
    {
      std::lock_guard<std::mutex> lk(h->res_mu);
      if (h->res_V && h->res_V_count >= vsz) {
        adopted = h->res_V;
        h->res_V = nullptr;
        h->res_V_count = 0;
      }
    }
    if (adopted) {
      LZ_TRY(dev_free(h, h->d_V));
      h->d_V = adopted;
    } else {
      LZ_TRY(dev_alloc(h, h->d_V, vsz));
    }
    if (getenv("LZ_DEBUG_PTR")) fprintf(stderr, "[lz] basis %p (%zu bytes, ld %lld)\n", (void*)h->d_V, vsz * sizeof(double), (long long)h->ldv);
    LZ_TRY(dev_alloc(h, h->d_r, (size_t)h->ldv));
    LZ_TRY(dev_alloc(h, h->d_r2, (size_t)h->ldv));
    LZ_TRY(dev_alloc(h, h->d_alpha, (size_t)n + 1));
    LZ_TRY(dev_alloc(h, h->d_beta, (size_t)n + 1));
    LZ_TRY(dev_alloc(h, h->d_c, 2 * ((size_t)2 * qtw_ldp(n + 2) + 8)));  // n + 1 coefficients; one-reduce mode: two runs + alpha (partial one-reduce loop: two such buffers alternate)
    LZ_TRY(dev_alloc(h, h->d_nrm2, 2));
  }
  if (fresh) {
    // Columns [rows_pad, ldv) of a basis row (ghost tail in halo mode, the chunk padding in all-gather mode, the stride
    // skew) are read by the SpMV as part of the extended vector but written by no kernel: they must not hold whatever
    // the recycled allocation held (0 * NaN = NaN in the dense GEMV over zero-padded columns).  Cleared once per allocation.
    if (h->tune[13] == 1) LZ_HIP(h, hipMemsetAsync(h->d_V, 0xFF, vsz * sizeof(double), h->stream));  // test knob: NaN-poison
    if (h->ldv > h->rows_pad)
      LZ_HIP(h, hipMemset2DAsync(h->d_V + h->rows_pad, (size_t)h->ldv * sizeof(double), 0, (size_t)(h->ldv - h->rows_pad) * sizeof(double),
                                 (size_t)n, h->stream));
  }
  h->n = n;
  h->qplan = plan_qtw(h->rows_pad, h->flags, h->tune, n);
  size_t need = (size_t)(n + 16) * (size_t)h->qplan.P;
  if (h->flags & LZ_FLAG_ONE_REDUCE) need = (size_t)2 * qtw_ldp(n + 2) * (size_t)h->qplan.P;
  if ((h->flags & LZ_FLAG_ONE_REDUCE) && (h->flags & LZ_FLAG_REORTH_PARTIAL)) need = onered_part_off(h) + std::max<size_t>(need, 8192);
  if (h->qplan.G <= 8) need = std::max<size_t>(need, fused_coff(h) + (size_t)(n + 16) * (size_t)h->qplan.G);  // fused small-problem path
  need = std::max<size_t>(need, 8192);  // (>= 3 x 2048: the three self-term partial runs of k_three_term_self)
  need = std::max<size_t>(need, (size_t)(h->rows / 4 + 64));                     // dense gemv / scalar spmv partials
  need = std::max<size_t>(need, (size_t)h->csr.n_rowblk + 64);
  if (h->csr.pb) need = std::max<size_t>(need, (size_t)pb_num_partials(h->csr.pb) + 64);
  if (h->csrT.pb) need = std::max<size_t>(need, (size_t)pb_num_partials(h->csrT.pb) + 64);
  LZ_TRY(ensure_part(h, need));
  LZ_HIP(h, hipMemsetAsync(h->d_V, 0, (size_t)std::min(zero_rows, n) * (size_t)h->ldv * sizeof(double), h->stream));
  LZ_HIP(h, hipMemsetAsync(h->d_r, 0, (size_t)h->ldv * sizeof(double), h->stream));
  LZ_HIP(h, hipMemsetAsync(h->d_r2, 0, (size_t)h->ldv * sizeof(double), h->stream));
  LZ_HIP(h, hipMemsetAsync(h->d_alpha, 0, ((size_t)n + 1) * sizeof(double), h->stream));
  LZ_HIP(h, hipMemsetAsync(h->d_beta, 0, ((size_t)n + 1) * sizeof(double), h->stream));
  LZ_HIP(h, hipMemsetAsync(h->d_c, 0, 2 * ((size_t)2 * qtw_ldp(n + 2) + 8) * sizeof(double), h->stream));
  LZ_HIP(h, hipMemsetAsync(h->d_nrm2, 0, 2 * sizeof(double), h->stream));
  LZ_HIP(h, hipStreamSynchronize(h->stream));
  return LZ_OK;
}

extern "C" {

int lz_basis_alloc(lz_handle h, int n) { return basis_alloc(h, n, n); }

int lz_basis_set_row(lz_handle h, int j, const double* row_local) {
  if (!h || !row_local) return LZ_ERR_ARG;
  LZ_TRY(require_basis(h, j));
  LZ_HIP(h, hipSetDevice(h->dev));
  LZ_HIP(h, hipMemcpyAsync(h->d_V + (int64_t)j * h->ldv, row_local, (size_t)h->rows * sizeof(double), hipMemcpyHostToDevice, h->stream));
  LZ_HIP(h, hipStreamSynchronize(h->stream));
  return LZ_OK;
}

int lz_basis_set_rows(lz_handle h, int j0, int count, const double* rows_local, int64_t ld) {
  if (!h || !rows_local) return LZ_ERR_ARG;
  if (count < 1 || ld < h->rows) return fail(h, LZ_ERR_ARG, "lz_basis_set_rows: need count >= 1 and ld >= rows_local");
  LZ_TRY(require_basis(h, j0));
  LZ_TRY(require_basis(h, j0 + count - 1));
  LZ_HIP(h, hipSetDevice(h->dev));
  // one strided copy for all rows (the static Lanczos.reorthogonalize(V, j) used to upload V row by row: n synchronous copies)
  LZ_TRY(upload2d(h, h->d_V + (int64_t)j0 * h->ldv, (size_t)h->ldv * sizeof(double), rows_local, (size_t)ld * sizeof(double), (size_t)h->rows * sizeof(double),
                  (size_t)count));
  LZ_HIP(h, hipStreamSynchronize(h->stream));
  return LZ_OK;
}

int lz_basis_get_row(lz_handle h, int j, double* row_local) {
  if (!h || !row_local) return LZ_ERR_ARG;
  LZ_TRY(require_basis(h, j));
  LZ_HIP(h, hipSetDevice(h->dev));
  LZ_HIP(h, hipMemcpyAsync(row_local, h->d_V + (int64_t)j * h->ldv, (size_t)h->rows * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  LZ_HIP(h, hipStreamSynchronize(h->stream));
  return LZ_OK;
}

int lz_r_set(lz_handle h, const double* r_local) {
  if (!h || !r_local) return LZ_ERR_ARG;
  LZ_TRY(require_basis(h, 0));
  LZ_HIP(h, hipSetDevice(h->dev));
  LZ_HIP(h, hipMemcpyAsync(h->d_r, r_local, (size_t)h->rows * sizeof(double), hipMemcpyHostToDevice, h->stream));
  LZ_HIP(h, hipStreamSynchronize(h->stream));
  return LZ_OK;
}

int lz_r_get(lz_handle h, double* r_local) {
  if (!h || !r_local) return LZ_ERR_ARG;
  LZ_TRY(require_basis(h, 0));
  LZ_HIP(h, hipSetDevice(h->dev));
  LZ_HIP(h, hipMemcpyAsync(r_local, h->d_r, (size_t)h->rows * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  LZ_HIP(h, hipStreamSynchronize(h->stream));
  return LZ_OK;
}

// ---- single steps ---------------------------------------------------------------
int lz_step_spmv(lz_handle h, int j, double* dot_out) {
  if (!h) return LZ_ERR_ARG;
  LZ_TRY(require_basis(h, j));
  LZ_HIP(h, hipSetDevice(h->dev));
  LZ_TRY(step_spmv(h, j));
  if (dot_out) LZ_HIP(h, hipMemcpyAsync(dot_out, h->d_alpha + j, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  LZ_HIP(h, hipStreamSynchronize(h->stream));
  return LZ_OK;
}

int lz_step_reorth(lz_handle h, int j, int nrows, int scale, double* beta_out, double* c_out) {
  if (!h) return LZ_ERR_ARG;
  LZ_TRY(require_basis(h, j));
  if (nrows < 1 || nrows > h->n || j >= nrows) return fail(h, LZ_ERR_ARG, "lz_step_reorth: need 1 <= nrows <= n and j < nrows");
  LZ_HIP(h, hipSetDevice(h->dev));
  if (scale) {
    // ||r||^2 of the current r: r = r - 0 * V[j] leaves r unchanged bit for bit and refreshes d_nrm2
    LZ_HIP(h, hipMemsetAsync(h->d_c + h->n, 0, sizeof(double), h->stream));
    LZ_TRY(step_three_term(h, j, -1, h->d_c + h->n, nullptr));
  }
  LZ_TRY(step_reorth(h, j, nrows, scale != 0, h->n));  // beta lands in d_beta[n] (scratch slot)
  if (beta_out && scale) LZ_HIP(h, hipMemcpyAsync(beta_out, h->d_beta + h->n, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (c_out) LZ_HIP(h, hipMemcpyAsync(c_out, h->d_c, (size_t)nrows * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  LZ_HIP(h, hipStreamSynchronize(h->stream));
  return LZ_OK;
}

int lz_step_three_term(lz_handle h, int j, int jm1, double alpha, double beta, double* norm2_out) {
  if (!h) return LZ_ERR_ARG;
  LZ_TRY(require_basis(h, j));
  if (jm1 >= h->n) return fail(h, LZ_ERR_ARG, "lz_step_three_term: jm1 out of range");
  LZ_HIP(h, hipSetDevice(h->dev));
  double ab[2] = {alpha, beta};
  // scratch scalars: d_c[n] is never used by the recurrence, d_beta[n] likewise
  LZ_HIP(h, hipMemcpyAsync(h->d_c + h->n, &ab[0], sizeof(double), hipMemcpyHostToDevice, h->stream));
  LZ_HIP(h, hipMemcpyAsync(h->d_beta + h->n, &ab[1], sizeof(double), hipMemcpyHostToDevice, h->stream));
  LZ_TRY(step_three_term(h, j, jm1, h->d_c + h->n, h->d_beta + h->n));
  if (norm2_out) LZ_HIP(h, hipMemcpyAsync(norm2_out, h->d_nrm2, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  LZ_HIP(h, hipStreamSynchronize(h->stream));
  return LZ_OK;
}

int lz_last_engine(lz_handle h, int* engine) {
  if (!h || !engine) return LZ_ERR_ARG;
  *engine = h->last_engine;
  return LZ_OK;
}

int lz_last_host_syncs(lz_handle h, int64_t* syncs) {
  if (!h || !syncs) return LZ_ERR_ARG;
  *syncs = h->host_syncs;
  return LZ_OK;
}

int lz_last_sweeps(lz_handle h, int* sweeps) {
  if (!h || !sweeps) return LZ_ERR_ARG;
  *sweeps = h->last_sweeps;
  return LZ_OK;
}

int lz_last_sweep_log(lz_handle h, int* log, int n) {
  if (!h || !log || n < 0) return LZ_ERR_ARG;
  if ((size_t)n > h->sweep_log.size()) return fail(h, LZ_ERR_ARG, "lz_last_sweep_log: the last lz_run had fewer steps (or none was run, or it was the host-decided loop)");
  for (int j = 0; j < n; ++j) log[j] = h->sweep_log[(size_t)j];
  return LZ_OK;
}

int lz_last_sweep_misses(lz_handle h, int* misses) {
  if (!h || !misses) return LZ_ERR_ARG;
  *misses = h->last_misses;
  return LZ_OK;
}

int lz_get_timings(lz_handle h, lz_timings* out) {
  if (!h || !out) return LZ_ERR_ARG;
  LZ_HIP(h, hipSetDevice(h->dev));
  LZ_TRY(drain_events(h));
  *out = h->acc;
  memset(&h->acc, 0, sizeof(h->acc));
  h->n_allreduce_last = h->n_allreduce;
  h->n_exchange_last = h->n_exchange;
  h->n_allreduce = h->n_exchange = 0;
  return LZ_OK;
}

int lz_comm_counts(lz_handle h, int64_t* allreduces, int64_t* exchanges) {
  if (!h || !allreduces || !exchanges) return LZ_ERR_ARG;
  *allreduces = h->n_allreduce_last;
  *exchanges = h->n_exchange_last;
  return LZ_OK;
}

}  // extern "C"
