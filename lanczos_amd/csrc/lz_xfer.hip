// Result publication: large device -> host copies (the basis V and the Ritz vectors Y, Lanczos.py:132-141,153-156: the two
// M x n arrays the reference hands to its caller).  The destination is the caller's NumPy memory - pageable, usually not even
// faulted in yet - and a plain hipMemcpy into it runs at 17-25 GB/s on this host (tools/publish_probe.py): the runtime pins,
// copies and unpins piece by piece, one after the other.  Here the three stages overlap: the copy engine fills a ring of pinned
// staging buffers at the link's rate while a few host threads move finished pieces into the destination (and take its page
// faults).  Pure data movement: no arithmetic, identical bytes.
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <system_error>
#include <thread>
#include <vector>

#include "lz_internal.h"

namespace lz {

struct Xfer {
  static constexpr int kBufs = 6;
  static constexpr size_t kPiece = (size_t)32 << 20;
  void* buf[kBufs] = {};
  hipEvent_t ev[kBufs] = {};
  bool ready = false;
};

void xfer_free(Xfer*& x) {
  if (!x) return;
  for (int b = 0; b < Xfer::kBufs; ++b) {
    if (x->buf[b]) hipHostFree(x->buf[b]);
    if (x->ev[b]) hipEventDestroy(x->ev[b]);
  }
  delete x;
  x = nullptr;
}

static hipError_t xfer_init(Xfer*& x) {
  if (!x) x = new Xfer();
  if (x->ready) return hipSuccess;
  for (int b = 0; b < Xfer::kBufs; ++b) {
    hipError_t e = hipHostMalloc(&x->buf[b], Xfer::kPiece, hipHostMallocDefault);
    if (e != hipSuccess) return e;
    e = hipEventCreateWithFlags(&x->ev[b], hipEventDisableTiming);
    if (e != hipSuccess) return e;
  }
  x->ready = true;
  return hipSuccess;
}

int xfer_threads() {
  static const int n = [] {
    const char* s = std::getenv("LZ_XFER_THREADS");  // 0: plain hipMemcpy (A/B)
    if (s && *s) return std::max(0, std::min(32, std::atoi(s)));
    const unsigned hc = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(6u, hc ? hc / 2 : 1u));
  }();
  return n;
}

// `height` rows of `width` bytes, row strides dpitch (host) / spitch (device), on `stream` (ordered behind what is queued there);
// returns when the destination is complete.  Small copies and LZ_XFER_THREADS=0 take the runtime's own path.
hipError_t xfer_d2h(int dev, hipStream_t stream, Xfer*& state, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width,
                    size_t height) {
  const int T = xfer_threads();
  if (width == 0 || height == 0) return hipStreamSynchronize(stream);
  auto plain = [&]() -> hipError_t {
    const hipError_t pe = hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyDeviceToHost, stream);
    return pe != hipSuccess ? pe : hipStreamSynchronize(stream);
  };
  if (T == 0 || width * height < ((size_t)192 << 20)) return plain();
  hipError_t e = xfer_init(state);
  if (e != hipSuccess) {  // no pinned memory to be had (locked-memory limit): the runtime's own path still works
    xfer_free(state);
    (void)hipGetLastError();
    return plain();
  }
  struct Job {
    int b;
    char* d;
    size_t dp, w, h;  // h rows of w bytes, packed in the staging buffer
  };
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Job> queue;
  bool busy[Xfer::kBufs] = {};
  bool closing = false;
  hipError_t werr = hipSuccess;
  auto worker = [&]() {
    (void)hipSetDevice(dev);
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !queue.empty() || closing; });
        if (queue.empty()) return;
        j = queue.front();
        queue.pop_front();
      }
      const hipError_t we = hipEventSynchronize(state->ev[j.b]);
      if (we == hipSuccess) {
        const char* s = static_cast<const char*>(state->buf[j.b]);
        if (j.dp == j.w || j.h == 1)
          std::memcpy(j.d, s, j.w * j.h);
        else
          for (size_t r = 0; r < j.h; ++r) std::memcpy(j.d + r * j.dp, s + r * j.w, j.w);
      }
      {
        std::lock_guard<std::mutex> lk(mu);
        if (we != hipSuccess && werr == hipSuccess) werr = we;
        busy[j.b] = false;
      }
      cv.notify_all();
    }
  };
  std::vector<std::thread> pool;
  try {
    for (int t = 0; t < T; ++t) pool.emplace_back(worker);
  } catch (const std::system_error&) {  // thread limit reached: carry on with the threads that did start, or plainly
    if (pool.empty()) return plain();
  }
  int next = 0;
  auto submit = [&](char* d, size_t dp, const char* s, size_t sp, size_t w, size_t h) -> hipError_t {
    const int b = next;
    next = (next + 1) % Xfer::kBufs;
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return !busy[b]; });
      if (werr != hipSuccess) return werr;
      busy[b] = true;
    }
    hipError_t se = h == 1 ? hipMemcpyAsync(state->buf[b], s, w, hipMemcpyDeviceToHost, stream)
                           : hipMemcpy2DAsync(state->buf[b], w, s, sp, w, h, hipMemcpyDeviceToHost, stream);
    if (se == hipSuccess) se = hipEventRecord(state->ev[b], stream);
    if (se != hipSuccess) {
      std::lock_guard<std::mutex> lk(mu);
      busy[b] = false;
      return se;
    }
    {
      std::lock_guard<std::mutex> lk(mu);
      queue.push_back({b, d, dp, w, h});
    }
    cv.notify_all();
    return hipSuccess;
  };
  char* d = static_cast<char*>(dst);
  const char* s = static_cast<const char*>(src);
  if (width >= Xfer::kPiece) {
    for (size_t r = 0; r < height && e == hipSuccess; ++r)
      for (size_t o = 0; o < width && e == hipSuccess; o += Xfer::kPiece)
        e = submit(d + r * dpitch + o, 0, s + r * spitch + o, 0, std::min(Xfer::kPiece, width - o), 1);
  } else {
    const size_t per = Xfer::kPiece / width;
    for (size_t r = 0; r < height && e == hipSuccess; r += per) e = submit(d + r * dpitch, dpitch, s + r * spitch, spitch, width, std::min(per, height - r));
  }
  {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] {
      if (!queue.empty()) return false;
      for (bool b : busy)
        if (b) return false;
      return true;
    });
    closing = true;
  }
  cv.notify_all();
  for (auto& t : pool) t.join();
  if (e == hipSuccess) e = werr;
  const hipError_t se = hipStreamSynchronize(stream);
  return e != hipSuccess ? e : se;
}

// The other direction of the boundary (host -> device: the CSR arrays of lz_set_csr, a dense matrix, the start vector) does
// NOT go through this ring.  Round 4 built the twin pipeline (host threads fill the pinned ring, the copy engine drains it) and
// measured it against the runtime's own path on the headline's 640 MB of CSR, a 4.6 GB dense matrix and the 80 MB start
// vector (tools/upload_probe.py, profiles/r04/ab_upload_{staged,plain}_h2d.json): 54 vs 57 GB/s for the dense matrix, 75 vs
// 51 ms for lz_set_csr, 2.8 vs 2.2 ms for the start vector - a resident pageable SOURCE is pinned in place and read at the
// link's rate by the runtime; it is the page faults of a fresh DESTINATION that made the staged path pay for results.  Removed.

}  // namespace lz
