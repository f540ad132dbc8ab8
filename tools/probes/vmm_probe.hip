// Can the Krylov basis be mapped progressively behind the solve?    Measures on the GPU box:
//   1. hipMalloc / hipFree of the headline's 16 GB basis (what the first execute_Lanczos of a process waits for),
//   2. the same 16 GB as one virtual range (hipMemAddressReserve) backed chunk by chunk (hipMemCreate + hipMemMap + hipMemSetAccess),
//   3. whether another host thread can launch kernels and upload 640 MB of pageable memory WHILE one thread allocates / maps,
//   4. that a kernel can stream across chunk boundaries of the mapped range at the usual rate.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/probes/vmm_probe tools/probes/vmm_probe.hip -lpthread
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                              \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      printf("{\"error\": \"%s -> %s\"}\n", #x, hipGetErrorString(e_));                      \
      return 1;                                                                            \
    }                                                                                      \
  } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void k_fill(double* p, size_t n, double v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void k_sum(const double* p, size_t n, double* out) {
  double a = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += p[i];
  if (a == 1.2345e300) out[0] = a;
}

// what a second thread experiences: a tiny kernel launch + sync, and an upload of `bytes` pageable bytes
static void other_thread(std::atomic<int>* stop, size_t bytes, double* worst_launch_ms, double* upload_s, int* uploads) {
  hipSetDevice(0);
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  double* d = nullptr;
  hipMalloc(&d, bytes);
  std::vector<char> host(bytes, 1);
  *worst_launch_ms = 0;
  *upload_s = 0;
  *uploads = 0;
  while (!stop->load()) {
    double t = now();
    hipLaunchKernelGGL(k_fill, dim3(1), dim3(64), 0, s, d, (size_t)64, 1.0);
    hipStreamSynchronize(s);
    double ms = 1e3 * (now() - t);
    if (ms > *worst_launch_ms) *worst_launch_ms = ms;
    t = now();
    hipMemcpyAsync(d, host.data(), bytes, hipMemcpyHostToDevice, s);
    hipStreamSynchronize(s);
    double u = now() - t;
    if (u > *upload_s) *upload_s = u;
    ++*uploads;
  }
  hipFree(d);
  hipStreamDestroy(s);
}

int main(int argc, char** argv) {
  const size_t GB = (size_t)1 << 30;
  const size_t total = (argc > 1 ? (size_t)atoll(argv[1]) : 16) * GB;
  const size_t chunk = (argc > 2 ? (size_t)atoll(argv[2]) : 2048) << 20;
  CK(hipSetDevice(0));
  CK(hipFree(0));
  // ---- 0. what a process that has USED its memory sees: allocate, touch every page, free, allocate again (three rounds)
  for (int round = 0; round < 3; ++round) {
    void* p = nullptr;
    double t0 = now();
    CK(hipMalloc(&p, total));
    double t1 = now();
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (double*)p, total / 8, 1.0);
    CK(hipDeviceSynchronize());
    double t2 = now();
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (double*)p, total / 8, 2.0);
    CK(hipDeviceSynchronize());
    double t3 = now();
    CK(hipFree(p));
    double t4 = now();
    printf("{\"what\": \"malloc + touch + free\", \"round\": %d, \"malloc_s\": %.4f, \"first_fill_s\": %.4f, \"second_fill_s\": %.4f, \"free_s\": %.4f}\n", round,
           t1 - t0, t2 - t1, t3 - t2, t4 - t3);
    fflush(stdout);
  }
  // ---- 1. plain hipMalloc, alone and with the other thread running
  for (int with_other = 0; with_other < 2; ++with_other) {
    std::atomic<int> stop{0};
    double wl = 0, up = 0;
    int nu = 0;
    std::thread th;
    if (with_other) {
      th = std::thread(other_thread, &stop, (size_t)640 << 20, &wl, &up, &nu);
      std::this_thread::sleep_for(std::chrono::milliseconds(300));
    }
    void* p = nullptr;
    double t0 = now();
    CK(hipMalloc(&p, total));
    double t1 = now();
    if (with_other) {
      stop = 1;
      th.join();
    }
    double t2 = now();
    CK(hipFree(p));
    double t3 = now();
    printf("{\"what\": \"hipMalloc\", \"GB\": %zu, \"other_thread\": %d, \"malloc_s\": %.4f, \"free_s\": %.4f, \"other_worst_launch_ms\": %.3f, \"other_worst_upload_640MB_s\": %.4f, \"other_uploads\": %d}\n",
           total / GB, with_other, t1 - t0, t3 - t2, wl, up, nu);
    fflush(stdout);
  }
  // ---- 2. reserve + map chunk by chunk
  hipMemAllocationProp prop;
  memset(&prop, 0, sizeof prop);
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  printf("{\"what\": \"granularity\", \"bytes\": %zu}\n", gran);
  for (int with_other = 0; with_other < 2; ++with_other) {
    std::atomic<int> stop{0};
    double wl = 0, up = 0;
    int nu = 0;
    std::thread th;
    if (with_other) {
      th = std::thread(other_thread, &stop, (size_t)640 << 20, &wl, &up, &nu);
      std::this_thread::sleep_for(std::chrono::milliseconds(300));
    }
    void* va = nullptr;
    double t0 = now();
    CK(hipMemAddressReserve(&va, total, gran, nullptr, 0));
    double t_res = now() - t0;
    std::vector<hipMemGenericAllocationHandle_t> hs;
    std::vector<double> per;
    hipMemAccessDesc acc;
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = 0;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    double t_all = now();
    for (size_t off = 0; off < total; off += chunk) {
      const size_t sz = std::min(chunk, total - off);
      double t = now();
      hipMemGenericAllocationHandle_t hh;
      CK(hipMemCreate(&hh, sz, &prop, 0));
      CK(hipMemMap((char*)va + off, sz, 0, hh, 0));
      CK(hipMemSetAccess((char*)va + off, sz, &acc, 1));
      per.push_back(now() - t);
      hs.push_back(hh);
    }
    t_all = now() - t_all;
    if (with_other) {
      stop = 1;
      th.join();
    }
    double mx = 0;
    for (double x : per) mx = std::max(mx, x);
    printf("{\"what\": \"reserve+map\", \"GB\": %zu, \"chunk_MB\": %zu, \"other_thread\": %d, \"reserve_s\": %.5f, \"map_all_s\": %.4f, \"chunks\": %zu, \"first_chunk_s\": %.4f, \"worst_chunk_s\": %.4f, \"other_worst_launch_ms\": %.3f, \"other_worst_upload_640MB_s\": %.4f, \"other_uploads\": %d}\n",
           total / GB, chunk >> 20, with_other, t_res, t_all, per.size(), per[0], mx, wl, up, nu);
    fflush(stdout);
    if (!with_other) {
      // ---- 4. stream across the mapped range
      double* out = nullptr;
      CK(hipMalloc(&out, 64));
      hipEvent_t a, b;
      hipEventCreate(&a);
      hipEventCreate(&b);
      const size_t n = total / 8;
      hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (double*)va, n, 0.5);
      CK(hipDeviceSynchronize());
      hipEventRecord(a, 0);
      hipLaunchKernelGGL(k_sum, dim3(8192), dim3(256), 0, 0, (const double*)va, n, out);
      hipEventRecord(b, 0);
      CK(hipDeviceSynchronize());
      float ms = 0;
      hipEventElapsedTime(&ms, a, b);
      printf("{\"what\": \"read mapped range\", \"GB\": %zu, \"ms\": %.3f, \"TBps\": %.3f}\n", total / GB, ms, total / (ms * 1e-3) / 1e12);
      hipFree(out);
    }
    double tu = now();
    for (size_t i = 0; i < hs.size(); ++i) {
      CK(hipMemUnmap((char*)va + i * chunk, std::min(chunk, total - i * chunk)));
      CK(hipMemRelease(hs[i]));
    }
    CK(hipMemAddressFree(va, total));
    printf("{\"what\": \"unmap+release\", \"s\": %.4f}\n", now() - tu);
    fflush(stdout);
  }
  // ---- 5. a second hipMalloc of the same size right after a free (does the runtime keep the pages?)
  {
    void* p = nullptr;
    double t0 = now();
    CK(hipMalloc(&p, total));
    double t1 = now();
    CK(hipFree(p));
    printf("{\"what\": \"hipMalloc again\", \"malloc_s\": %.4f}\n", t1 - t0);
  }
  return 0;
}
