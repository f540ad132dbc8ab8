// hipMalloc against reserve + hipMemCreate + hipMemMap for a buffer of the headline basis' size, each followed by a first and a second
// touch of every page and a release, in a mixed sequence: which of the two ever makes the caller wait, and when.  (VERDICT r4 item 5:
// the first execute_Lanczos of an object waited 0.1 - 0.9 s "for the allocator".)
// build: hipcc --offload-arch=gfx950 -O2 -o tools/probes/alloc_pattern_probe tools/probes/alloc_pattern_probe.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
static int fill(void* p, size_t bytes, double v, double* s) {
  double t = now();
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (double*)p, bytes / 8, v);
  CK(hipDeviceSynchronize());
  *s = now() - t;
  return 0;
}
int main(int argc, char** argv) {
  const size_t total = (argc > 1 ? (size_t)atoll(argv[1]) : 16) << 30;
  const size_t chunk = (argc > 2 ? (size_t)atoll(argv[2]) : 2048) << 20;
  const char* seq = argc > 3 ? argv[3] : "MMMVVMVMMVVM";
  CK(hipSetDevice(0));
  CK(hipFree(0));
  hipMemAllocationProp prop;
  memset(&prop, 0, sizeof prop);
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc acc;
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = 0;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  for (int i = 0; seq[i]; ++i) {
    double f1 = 0, f2 = 0;
    if (seq[i] == 'M') {
      void* p = nullptr;
      double t0 = now();
      CK(hipMalloc(&p, total));
      double t1 = now();
      if (fill(p, total, 1.0, &f1) || fill(p, total, 2.0, &f2)) return 1;
      double t2 = now();
      CK(hipFree(p));
      printf("{\"op\": \"hipMalloc\", \"i\": %d, \"alloc_s\": %.4f, \"first_touch_s\": %.4f, \"second_touch_s\": %.4f, \"free_s\": %.4f}\n", i, t1 - t0, f1, f2, now() - t2);
    } else {
      void* va = nullptr;
      double t0 = now();
      CK(hipMemAddressReserve(&va, total, 1 << 21, nullptr, 0));
      std::vector<hipMemGenericAllocationHandle_t> hs;
      for (size_t off = 0; off < total; off += chunk) {
        const size_t sz = total - off < chunk ? total - off : chunk;
        hipMemGenericAllocationHandle_t hh;
        CK(hipMemCreate(&hh, sz, &prop, 0));
        CK(hipMemMap((char*)va + off, sz, 0, hh, 0));
        CK(hipMemSetAccess((char*)va + off, sz, &acc, 1));
        hs.push_back(hh);
      }
      double t1 = now();
      if (fill(va, total, 1.0, &f1) || fill(va, total, 2.0, &f2)) return 1;
      double t2 = now();
      for (size_t k = 0; k < hs.size(); ++k) {
        const size_t off = k * chunk, sz = total - off < chunk ? total - off : chunk;
        CK(hipMemUnmap((char*)va + off, sz));
        CK(hipMemRelease(hs[k]));
      }
      CK(hipMemAddressFree(va, total));
      printf("{\"op\": \"reserve+create+map\", \"i\": %d, \"chunk_MB\": %zu, \"alloc_s\": %.4f, \"first_touch_s\": %.4f, \"second_touch_s\": %.4f, \"free_s\": %.4f}\n", i, chunk >> 20,
             t1 - t0, f1, f2, now() - t2);
    }
    fflush(stdout);
  }
  return 0;
}
