// What read bandwidth does a pure streaming kernel reach on this MI355X?  (ceiling for the re-orthogonalisation passes)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_read(const d2* __restrict__ p, size_t n2, double* out) {
  double acc = 0;
  size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (; i + (U - 1) * 256 < n2; i += stride) {
    d2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * 256) : p[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y;
  }
  if (acc == 1.2345e300) out[0] = acc;
}
template <int U>
__global__ __launch_bounds__(256) void k_copy(const d2* __restrict__ p, d2* __restrict__ q, size_t n2) {
  size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (; i + (U - 1) * 256 < n2; i += stride) {
    d2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + i + u * 256);
#pragma unroll
    for (int u = 0; u < U; ++u) q[i + u * 256] = v[u];
  }
}
// same bytes, but walked like the re-orthogonalisation update: lane = one column position, loop over basis rows
template <int U>
__global__ __launch_bounds__(256) void k_read_rows(const d2* __restrict__ p, size_t ld2, int nrows, double* out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= ld2) return;
  double acc = 0;
  for (int k = 0; k + U <= nrows; k += U) {
    d2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + (size_t)(k + u) * ld2 + i);
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y;
  }
  if (acc == 1.2345e300) out[0] = acc;
}
// update-like WITH the arithmetic: per row a scalar coefficient, unfused multiply then add, final read-modify-write
template <int U, int MODE>  // MODE bit0: scalar-load coefficients, bit1: unfused mul+add (else fma), bit2: final RMW store
__global__ __launch_bounds__(256) void k_update_like(d2* __restrict__ p, size_t ld2, int nrows, const double* __restrict__ c, double* out, d2* __restrict__ other = nullptr) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= ld2) return;
  double tx = 0, ty = 0;
  for (int k = 0; k + U <= nrows; k += U) {
    d2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + (size_t)(k + u) * ld2 + i);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const double ck = (MODE & 1) ? c[k + u] : 0.5 + u;
      if (MODE & 2) {
        tx = __dadd_rn(tx, __dmul_rn(ck, v[u].x));
        ty = __dadd_rn(ty, __dmul_rn(ck, v[u].y));
      } else {
        tx = fma(ck, v[u].x, tx);
        ty = fma(ck, v[u].y, ty);
      }
    }
  }
  if (MODE & 32) {  // nt store into the streamed buffer, row nrows (NOT read by the loop)
    __builtin_nontemporal_store(d2{tx, ty}, p + (size_t)nrows * ld2 + i);
  } else if (MODE & 64) {  // RMW of the last-read row with a non-temporal store
    d2* o = p + (size_t)(nrows - 1) * ld2 + i;
    d2 w = __builtin_nontemporal_load(o);
    w.x = 2.0 * w.x - tx;
    w.y = 2.0 * w.y - ty;
    __builtin_nontemporal_store(w, o);
  } else if (MODE & 128) {  // plain store into the streamed buffer, row nrows (NOT read by the loop)
    p[(size_t)nrows * ld2 + i] = d2{tx, ty};
  } else if (MODE & 8) {  // store only, to another buffer
    other[i] = d2{tx, ty};
  } else if (MODE & 16) {  // extra read only (no store)
    d2 w = p[(size_t)(nrows - 1) * ld2 + i];
    if (w.x + tx == 1.2345e300) out[0] = ty;
  } else if (MODE & 4) {
    d2* o = p + (size_t)(nrows - 1) * ld2 + i;
    d2 w = *o;
    w.x = 2.0 * w.x - tx;
    w.y = 2.0 * w.y - ty;
    *o = w;
  } else if (tx + ty == 1.2345e300) out[0] = tx;
}
// ... and like Q^T w: a block owns a 40 KB slice of every row and walks the rows 8 at a time
template <int R, int U, int PART = 0>  // PART 1: per-wave 8-byte partial store after every tile ([row][wave] layout), 2: one burst at the end
__global__ __launch_bounds__(256) void k_read_slices(const d2* __restrict__ p, size_t ld2, int nrows, int cnt2, double* out, double* part = nullptr) {
  const size_t base = (size_t)blockIdx.x * cnt2;
  double acc = 0;
  const int pid = blockIdx.x * 4 + (threadIdx.x >> 6), P = gridDim.x * 4;
  __shared__ double keep[4][512];
  for (int i0 = 0; i0 + R <= nrows; i0 += R) {
    if (PART) acc = 0;
    for (int t = threadIdx.x; t < cnt2; t += 256 * U) {
      d2 v[R][U];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int q = 0; q < R; ++q) v[q][u] = (t + 256 * u < cnt2) ? __builtin_nontemporal_load(p + (size_t)(i0 + q) * ld2 + base + t + 256 * u) : d2{0, 0};
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int q = 0; q < R; ++q) acc += v[q][u].x + v[q][u].y;
    }
    if (PART == 1 && (threadIdx.x & 63) < R) part[(size_t)(i0 + (threadIdx.x & 63)) * P + pid] = acc;
    if (PART == 2 && (threadIdx.x & 63) < R) keep[threadIdx.x >> 6][i0 + (threadIdx.x & 63)] = acc;
  }
  if (PART == 2) {
    for (int i = threadIdx.x & 63; i < nrows; i += 64) __builtin_nontemporal_store(keep[threadIdx.x >> 6][i], part + (size_t)pid * 512 + i);
  }
  if (!PART && acc == 1.2345e300) out[0] = acc;
}
__global__ void k_fill_random(double* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long h = i * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    p[i] = (double)(h >> 11) * (2.0 / 9007199254740992.0) - 1.0;
  }
}

int main() {
  const size_t bytes = (size_t)8 << 30, n2 = bytes / 16;
  d2 *a, *b; double* out;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&out, 8);
  hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](auto launch, double gb, const char* name) {
    launch(); hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 5; ++r) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    printf("%-28s %8.3f ms  %7.1f GB/s\n", name, best, gb / best * 1e3 / 1e9 * 1e0);
  };
  for (int g : {1024, 2048, 4096, 8192, 19532}) {
    char nm[64];
    snprintf(nm, 64, "read U8 nt grid=%d", g);  time([&] { hipLaunchKernelGGL((k_read<8, true>), dim3(g), dim3(256), 0, 0, a, n2, out); }, (double)bytes, nm);
    snprintf(nm, 64, "read U8 plain grid=%d", g); time([&] { hipLaunchKernelGGL((k_read<8, false>), dim3(g), dim3(256), 0, 0, a, n2, out); }, (double)bytes, nm);
  }
  time([&] { hipLaunchKernelGGL((k_read<16, true>), dim3(4096), dim3(256), 0, 0, a, n2, out); }, (double)bytes, "read U16 nt grid=4096");
  time([&] { hipLaunchKernelGGL((k_read<4, true>), dim3(8192), dim3(256), 0, 0, a, n2, out); }, (double)bytes, "read U4 nt grid=8192");
  time([&] { hipLaunchKernelGGL((k_copy<8>), dim3(4096), dim3(256), 0, 0, a, b, n2); }, 2.0 * bytes, "copy U8 nt grid=4096 (r+w)");
  {
    const size_t ld2 = 5000000; const int nrows = 100;  // 100 rows x 80 MB = 8 GB
    time([&] { hipLaunchKernelGGL((k_read_rows<8>), dim3((unsigned)((ld2 + 255) / 256)), dim3(256), 0, 0, a, ld2, nrows, out); }, 16.0 * ld2 * nrows, "rows-walk U8 (update-like)");
    time([&] { hipLaunchKernelGGL((k_read_rows<16>), dim3((unsigned)((ld2 + 255) / 256)), dim3(256), 0, 0, a, ld2, nrows, out); }, 16.0 * ld2 * (nrows / 16 * 16), "rows-walk U16");
    time([&] { hipLaunchKernelGGL((k_read_rows<4>), dim3((unsigned)((ld2 + 255) / 256)), dim3(256), 0, 0, a, ld2, nrows, out); }, 16.0 * ld2 * nrows, "rows-walk U4");
    double* c; hipMalloc(&c, 1024 * 8); hipMemset(c, 0, 1024 * 8);
#define UL(mode, name) time([&] { hipLaunchKernelGGL((k_update_like<8, mode>), dim3((unsigned)((ld2 + 255) / 256)), dim3(256), 0, 0, a, ld2, 96, c, out); }, 16.0 * ld2 * 96, name);
    UL(0, "upd-like: fma, const coef")
    UL(1, "upd-like: fma, scalar-load coef")
    UL(2, "upd-like: mul+add, const coef")
    UL(3, "upd-like: mul+add, scalar coef")
    UL(7, "upd-like: mul+add, scalar coef, RMW")
    UL(4, "upd-like: fma, const coef, RMW")
    time([&] { hipLaunchKernelGGL((k_update_like<8, 8 | 2>), dim3((unsigned)((ld2 + 255) / 256)), dim3(256), 0, 0, a, ld2, 96, c, out, b); }, 16.0 * ld2 * 96, "upd-like: store to OTHER buffer");
    UL(16 | 2, "upd-like: extra read, no store")
    UL(32 | 2, "upd-like: nt store, same buffer, unread row")
    UL(128 | 2, "upd-like: plain store, same buffer, unread row")
    UL(64 | 2, "upd-like: RMW last row, nt load + nt store")
    const int cnt2 = 2560;  // 40 KB slices
    const unsigned G = (unsigned)((ld2 + cnt2 - 1) / cnt2);
    time([&] { hipLaunchKernelGGL((k_read_slices<8, 2>), dim3(G), dim3(256), 0, 0, a, ld2, nrows, cnt2, out); }, 16.0 * ld2 * (nrows / 8 * 8), "slices R8 U2 (qtw-like)");
    time([&] { hipLaunchKernelGGL((k_read_slices<4, 2>), dim3(G), dim3(256), 0, 0, a, ld2, nrows, cnt2, out); }, 16.0 * ld2 * nrows, "slices R4 U2");
    time([&] { hipLaunchKernelGGL((k_read_slices<4, 5>), dim3(G), dim3(256), 0, 0, a, ld2, nrows, cnt2, out); }, 16.0 * ld2 * nrows, "slices R4 U5");
    double* part; hipMalloc(&part, (size_t)G * 4 * 512 * 8);
    time([&] { hipLaunchKernelGGL((k_read_slices<8, 2, 1>), dim3(G), dim3(256), 0, 0, a, ld2, nrows, cnt2, out, part); }, 16.0 * ld2 * (nrows / 8 * 8), "slices R8 U2 + partial store per tile");
    time([&] { hipLaunchKernelGGL((k_read_slices<8, 2, 2>), dim3(G), dim3(256), 0, 0, a, ld2, nrows, cnt2, out, part); }, 16.0 * ld2 * (nrows / 8 * 8), "slices R8 U2 + partials in one nt burst");
    // occupancy: the real Q.w kernel holds a 40 KB slice of w in LDS (4 blocks per CU); same stream with that much LDS reserved
    for (int kb : {0, 20, 32, 40, 64}) {
      char nm[64]; snprintf(nm, 64, "slices R8 U2, %d KB LDS reserved", kb);
      time([&] { hipLaunchKernelGGL((k_read_slices<8, 2>), dim3(G), dim3(256), (size_t)kb * 1024, 0, a, ld2, nrows, cnt2, out); }, 16.0 * ld2 * (nrows / 8 * 8), nm);
    }
    // Infinity Cache (256 MB memory-side): the same S megabytes read again and again (20 back-to-back launches)
    for (int mb : {16, 32, 64, 128, 192, 256, 384, 1024}) {
      const size_t nn = (size_t)mb * (1 << 20) / 16;
      for (int ntl = 0; ntl < 2; ++ntl) {
        auto go = [&] { if (ntl) hipLaunchKernelGGL((k_read<8, true>), dim3(4096), dim3(256), 0, 0, a, nn, out); else hipLaunchKernelGGL((k_read<8, false>), dim3(4096), dim3(256), 0, 0, a, nn, out); };
        go(); go(); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int it = 0; it < 20; ++it) go();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("re-read %4d MB x20, %s loads: %7.1f GB/s\n", mb, ntl ? "nt" : "plain", 20.0 * nn * 16 / ms / 1e6);
      }
    }
    // Sustained rate: the headline solve streams for 0.5 s at a time and carries random mantissas, while the numbers
    // above are best-of-5 of ~1 ms launches over a constant byte pattern.  3 x 600 back-to-back launches per pattern.
    for (int pattern = 0; pattern < 2; ++pattern) {
      if (pattern == 1) { hipLaunchKernelGGL(k_fill_random, dim3(8192), dim3(256), 0, 0, reinterpret_cast<double*>(a), bytes / 8); hipDeviceSynchronize(); }
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int it = 0; it < 600; ++it) hipLaunchKernelGGL((k_read_slices<8, 2>), dim3(G), dim3(256), 0, 0, a, ld2, nrows, cnt2, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("sustained slices R8 U2, %s data, 600 launches: %8.1f ms  %7.1f GB/s\n", pattern ? "random" : "constant", ms, 600.0 * 16.0 * ld2 * (nrows / 8 * 8) / ms / 1e6);
      }
    }
  }
  return 0;
}
