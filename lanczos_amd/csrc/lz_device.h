// Device-side helpers shared by the kernel translation units (gfx950: wave = 64 lanes).
#pragma once

#include "lz_internal.h"

namespace lz {

typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;  // lane 0 holds the sum
}

// sum over the block; result valid in thread 0.  `sm` has kTPB/64 doubles.
__device__ __forceinline__ double block_sum(double v, double* sm) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) sm[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < kTPB / 64; ++i) t += sm[i];
  }
  __syncthreads();
  return t;
}

// Bijective XCD-aware remap: blocks b, b+8, b+16, ... share an XCD (round-robin
// dispatch), so give each XCD one contiguous band of tiles -> neighbouring
// tiles (which re-use the same x entries in a stencil SpMV) share an L2.
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
  const int q = nwg / kNumXCD, r = nwg % kNumXCD, x = b % kNumXCD;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + b / kNumXCD;
}

typedef double d2v_t __attribute__((ext_vector_type(2)));
typedef int i2v_t __attribute__((ext_vector_type(2)));
// Streamed-once data (basis rows, matrix entries) is loaded non-temporally so it does not push the
// re-used data (x, w, the most recent basis rows) out of L2 / Infinity Cache.  VAR == 0: plain load.
template <int VAR>
__device__ __forceinline__ double2 ld_stream(const double2* p) {
  if (VAR == 1) {
    const d2v_t v = __builtin_nontemporal_load(reinterpret_cast<const d2v_t*>(p));
    return make_double2(v.x, v.y);
  }
  return *p;
}
// Non-temporal 16-byte store: results that are not re-read soon (the new basis row) should not sit dirty in L2 while the
// other rows stream through it (tools/probes/hbm_read_peak.hip: a trailing plain store costs 17 % of the pass, an nt one 9 %).
template <int VAR>
__device__ __forceinline__ void st_stream(double2* p, double2 v) {
  if (VAR == 1) {
    d2v_t t;
    t.x = v.x;
    t.y = v.y;
    __builtin_nontemporal_store(t, reinterpret_cast<d2v_t*>(p));
  } else {
    *p = v;
  }
}
template <int VAR>
__device__ __forceinline__ int2 ld_stream(const int2* p) {
  if (VAR == 1) {
    const i2v_t v = __builtin_nontemporal_load(reinterpret_cast<const i2v_t*>(p));
    return make_int2(v.x, v.y);
  }
  return *p;
}


// out = sum(part[0..n)) with EXACTLY the grouping of k_final_sum (1024 threads: four strided accumulators per thread,
// wave shuffle tree, 16 wave sums added in order), evaluated by a 256-thread block: every real thread plays four of the
// 1024 virtual ones.  Used where a consumer kernel folds the second-stage reduction into its prologue (small problems:
// one launch less per reduction, same bits).  sm16: 16 doubles of LDS.  Every thread returns the sum.
__device__ __forceinline__ double final_sum_emulated(const double* __restrict__ part, int n, double* sm16) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  constexpr int FT = 1024;
  for (int vw = w; vw < FT / 64; vw += kTPB / 64) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int i = vw * 64 + lane;
    for (; i + 3 * FT < n; i += 4 * FT) {
      a0 += part[i];
      a1 += part[i + FT];
      a2 += part[i + 2 * FT];
      a3 += part[i + 3 * FT];
    }
    for (; i < n; i += FT) a0 += part[i];
    const double acc = wave_sum((a0 + a1) + (a2 + a3));
    if (lane == 0) sm16[vw] = acc;
  }
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int k = 0; k < FT / 64; ++k) t += sm16[k];
  __syncthreads();
  return t;
}

// One wave, one dense row (row-major, 16-byte aligned): 16-byte non-temporal loads of the row, x through L1/L2, four
// independent accumulators per lane, wave-shuffle reduction; the sum is valid in lane 0.  Shared by k_gemv_dense and the
// small-problem engine (lz_small.hip) - the SAME arithmetic, so both produce the same bits.
__device__ __forceinline__ double gemv_row_wave(const double* __restrict__ a, const double* __restrict__ x, int64_t cols, int lane) {
  double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
  int64_t c = 0;
  {
    const double2* a2 = reinterpret_cast<const double2*>(a);
    const double2* x2 = reinterpret_cast<const double2*>(x);
    const int64_t m2 = cols >> 1;
    int64_t p = lane;
    for (; p + 448 < m2; p += 512) {  // eight 16-byte row loads in flight per lane
      double2 u[8], xv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) u[q] = ld_stream<1>(a2 + p + 64 * q);
#pragma unroll
      for (int q = 0; q < 8; ++q) xv[q] = x2[p + 64 * q];
#pragma unroll
      for (int q = 0; q < 8; q += 2) {
        acc0 = fma(u[q].x, xv[q].x, acc0);
        acc1 = fma(u[q].y, xv[q].y, acc1);
        acc2 = fma(u[q + 1].x, xv[q + 1].x, acc2);
        acc3 = fma(u[q + 1].y, xv[q + 1].y, acc3);
      }
    }
    for (; p + 64 < m2; p += 128) {
      const double2 u = ld_stream<1>(a2 + p), v = ld_stream<1>(a2 + p + 64);
      const double2 xu = x2[p], xv = x2[p + 64];
      acc0 = fma(u.x, xu.x, acc0);
      acc1 = fma(u.y, xu.y, acc1);
      acc2 = fma(v.x, xv.x, acc2);
      acc3 = fma(v.y, xv.y, acc3);
    }
    for (; p < m2; p += 64) {
      const double2 u = ld_stream<1>(a2 + p);
      const double2 xu = x2[p];
      acc0 = fma(u.x, xu.x, acc0);
      acc1 = fma(u.y, xu.y, acc1);
    }
    c = 2 * m2;
  }
  for (c += lane; c < cols; c += 64) acc0 = fma(a[c], x[c], acc0);  // odd column count: the last column
  return wave_sum((acc0 + acc1) + (acc2 + acc3));
}

}  // namespace lz
