// Hand-written gfx950 kernels for the Lanczos hot path.
//
// All kernels here are HBM-bandwidth bound (<= 0.25 flop/byte), so the design
// rules are: 16-byte coalesced accesses, many independent loads in flight per
// lane, deterministic two-stage reductions (wave shuffle -> LDS -> per-block
// partial -> tiny second-stage kernel), and no atomics.
//
// Arithmetic contract (DESIGN.md "numerics"): element-wise results follow the
// reference CPU branch's NumPy expression order with NO fused multiply-add
// (this file is compiled with -ffp-contract=off), so SpMV, the re-orthogonalisation
// update and the three-term recurrence are bit-identical to NumPy/SciPy given
// the same scalar inputs; only the inner products differ (summation order).
#include "lz_internal.h"

namespace lz {

// ------------------------------------------------------------------ helpers
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
template <int VAR>
__device__ __forceinline__ int2 ld_stream(const int2* p) {
  if (VAR == 1) {
    const i2v_t v = __builtin_nontemporal_load(reinterpret_cast<const i2v_t*>(p));
    return make_int2(v.x, v.y);
  }
  return *p;
}

// ------------------------------------------------------------------ second-stage reductions
constexpr int kFinalThreads = 1024;
__global__ __launch_bounds__(kFinalThreads) void k_final_sum(const double* __restrict__ part, int n, double* __restrict__ out) {
  __shared__ double sm[kFinalThreads / 64];
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int i = threadIdx.x;
  for (; i + 3 * kFinalThreads < n; i += 4 * kFinalThreads) {
    a0 += part[i];
    a1 += part[i + kFinalThreads];
    a2 += part[i + 2 * kFinalThreads];
    a3 += part[i + 3 * kFinalThreads];
  }
  for (; i < n; i += kFinalThreads) a0 += part[i];
  double acc = wave_sum((a0 + a1) + (a2 + a3));
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) sm[w] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < kFinalThreads / 64; ++k) t += sm[k];
    out[0] = t;
  }
}

__global__ __launch_bounds__(kTPB) void k_final_rows(const double* __restrict__ part, int G, double* __restrict__ c) {
  __shared__ double sm[kTPB / 64];
  const double* p = part + (int64_t)blockIdx.x * G;
  double acc = 0.0;
  for (int i = threadIdx.x; i < G; i += kTPB) acc += p[i];
  acc = block_sum(acc, sm);
  if (threadIdx.x == 0) c[blockIdx.x] = acc;
}

void launch_final_sum(const double* part, int n, double* out, hipStream_t s) {
  hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(kFinalThreads), 0, s, part, n, out);
}
void launch_final_rows(const double* part, int nrows, int G, double* c, hipStream_t s) {
  if (nrows <= 0) return;
  hipLaunchKernelGGL(k_final_rows, dim3(nrows), dim3(kTPB), 0, s, part, G, c);
}

// ------------------------------------------------------------------ CSR SpMV
// (a) plain one-thread-per-row kernel: baseline / A-B arm.
__global__ __launch_bounds__(kTPB) void k_spmv_scalar(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                                     const double* __restrict__ vals, const double* __restrict__ x,
                                                     const double* __restrict__ xown, double* __restrict__ y, int64_t rows,
                                                     double* __restrict__ part) {
  __shared__ double sm[kTPB / 64];
  const int64_t row = (int64_t)blockIdx.x * kTPB + threadIdx.x;
  double d = 0.0;
  if (row < rows) {
    double sum = 0.0;
    const int a = rowptr[row], b = rowptr[row + 1];
    for (int k = a; k < b; ++k) sum += vals[k] * x[colidx[k]];
    y[row] = sum;
    d = xown[row] * sum;
  }
  d = block_sum(d, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = d;
}

// (b) CSR-stream: the block's contiguous slice of vals/colidx is read with
// 16-byte/8-byte coalesced loads, products are staged in LDS, then each thread
// adds up its rows from LDS in CSR order (same order and rounding as SciPy's
// csr_matvec: sum += a*x, no FMA).  Row blocks are precomputed on the host so
// that one block's products fit the LDS tile.

// ABL != 0 instantiations are timing-only ablation arms for tools/kbench.py (wrong results on purpose):
// 1 = no x gather, 2 = no colidx load, 4 = no vals load, 8 = no y store.
template <int FIXED_K, int ABL = 0>
__global__ __launch_bounds__(kTPB) void k_spmv_stream(const int32_t* __restrict__ rowblk, const int32_t* __restrict__ rowptr,
                                                     const int32_t* __restrict__ colidx, const double* __restrict__ vals,
                                                     const double* __restrict__ x, const double* __restrict__ xown,
                                                     double* __restrict__ y, int fixed_k, int nnz_cap,
                                                     double* __restrict__ part) {
  extern __shared__ double prod[];  // nnz_cap + 2 products
  __shared__ double sm[kTPB / 64];
  const int blk = xcd_remap(blockIdx.x, gridDim.x);
  const int r0 = rowblk[blk], r1 = rowblk[blk + 1];
  const int K = FIXED_K > 0 ? FIXED_K : fixed_k;
  const int k0 = K > 0 ? r0 * K : rowptr[r0];
  const int k1 = K > 0 ? r1 * K : rowptr[r1];
  double d = 0.0;
  if (k1 - k0 <= nnz_cap) {
    // phase 1: products, two entries per lane per step, aligned to even k.  Batches of 4 steps:
    // all (vals, colidx) loads of a batch are issued first, then its 8 x gathers, then the LDS
    // stores - 3 dependent round trips per batch instead of 8.
    const int kk = k0 & ~1;
    const int npair = (k1 - kk + 1) >> 1;
    constexpr int NB = 4;
    for (int pb = threadIdx.x; pb < npair; pb += NB * kTPB) {
      double2 a[NB];
      int2 c[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        int p = pb + kTPB * i;
        if (p >= npair) p = pb;  // clamped duplicate, discarded below
        const int k = kk + 2 * p;
        a[i] = (ABL & 4) ? make_double2(1.0, 2.0) : ld_stream<1>(reinterpret_cast<const double2*>(vals + k));
        c[i] = (ABL & 2) ? make_int2(k / (K > 0 ? K : 1), (k + 1) / (K > 0 ? K : 1)) : ld_stream<1>(reinterpret_cast<const int2*>(colidx + k));
        if (ABL & 1) c[i] = make_int2(c[i].x & 1023, c[i].y & 1023);
      }
      double2 xv[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) xv[i] = make_double2(x[c[i].x], x[c[i].y]);
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int p = pb + kTPB * i;
        if (p < npair) {
          const int k = kk + 2 * p;
          const double p0 = (k >= k0) ? a[i].x * xv[i].x : 0.0;
          const double p1 = (k + 1 < k1) ? a[i].y * xv[i].y : 0.0;
          *reinterpret_cast<double2*>(&prod[2 * p]) = make_double2(p0, p1);
        }
      }
    }
    __syncthreads();
    // phase 2: per-row sequential sums out of LDS
    const int shift = k0 - kk;  // 0 or 1
    for (int row = r0 + threadIdx.x; row < r1; row += kTPB) {
      int a, b;
      if (K > 0) {
        a = (row - r0) * K + shift;
        b = a + K;
      } else {
        a = rowptr[row] - kk;
        b = rowptr[row + 1] - kk;
      }
      double sum = 0.0;
      if (FIXED_K > 0) {
#pragma unroll
        for (int k = 0; k < FIXED_K; ++k) sum += prod[a + k];
      } else {
        for (int k = a; k < b; ++k) sum += prod[k];
      }
      if (!(ABL & 8)) y[row] = sum;
      d += xown[row] * sum;
    }
  } else {
    // long row(s): the host gives such a row a block of its own
    for (int row = r0; row < r1; ++row) {
      const int a = rowptr[row], b = rowptr[row + 1];
      double acc = 0.0;
      for (int k = a + threadIdx.x; k < b; k += kTPB) acc = fma(vals[k], x[colidx[k]], acc);
      acc = block_sum(acc, sm);
      if (threadIdx.x == 0) {
        y[row] = acc;
        d += xown[row] * acc;
      }
    }
  }
  d = block_sum(d, sm);
  if (threadIdx.x == 0) part[blk] = d;
}

// (c) fixed-K rows (stencils): every block owns exactly RB rows = RB*K contiguous entries, so all
// trip counts are compile-time: each lane first issues ALL its 16-byte vals and 8-byte colidx loads
// (NP of each), then all 2*NP x gathers, then stages the products in LDS; after one barrier each
// lane adds up RB/256 rows from LDS in CSR order.  Three dependent memory round trips per block
// instead of 2*NP, and an LDS tile of exactly RB*K products.  rowptr is never read.
template <int K, int RB, int NT = 1>
__global__ __launch_bounds__(kTPB) void k_spmv_fixed(const int32_t* __restrict__ colidx, const double* __restrict__ vals,
                                                    const double* __restrict__ x, const double* __restrict__ xown,
                                                    double* __restrict__ y, int rows, double* __restrict__ part) {
  constexpr int NNZ = RB * K;              // even (RB is a multiple of 256)
  constexpr int NP = NNZ / 2 / kTPB;       // double2 pairs per lane
  constexpr int RPT = RB / kTPB;           // rows per lane
  static_assert(NNZ % (2 * kTPB) == 0, "RB*K must be a multiple of 512");
  __shared__ double prod[NNZ];
  __shared__ double sm[kTPB / 64];
  const int blk = xcd_remap(blockIdx.x, gridDim.x);
  const int r0 = blk * RB;
  const int nr = rows - r0 < RB ? rows - r0 : RB;
  const int64_t k0 = (int64_t)r0 * K;
  const int kcnt = nr * K;                 // entries of this block
  const double2* v2 = reinterpret_cast<const double2*>(vals + k0);
  const int2* c2 = reinterpret_cast<const int2*>(colidx + k0);
  double2 a[NP];
  int2 c[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    int p = threadIdx.x + kTPB * i;
    if (2 * p >= kcnt) p = 0;              // tail block: valid address, product discarded below
    a[i] = ld_stream<NT>(v2 + p);
    c[i] = ld_stream<NT>(c2 + p);
  }
  double2 xv[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) xv[i] = make_double2(x[c[i].x], x[c[i].y]);
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int p = threadIdx.x + kTPB * i;
    double2 pr = make_double2(a[i].x * xv[i].x, a[i].y * xv[i].y);
    if (2 * p >= kcnt) pr = make_double2(0.0, 0.0);
    else if (2 * p + 1 >= kcnt) pr.y = 0.0;
    *reinterpret_cast<double2*>(&prod[2 * p]) = pr;
  }
  __syncthreads();
  double d = 0.0;
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int lr = threadIdx.x + kTPB * q;
    if (lr < nr) {
      double sum = 0.0;
#pragma unroll
      for (int k = 0; k < K; ++k) sum += prod[lr * K + k];
      y[r0 + lr] = sum;
      d += xown[r0 + lr] * sum;
    }
  }
  d = block_sum(d, sm);
  if (threadIdx.x == 0) part[blk] = d;
}

template <int K>
static int launch_spmv_fixed(const CsrDev& A, const double* x, double* y, const double* x_own, double* part, int rb,
                             hipStream_t s) {
  if (rb == 1024) {
    const int grid = (int)((A.rows + 1023) / 1024);
    hipLaunchKernelGGL((k_spmv_fixed<K, 1024>), dim3(grid), dim3(kTPB), 0, s, A.colidx, A.vals, x, x_own, y, (int)A.rows, part);
    return grid;
  }
  const int grid = (int)((A.rows + 511) / 512);
  if (rb == 513)  // A/B arm: plain (cached) loads of the matrix stream
    hipLaunchKernelGGL((k_spmv_fixed<K, 512, 0>), dim3(grid), dim3(kTPB), 0, s, A.colidx, A.vals, x, x_own, y, (int)A.rows, part);
  else
    hipLaunchKernelGGL((k_spmv_fixed<K, 512>), dim3(grid), dim3(kTPB), 0, s, A.colidx, A.vals, x, x_own, y, (int)A.rows, part);
  return grid;
}

int launch_spmv_csr(const CsrDev& A, const double* x, double* y, const double* x_own, double* part, int flags,
                    hipStream_t s) {
  if (A.rows == 0) return 0;
  if (flags & LZ_FLAG_SPMV_SCALAR) {
    const int grid = (int)((A.rows + kTPB - 1) / kTPB);
    hipLaunchKernelGGL(k_spmv_scalar, dim3(grid), dim3(kTPB), 0, s, A.rowptr, A.colidx, A.vals, x, x_own, y, A.rows, part);
    return grid;
  }
  if (!A.ablation && !(flags & LZ_FLAG_SPMV_STREAM)) {
    if (A.fixed_k == 5) return launch_spmv_fixed<5>(A, x, y, x_own, part, A.fixed_rb, s);
    if (A.fixed_k == 7) return launch_spmv_fixed<7>(A, x, y, x_own, part, A.fixed_rb, s);
  }
  const int grid = A.n_rowblk;
  const size_t lds = (size_t)(A.blk_nnz_cap + 2) * sizeof(double);
#define LZ_ABL(n)                                                                                                   \
  case n:                                                                                                          \
    hipLaunchKernelGGL((k_spmv_stream<5, n>), dim3(grid), dim3(kTPB), lds, s, A.rowblk, A.rowptr, A.colidx, A.vals, x, x_own, y, \
                       5, A.blk_nnz_cap, part);                                                                    \
    return grid;
  if (A.fixed_k == 5 && A.ablation) {
    switch (A.ablation) {
      LZ_ABL(1) LZ_ABL(2) LZ_ABL(3) LZ_ABL(4) LZ_ABL(7) LZ_ABL(8) LZ_ABL(15)
      default: break;
    }
  }
#undef LZ_ABL
  if (A.fixed_k == 5)
    hipLaunchKernelGGL(k_spmv_stream<5>, dim3(grid), dim3(kTPB), lds, s, A.rowblk, A.rowptr, A.colidx, A.vals, x, x_own, y, 5,
                       A.blk_nnz_cap, part);
  else if (A.fixed_k == 7)
    hipLaunchKernelGGL(k_spmv_stream<7>, dim3(grid), dim3(kTPB), lds, s, A.rowblk, A.rowptr, A.colidx, A.vals, x, x_own, y, 7,
                       A.blk_nnz_cap, part);
  else
    hipLaunchKernelGGL(k_spmv_stream<0>, dim3(grid), dim3(kTPB), lds, s, A.rowblk, A.rowptr, A.colidx, A.vals, x, x_own, y,
                       A.fixed_k, A.blk_nnz_cap, part);
  return grid;
}

// ------------------------------------------------------------------ dense GEMV (row-major A, one wave per row)
__global__ __launch_bounds__(kTPB) void k_gemv_dense(const double* __restrict__ A, int64_t M, const double* __restrict__ x,
                                                    double* __restrict__ y, double* __restrict__ part) {
  __shared__ double sm[kTPB / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * (kTPB / 64) + w;
  double d = 0.0;
  if (row < M) {
    const double* a = A + row * M;
    double acc = 0.0;
    for (int64_t c = lane; c < M; c += 64) acc = fma(a[c], x[c], acc);
    acc = wave_sum(acc);
    if (lane == 0) {
      y[row] = acc;
      d = x[row] * acc;
    }
  }
  d = block_sum(d, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = d;
}

int launch_gemv_dense(const double* A, int64_t M, const double* x, double* y, double* part, hipStream_t s) {
  const int grid = (int)((M + kTPB / 64 - 1) / (kTPB / 64));
  hipLaunchKernelGGL(k_gemv_dense, dim3(grid), dim3(kTPB), 0, s, A, M, x, y, part);
  return grid;
}

// ------------------------------------------------------------------ re-orthogonalisation pass 1: c = Q^T w
// Block b owns the contiguous slice [b*L, b*L+cnt) of every basis row.  The
// slice of w (= V[j], optionally formed here as r / beta and stored) is kept in
// LDS for the whole launch; the block then streams the same slice of rows
// 0..nrows-1.
//
//  * VALU variant: R rows at a time, each lane accumulating R partial dots over
//    U*R independent 16-byte loads, one shuffle + LDS reduction per R rows.
//  * MFMA variant: each wave owns a quarter of the slice and walks 16-row tiles
//    with v_mfma_f64_16x16x4_f64: A = 16 basis rows x 4 consecutive elements,
//    B = the matching 4 entries of w broadcast over the 16 columns.  The
//    contraction over the long dimension happens inside the matrix core, so the
//    main loop has no cross-lane reduction and no barrier at all; 15/16 of the
//    MFMA columns are redundant, which is affordable because the step is HBM
//    bound (the matrix pipe is ~30 % busy at full HBM rate).
constexpr int kQtwMaxL = 5120;  // 40 KiB of LDS -> 4 blocks per CU
typedef double double4_t __attribute__((ext_vector_type(4)));

// SCALE: 0 = w is V[j] as stored; 1 = w = r / sqrt(nrm2), stored to V[j] (beta to beta_slot);
//        2 = w = r as is, nothing stored (fused-norm mode: the caller divides by beta afterwards).
template <int SCALE>
__device__ __forceinline__ double qtw_stage_w(double* __restrict__ V, int64_t ldv, int j, const double* __restrict__ r,
                                              const double* __restrict__ nrm2, double* __restrict__ beta_slot, int64_t base,
                                              int cnt2, double2* sw) {
  double2* vj = reinterpret_cast<double2*>(V + (int64_t)j * ldv + base);
  double self = 0.0;
  if (SCALE == 1) {
    const double beta = sqrt(nrm2[0]);
    if (blockIdx.x == 0 && threadIdx.x == 0) beta_slot[0] = beta;
    const double2* rr = reinterpret_cast<const double2*>(r + base);
    for (int t = threadIdx.x; t < cnt2; t += kTPB) {
      double2 v = rr[t];
      v.x = v.x / beta;
      v.y = v.y / beta;
      vj[t] = v;
      sw[t] = v;
      self = fma(v.x, v.x, self);
      self = fma(v.y, v.y, self);
    }
  } else {
    const double2* src = SCALE == 2 ? reinterpret_cast<const double2*>(r + base) : vj;
    for (int t = threadIdx.x; t < cnt2; t += kTPB) {
      const double2 v = src[t];
      sw[t] = v;
      self = fma(v.x, v.x, self);
      self = fma(v.y, v.y, self);
    }
  }
  return self;
}

template <int SCALE, int R, int U, int NT = 1>
__global__ __launch_bounds__(kTPB) void k_qtw_valu(double* __restrict__ V, int64_t ldv, int64_t len, int nrows, int j,
                                                  const double* __restrict__ r, const double* __restrict__ nrm2,
                                                  double* __restrict__ beta_slot, int64_t L, int G,
                                                  double* __restrict__ part) {
  extern __shared__ double2 sw[];
  __shared__ double red[R][kTPB / 64];
  const int64_t base = (int64_t)blockIdx.x * L;
  const int cnt2 = (int)((len - base < L ? len - base : L) >> 1);  // double2 count of this slice
  const double self = qtw_stage_w<SCALE>(V, ldv, j, r, nrm2, beta_slot, base, cnt2, sw);
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // Tiles are walked from the newest rows down to row 0: pass 2 (k_update) must add rows in ascending
  // order, so the last ~256 MB this pass reads (rows 0, 1, ...) are the first pass 2 needs - they are still in
  // the Infinity Cache.
  for (int i0 = ((nrows - 1) / R) * R; i0 >= 0; i0 -= R) {
    const double2* row[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
      int i = i0 + q;
      if (i >= nrows) i = nrows - 1;                    // clamped duplicate, result discarded
      if (i == j) i = i > 0 ? i - 1 : (nrows > 1 ? 1 : 0);  // self term comes from LDS: do not stream row j
      row[q] = reinterpret_cast<const double2*>(V + (int64_t)i * ldv + base);
    }
    double acc[R];
#pragma unroll
    for (int q = 0; q < R; ++q) acc[q] = 0.0;
#pragma unroll U
    for (int t = threadIdx.x; t < cnt2; t += kTPB) {
      const double2 wv = sw[t];
#pragma unroll
      for (int q = 0; q < R; ++q) {
        const double2 v = ld_stream<NT>(row[q] + t);
        acc[q] = fma(v.x, wv.x, acc[q]);
        acc[q] = fma(v.y, wv.y, acc[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < R; ++q) {
      if (i0 + q == j) acc[q] = self;  // c_j = w.w from the LDS-resident values
      const double s = wave_sum(acc[q]);
      if (lane == 0) red[q][w] = s;
    }
    __syncthreads();
    if (threadIdx.x < R && i0 + threadIdx.x < nrows) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < kTPB / 64; ++k) s += red[threadIdx.x][k];
      part[(int64_t)(i0 + threadIdx.x) * G + blockIdx.x] = s;
    }
    __syncthreads();
  }
}

// Software-pipelined VALU variant.  The (row-tile, position) items of a block are
// walked as ONE stream with D register buffers of R loads each: the loads of item
// k+D-1 are issued before item k is consumed, also across tile boundaries, so a
// lane always has (D-1)*R .. D*R 16-byte loads in flight and nothing drains at the
// per-tile reduction.  Each wave reduces and writes its own partials (no barrier in
// the main loop): P = 4*G partials per basis row.
template <int SCALE, int R, int D>
__global__ __launch_bounds__(kTPB) void k_qtw_pipe(double* __restrict__ V, int64_t ldv, int64_t len, int nrows, int j,
                                                  const double* __restrict__ r, const double* __restrict__ nrm2,
                                                  double* __restrict__ beta_slot, int64_t L, int P,
                                                  double* __restrict__ part) {
  extern __shared__ double2 sw[];
  const int64_t base = (int64_t)blockIdx.x * L;
  const int cnt2 = (int)((len - base < L ? len - base : L) >> 1);
  const int npos = (cnt2 + kTPB - 1) / kTPB;  // block-uniform
  double self = qtw_stage_w<SCALE>(V, ldv, j, r, nrm2, beta_slot, base, cnt2, sw);
  for (int t = cnt2 + threadIdx.x; t < npos * kTPB; t += kTPB) sw[t] = make_double2(0.0, 0.0);  // weights of the padding lanes
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int pid = blockIdx.x * (kTPB / 64) + w;
  const int ntiles = (nrows + R - 1) / R;
  const int K = ntiles * npos;
  const double2* Vb = reinterpret_cast<const double2*>(V + base);
  const int64_t ld2 = ldv >> 1;
  double2 buf[D][R];
  double acc[R];
#pragma unroll
  for (int q = 0; q < R; ++q) acc[q] = 0.0;
  int ptile = 0, ppos = 0, ctile = 0, cpos = 0;

  auto issue = [&](double2(&b)[R]) {
    int t = threadIdx.x + kTPB * ppos;
    if (t >= cnt2) t = cnt2 - 1;  // padding lane: valid address, zero weight
#pragma unroll
    for (int q = 0; q < R; ++q) {
      int i = ptile * R + q;
      if (i >= nrows) i = nrows - 1;  // clamped duplicate, discarded at the store
      b[q] = Vb[(int64_t)i * ld2 + t];
    }
    if (++ppos == npos) {
      ppos = 0;
      ++ptile;
    }
  };
  auto consume = [&](const double2(&b)[R]) {
    const double2 wv = sw[threadIdx.x + kTPB * cpos];
#pragma unroll
    for (int q = 0; q < R; ++q) {
      acc[q] = fma(b[q].x, wv.x, acc[q]);
      acc[q] = fma(b[q].y, wv.y, acc[q]);
    }
    if (++cpos == npos) {
#pragma unroll
      for (int q = 0; q < R; ++q) {
        const int row = ctile * R + q;
        const double s = wave_sum(row == j ? self : acc[q]);  // c_j = w.w from the LDS-resident values
        if (lane == 0 && row < nrows) part[(int64_t)row * P + pid] = s;
        acc[q] = 0.0;
      }
      cpos = 0;
      ++ctile;
    }
  };

#pragma unroll
  for (int d = 0; d < D - 1; ++d)
    if (d < K) issue(buf[d]);
  for (int k = 0; k < K; k += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (k + d + D - 1 < K) issue(buf[(d + D - 1) % D]);
      if (k + d < K) consume(buf[d]);
    }
  }
}

// MFMA variant.  Lane l of a wave: row r = l & 15 of the current 16-row tile,
// k-group g = l >> 4.  Step s covers 8 consecutive elements of the slice
// (64 B per row): the lane loads the double2 at element 8*s + 2*g of its row;
// the two halves feed two MFMAs whose B operands are the matching w entries
// (one ds_read_b128 per step, 4 distinct addresses per wave -> conflict free).
// D layout (f64 16x16x4): lane l holds D[row = (l>>4) + 4*reg][col = l&15];
// all 16 columns are equal, column-0 lanes write the per-wave partials.
template <int SCALE, int U, int T>
__global__ __launch_bounds__(kTPB) void k_qtw_mfma(double* __restrict__ V, int64_t ldv, int64_t len, int nrows, int j,
                                                  const double* __restrict__ r, const double* __restrict__ nrm2,
                                                  double* __restrict__ beta_slot, int64_t L, int P,
                                                  double* __restrict__ part) {
  extern __shared__ double2 sw[];
  const int64_t base = (int64_t)blockIdx.x * L;
  const int cnt = (int)(len - base < L ? len - base : L);
  static_assert(SCALE != 2, "the MFMA kernel streams row j from V[j]; fused-norm mode uses the VALU kernel");
  qtw_stage_w<SCALE>(V, ldv, j, r, nrm2, beta_slot, base, cnt >> 1, sw);
  __syncthreads();  // also makes this block's V[j] stores visible to its own later loads
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const int sub = (int)(L >> 2);                 // elements per wave (multiple of 128)
  const int m_lo = w * sub;
  int m_hi = m_lo + sub;
  if (m_hi > cnt) m_hi = cnt;
  const int nsteps = m_hi > m_lo ? (m_hi - m_lo) >> 3 : 0;  // multiple of 4 (cnt and sub are multiples of 32)
  const int pid = blockIdx.x * (kTPB / 64) + w;
  const double2* swl = sw + (m_lo >> 1) + g;     // + 4*s per step
  for (int i0 = ((nrows - 1) / (16 * T)) * (16 * T); i0 >= 0; i0 -= 16 * T) {  // newest rows first (see k_qtw_valu)
    const double2* a[T];
    double4_t acc[T][2];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      int i = i0 + 16 * t + lr;
      if (i >= nrows) i = nrows - 1;             // clamped duplicate, discarded at the store
      a[t] = reinterpret_cast<const double2*>(V + (int64_t)i * ldv + base + m_lo) + g;
      acc[t][0] = (double4_t){0.0, 0.0, 0.0, 0.0};
      acc[t][1] = (double4_t){0.0, 0.0, 0.0, 0.0};
    }
    for (int s0 = 0; s0 < nsteps; s0 += U) {
      double2 av[T][U];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int t = 0; t < T; ++t) av[t][u] = (s0 + u < nsteps) ? ld_stream<1>(a[t] + 4 * (s0 + u)) : make_double2(0.0, 0.0);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const double2 bv = (s0 + u < nsteps) ? swl[4 * (s0 + u)] : make_double2(0.0, 0.0);
#pragma unroll
        for (int t = 0; t < T; ++t) {
          acc[t][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t][u].x, bv.x, acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t][u].y, bv.y, acc[t][1], 0, 0, 0);
        }
      }
    }
    if (lr == 0) {
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = i0 + 16 * t + g + 4 * q;
          if (row < nrows) part[(int64_t)row * P + pid] = acc[t][0][q] + acc[t][1][q];
        }
    }
  }
}

QtwPlan plan_qtw(int64_t len, int flags, const int* tune) {
  QtwPlan p;
  int64_t target = (tune && tune[0] > 0) ? tune[0] : 0;
  // Measured on MI355X (profiles/r01/ab_qtw_slice_small.json): long slices amortise the per-tile reduction, as long
  // as there are still a few blocks per CU: 5120 (40 KiB LDS, 4 blocks/CU) for >= 1024 blocks, else 2560, 1024, 512.
  int64_t L;
  if (target > 0) L = round_up(target, 512);
  else if (len >= (int64_t)5120 * 1024) L = 5120;
  else if (len >= (int64_t)2560 * 384) L = 2560;
  else if (len >= (int64_t)1024 * 256) L = 1024;
  else L = 512;
  if (L > kQtwMaxL) L = kQtwMaxL;
  p.L = L;
  p.G = (int)((len + L - 1) / L);
  p.mfma = (flags & LZ_FLAG_QTW_MFMA) != 0 && (flags & LZ_FLAG_QTW_VALU) == 0;
  p.variant = tune ? tune[1] : 0;
  p.P = (p.mfma || p.variant == 8 || p.variant == 9) ? p.G * (kTPB / 64) : p.G;
  return p;
}

template <int SCALE>
static void launch_qtw_t(double* V, int64_t ldv, int64_t len, int nrows, int j, const double* r, const double* nrm2,
                         double* beta_slot, const QtwPlan& plan, double* part, hipStream_t s) {
  const size_t lds = (size_t)plan.L * sizeof(double);
  const dim3 grid(plan.G), block(kTPB);
#define LZ_QTW_ARGS V, ldv, len, nrows, j, r, nrm2, beta_slot, plan.L, plan.P, part
  if constexpr (SCALE != 2) {
    if (plan.mfma) {
      switch (plan.variant) {
        case 1: hipLaunchKernelGGL((k_qtw_mfma<SCALE, 4, 1>), grid, block, lds, s, LZ_QTW_ARGS); break;
        case 2: hipLaunchKernelGGL((k_qtw_mfma<SCALE, 16, 1>), grid, block, lds, s, LZ_QTW_ARGS); break;
        case 3: hipLaunchKernelGGL((k_qtw_mfma<SCALE, 4, 2>), grid, block, lds, s, LZ_QTW_ARGS); break;
        case 4: hipLaunchKernelGGL((k_qtw_mfma<SCALE, 8, 2>), grid, block, lds, s, LZ_QTW_ARGS); break;
        default: hipLaunchKernelGGL((k_qtw_mfma<SCALE, 8, 1>), grid, block, lds, s, LZ_QTW_ARGS); break;
      }
      return;
    }
  }
  switch (plan.variant) {
    case 1: hipLaunchKernelGGL((k_qtw_valu<SCALE, 4, 2>), grid, block, lds, s, LZ_QTW_ARGS); break;
    case 6: hipLaunchKernelGGL((k_qtw_valu<SCALE, 16, 2>), grid, block, lds, s, LZ_QTW_ARGS); break;
    case 7: hipLaunchKernelGGL((k_qtw_valu<SCALE, 8, 2, 0>), grid, block, lds, s, LZ_QTW_ARGS); break;  // plain (cached) loads
    case 8: hipLaunchKernelGGL((k_qtw_pipe<SCALE, 8, 2>), grid, block, lds, s, LZ_QTW_ARGS); break;
    case 9: hipLaunchKernelGGL((k_qtw_pipe<SCALE, 8, 3>), grid, block, lds, s, LZ_QTW_ARGS); break;
    default:  // measured best on MI355X (profiles/r01): 8 rows x 2 positions = 16 loads in flight per lane
      if (nrows > 4)
        hipLaunchKernelGGL((k_qtw_valu<SCALE, 8, 2>), grid, block, lds, s, LZ_QTW_ARGS);
      else
        hipLaunchKernelGGL((k_qtw_valu<SCALE, 4, 2>), grid, block, lds, s, LZ_QTW_ARGS);
      break;
  }
#undef LZ_QTW_ARGS
}

void launch_qtw(double* V, int64_t ldv, int64_t len, int nrows, int j, const double* r, const double* nrm2,
                double* beta_slot, const QtwPlan& plan, double* part, int mode, hipStream_t s) {
  if (mode == 2)
    launch_qtw_t<2>(V, ldv, len, nrows, j, r, nrm2, beta_slot, plan, part, s);
  else if (mode == 1)
    launch_qtw_t<1>(V, ldv, len, nrows, j, r, nrm2, beta_slot, plan, part, s);
  else
    launch_qtw_t<0>(V, ldv, len, nrows, j, r, nrm2, beta_slot, plan, part, s);
}

// ------------------------------------------------------------------ re-orthogonalisation pass 2
// V[j] = 2 V[j] - (((c0 V0 + c1 V1) + c2 V2) + ...): NumPy's axis-0 reduction
// order of np.sum(c[:, None] * V, axis=0) (Lanczos.py:249), products and sums
// rounded separately.  One double2 column position per lane; the row loop is
// unrolled so 8 independent 16-byte loads are in flight per lane.
template <bool FUSED, int VAR = 0, int UN = 8>
__global__ __launch_bounds__(kTPB) void k_update(double* __restrict__ V, int64_t ldv, int64_t n2, int nrows, int j,
                                                const double* __restrict__ c, const double* __restrict__ r,
                                                const double* __restrict__ beta) {
  const int64_t i = (int64_t)blockIdx.x * kTPB + threadIdx.x;
  if (i >= n2) return;
  const double2* col = reinterpret_cast<const double2*>(V) + i;
  const int64_t ld2 = ldv >> 1;
  double2* out = reinterpret_cast<double2*>(V) + (int64_t)j * ld2 + i;
  double2 w;
  if (FUSED) {  // fused-norm mode: V[j] has not been formed yet, w = r / beta
    const double b = beta[0];
    w = reinterpret_cast<const double2*>(r)[i];
    w.x = w.x / b;
    w.y = w.y / b;
  }
  double tx = 0.0, ty = 0.0;
  int k = 0;
  for (; k + UN <= nrows; k += UN) {
    double2 q[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) q[u] = (FUSED && k + u == j) ? w : ld_stream<VAR>(col + (int64_t)(k + u) * ld2);
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const double ck = c[k + u];
      tx = tx + ck * q[u].x;
      ty = ty + ck * q[u].y;
    }
  }
  for (; k < nrows; ++k) {
    const double2 q = (FUSED && k == j) ? w : col[(int64_t)k * ld2];
    const double ck = c[k];
    tx = tx + ck * q.x;
    ty = ty + ck * q.y;
  }
  const double2 v = FUSED ? w : *out;
  *out = make_double2(2.0 * v.x - tx, 2.0 * v.y - ty);
}

void launch_update(double* V, int64_t ldv, int64_t len, int nrows, int j, const double* c, const double* r_fused,
                   const double* beta, int variant, hipStream_t s) {
  const int64_t n2 = len >> 1;
  const int grid = (int)((n2 + kTPB - 1) / kTPB);
  if (!r_fused && variant == 1) {  // A/B arm: plain (cached) loads
    hipLaunchKernelGGL((k_update<false, 0, 8>), dim3(grid), dim3(kTPB), 0, s, V, ldv, n2, nrows, j, c, r_fused, beta);
    return;
  }
  if (!r_fused && variant == 2) {
    hipLaunchKernelGGL((k_update<false, 0, 16>), dim3(grid), dim3(kTPB), 0, s, V, ldv, n2, nrows, j, c, r_fused, beta);
    return;
  }
  if (!r_fused && variant == 3) {
    hipLaunchKernelGGL((k_update<false, 1, 16>), dim3(grid), dim3(kTPB), 0, s, V, ldv, n2, nrows, j, c, r_fused, beta);
    return;
  }
  if (r_fused)
    hipLaunchKernelGGL((k_update<true, 1, 8>), dim3(grid), dim3(kTPB), 0, s, V, ldv, n2, nrows, j, c, r_fused, beta);
  else
    hipLaunchKernelGGL((k_update<false, 1, 8>), dim3(grid), dim3(kTPB), 0, s, V, ldv, n2, nrows, j, c, r_fused, beta);
}

// fused-norm mode: c holds the all-reduced [V_0.r, ..., V_{j-1}.r, r.r]; turn it into the coefficients of
// w = r / beta:  beta = sqrt(r.r), c_i /= beta, c_j = (r.r) / beta^2
__global__ void k_fused_prepare(double* __restrict__ c, int j, double* __restrict__ beta_slot) {
  const double rr = c[j];
  const double b = sqrt(rr);
  for (int i = threadIdx.x; i < j; i += blockDim.x) c[i] = c[i] / b;
  __syncthreads();
  if (threadIdx.x == 0) {
    c[j] = rr / (b * b);
    beta_slot[0] = b;
  }
}
void launch_fused_prepare(double* c, int j, double* beta_slot, hipStream_t s) {
  hipLaunchKernelGGL(k_fused_prepare, dim3(1), dim3(kTPB), 0, s, c, j, beta_slot);
}

// ------------------------------------------------------------------ three-term recurrence + ||r||^2
// r = (r - v_j * alpha) - v_{j-1} * beta   (Lanczos.py:119, NumPy evaluation order)
__global__ __launch_bounds__(kTPB) void k_three_term(double* __restrict__ r, const double* __restrict__ vj,
                                                    const double* __restrict__ vjm1, const double* __restrict__ alpha,
                                                    const double* __restrict__ beta, int64_t n2, double* __restrict__ part) {
  __shared__ double sm[kTPB / 64];
  const double a = alpha[0];
  const double b = vjm1 ? beta[0] : 0.0;
  double2* r2 = reinterpret_cast<double2*>(r);
  const double2* v2 = reinterpret_cast<const double2*>(vj);
  const double2* m2 = reinterpret_cast<const double2*>(vjm1);
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kTPB + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kTPB) {
    double2 x = r2[i];
    const double2 v = ld_stream<1>(v2 + i);
    x.x = x.x - v.x * a;
    x.y = x.y - v.y * a;
    if (vjm1) {
      const double2 m = ld_stream<1>(m2 + i);
      x.x = x.x - m.x * b;
      x.y = x.y - m.y * b;
    }
    r2[i] = x;
    acc = fma(x.x, x.x, acc);
    acc = fma(x.y, x.y, acc);
  }
  acc = block_sum(acc, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

int launch_three_term(double* r, const double* vj, const double* vjm1, const double* alpha, const double* beta,
                      int64_t len, double* part, hipStream_t s) {
  const int64_t n2 = len >> 1;
  int64_t g = (n2 + kTPB - 1) / kTPB;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(k_three_term, dim3((int)g), dim3(kTPB), 0, s, r, vj, vjm1, alpha, beta, n2, part);
  return (int)g;
}

// ------------------------------------------------------------------ halo pack
__global__ __launch_bounds__(kTPB) void k_gather(const double* __restrict__ x, const int32_t* __restrict__ idx, int64_t n,
                                                double* __restrict__ buf) {
  const int64_t i = (int64_t)blockIdx.x * kTPB + threadIdx.x;
  if (i < n) buf[i] = x[idx[i]];
}
void launch_gather(const double* x, const int32_t* idx, int64_t n, double* buf, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_gather, dim3((int)((n + kTPB - 1) / kTPB)), dim3(kTPB), 0, s, x, idx, n, buf);
}

// ------------------------------------------------------------------ Ritz back-transform (FP64 MFMA)
// Y[m][i] = sum_k V[k][m] * S[k][i].  v_mfma_f64_16x16x4_f64: lane l supplies
// A[row = l&15][k = l>>4] and B[k = l>>4][col = l&15]; the 4 results per lane are
// D[row = (l>>4) + 4*reg][col = l&15].  One wave owns a 32(m) x 64(i) tile
// (8 accumulators), the 4 waves of a block stack along m.
// C[z][m][i] = sum_{k in K-chunk z} A[k][m] * B[k][i]  ("TN" product of two row-major, k-major operands).
//   Ritz back-transform: A = V (k = basis index, m = matrix row), B = S, one chunk:  Y = V^T S.
//   Gram matrix        : A = B = Y (k = matrix row), split over gridDim.z chunks:   G = Y^T Y.
// One wave owns 32 rows (m) x NT*16 columns (all of them when ncols <= 256), so A is streamed from HBM exactly
// once per column group; B rows are re-read by every wave (L2).  Operands are prefetched in registers: B one
// k-step ahead, A (the HBM stream) four k-steps ahead.  B must be readable up to 15 doubles past a row's end.
template <int NT>
__global__ __launch_bounds__(kTPB) void k_gemm_tn(const double* __restrict__ A, int64_t lda, int64_t mdim, int64_t kcount,
                                                 int64_t kchunk, const double* __restrict__ B, int64_t ldb, int ncols,
                                                 double* __restrict__ C, int64_t ldc, int64_t zstride) {
  constexpr int PA = 4;  // A prefetch distance (k-steps)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  const int64_t m0 = ((int64_t)blockIdx.x * (kTPB / 64) + w) * 32;
  if (m0 >= mdim) return;  // wave-uniform
  const int64_t k_lo = (int64_t)blockIdx.z * kchunk;
  const int64_t k_hi = k_lo + kchunk < kcount ? k_lo + kchunk : kcount;
  const int nsteps = k_hi > k_lo ? (int)((k_hi - k_lo + 3) >> 2) : 0;
  const int ct0 = blockIdx.y * NT;
  const int CT = (ncols + 15) / 16;
  int64_t ma = m0 + lr, mb = m0 + 16 + lr;
  if (ma >= mdim) ma = mdim - 1;
  if (mb >= mdim) mb = mdim - 1;
  int colb[NT];
#pragma unroll
  for (int b = 0; b < NT; ++b) colb[b] = 16 * (ct0 + b < CT ? ct0 + b : CT - 1) + lr;
  double4_t acc[2][NT];
#pragma unroll
  for (int b = 0; b < NT; ++b) {
    acc[0][b] = (double4_t){0.0, 0.0, 0.0, 0.0};
    acc[1][b] = (double4_t){0.0, 0.0, 0.0, 0.0};
  }
  auto load_a = [&](int step, double& x0, double& x1) {
    int64_t kr = k_lo + 4 * (int64_t)step + lk;
    const bool ok = kr < k_hi;  // rows past the chunk contribute zero whatever B holds there
    if (!ok) kr = k_hi - 1;
    const double* ar = A + kr * lda;
    x0 = ok ? ar[ma] : 0.0;
    x1 = ok ? ar[mb] : 0.0;
  };
  auto brow = [&](int step) {
    int64_t kr = k_lo + 4 * (int64_t)step + lk;
    if (kr >= k_hi) kr = k_hi - 1;
    return B + kr * ldb;
  };
  double ra[PA][2];
#pragma unroll
  for (int p = 0; p < PA; ++p) {
    ra[p][0] = ra[p][1] = 0.0;
    if (nsteps > 0) load_a(p < nsteps ? p : nsteps - 1, ra[p][0], ra[p][1]);
  }
  double bcur[NT];
  if (nsteps > 0) {
    const double* sr = brow(0);
#pragma unroll
    for (int b = 0; b < NT; ++b) bcur[b] = sr[colb[b]];
  }
  for (int s0 = 0; s0 < nsteps; s0 += PA) {
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      const int st = s0 + p;
      if (st < nsteps) {
        const double a0 = ra[p][0], a1 = ra[p][1];
        load_a(st + PA < nsteps ? st + PA : nsteps - 1, ra[p][0], ra[p][1]);  // refill this ring slot
        double bnxt[NT];
        {
          const double* sr = brow(st + 1 < nsteps ? st + 1 : st);
#pragma unroll
          for (int b = 0; b < NT; ++b) bnxt[b] = sr[colb[b]];
        }
#pragma unroll
        for (int b = 0; b < NT; ++b) {
          acc[0][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bcur[b], acc[0][b], 0, 0, 0);
          acc[1][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bcur[b], acc[1][b], 0, 0, 0);
        }
#pragma unroll
        for (int b = 0; b < NT; ++b) bcur[b] = bnxt[b];
      }
    }
  }
  double* Cz = C + (int64_t)blockIdx.z * zstride;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      if (ct0 + b >= CT) continue;
      const int col = 16 * (ct0 + b) + lr;
      if (col >= ncols) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int64_t m = m0 + 16 * a + lk + 4 * g;
        if (m < mdim) Cz[m * ldc + col] = acc[a][b][g];
      }
    }
}

static void launch_gemm_tn(const double* A, int64_t lda, int64_t mdim, int64_t kcount, int64_t kchunk, int nz, const double* B,
                           int64_t ldb, int ncols, double* C, int64_t ldc, int64_t zstride, hipStream_t s) {
  const int CT = (ncols + 15) / 16;
  const int ngroups = (CT + 15) / 16;
  const int NT = (CT + ngroups - 1) / ngroups;
  dim3 grid((unsigned)((mdim + 127) / 128), (unsigned)ngroups, (unsigned)nz);
#define LZ_TN(nt)                                                                                                          \
  case nt:                                                                                                                 \
    hipLaunchKernelGGL((k_gemm_tn<nt>), grid, dim3(kTPB), 0, s, A, lda, mdim, kcount, kchunk, B, ldb, ncols, C, ldc, zstride); \
    break;
  switch (NT) {
    LZ_TN(1) LZ_TN(2) LZ_TN(3) LZ_TN(4) LZ_TN(5) LZ_TN(6) LZ_TN(7) LZ_TN(8)
    LZ_TN(9) LZ_TN(10) LZ_TN(11) LZ_TN(12) LZ_TN(13) LZ_TN(14) LZ_TN(15) LZ_TN(16)
    default: break;
  }
#undef LZ_TN
}

void launch_ritz_gemm(const double* V, int64_t ldv, int64_t rows, int n, const double* Spad, int npad, double* Y,
                      int64_t ldy, hipStream_t s) {
  launch_gemm_tn(V, ldv, rows, n, n, 1, Spad, npad, n, Y, ldy, 0, s);
}

// out[i] = sum_z part[z*count + i] (fixed order)
__global__ __launch_bounds__(kTPB) void k_sum_slices(const double* __restrict__ part, int nz, int64_t count,
                                                    double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kTPB + threadIdx.x;
  if (i >= count) return;
  double a0 = 0.0, a1 = 0.0;
  int z = 0;
  for (; z + 1 < nz; z += 2) {
    a0 += part[(int64_t)z * count + i];
    a1 += part[(int64_t)(z + 1) * count + i];
  }
  if (z < nz) a0 += part[(int64_t)z * count + i];
  out[i] = a0 + a1;
}
void launch_sum_slices(const double* part, int nz, int64_t count, double* out, hipStream_t s) {
  hipLaunchKernelGGL(k_sum_slices, dim3((unsigned)((count + kTPB - 1) / kTPB)), dim3(kTPB), 0, s, part, nz, count, out);
}

// G = Y^T Y as nz K-chunk partials (n x n each) in `part`; returns nz.  Y needs 16 doubles of slack at its end.
int launch_gram(const double* Y, int64_t ldy, int64_t rows, int n, double* part, int nz_max, hipStream_t s) {
  int nz = (int)((rows + 2047) / 2048);
  if (nz > nz_max) nz = nz_max;
  if (nz < 1) nz = 1;
  int64_t kchunk = (rows + nz - 1) / nz;
  kchunk = (kchunk + 3) & ~(int64_t)3;
  nz = (int)((rows + kchunk - 1) / kchunk);
  launch_gemm_tn(Y, ldy, n, rows, kchunk, nz, Y, ldy, n, part, n, (int64_t)n * n, s);
  return nz;
}

// Eigenvector quality sums for every Ritz vector at once (print_good_eigs, Lanczos.py:169-175):
//   z = A y_i ;  s1_i = z . y_i ;  s2_i = z . z        (quality_i = s1_i^2 / s2_i)
// Y is (rows x n) row-major, so row r of A Y is a sum of whole rows of Y: each lane owns columns i, i+256, ...
// and walks the block's rows; matrix entries are wave-uniform (scalar) loads, Y rows are coalesced 16-row... reads.
__global__ __launch_bounds__(kTPB) void k_ritz_quality(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                                      const double* __restrict__ vals, const double* __restrict__ Y,
                                                      int64_t ldy, int64_t rows, int n, int rows_per_block,
                                                      double* __restrict__ part) {
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  for (int i = threadIdx.x; i < n; i += kTPB) {
    double s1 = 0.0, s2 = 0.0;
    for (int64_t r = r0; r < r1; ++r) {
      const int a = rowptr[r], b = rowptr[r + 1];
      double z = 0.0;
      for (int k = a; k < b; ++k) z = fma(vals[k], Y[(int64_t)colidx[k] * ldy + i], z);
      s1 = fma(z, Y[r * ldy + i], s1);
      s2 = fma(z, z, s2);
    }
    part[(int64_t)blockIdx.x * 2 * n + i] = s1;
    part[(int64_t)blockIdx.x * 2 * n + n + i] = s2;
  }
}
int launch_ritz_quality(const CsrDev& A, const double* Y, int64_t ldy, int n, double* part, hipStream_t s) {
  const int rpb = 2048;
  const int grid = (int)((A.rows + rpb - 1) / rpb);
  hipLaunchKernelGGL(k_ritz_quality, dim3(grid), dim3(kTPB), 0, s, A.rowptr, A.colidx, A.vals, Y, ldy, A.rows, n, rpb, part);
  return grid;
}

}  // namespace lz
