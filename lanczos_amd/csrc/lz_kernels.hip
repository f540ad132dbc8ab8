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

// ------------------------------------------------------------------ second-stage reductions
__global__ __launch_bounds__(kTPB) void k_final_sum(const double* __restrict__ part, int n, double* __restrict__ out) {
  __shared__ double sm[kTPB / 64];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += kTPB) acc += part[i];
  acc = block_sum(acc, sm);
  if (threadIdx.x == 0) out[0] = acc;
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
  hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(kTPB), 0, s, part, n, out);
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
constexpr int kStreamNnz = 4096;   // products per block tile (32 KiB of LDS)
constexpr int kStreamRows = 512;   // max rows per block

template <int FIXED_K>
__global__ __launch_bounds__(kTPB) void k_spmv_stream(const int32_t* __restrict__ rowblk, const int32_t* __restrict__ rowptr,
                                                     const int32_t* __restrict__ colidx, const double* __restrict__ vals,
                                                     const double* __restrict__ x, const double* __restrict__ xown,
                                                     double* __restrict__ y, int fixed_k, double* __restrict__ part) {
  __shared__ double prod[kStreamNnz + 2];
  __shared__ double sm[kTPB / 64];
  const int blk = xcd_remap(blockIdx.x, gridDim.x);
  const int r0 = rowblk[blk], r1 = rowblk[blk + 1];
  const int K = FIXED_K > 0 ? FIXED_K : fixed_k;
  const int k0 = K > 0 ? r0 * K : rowptr[r0];
  const int k1 = K > 0 ? r1 * K : rowptr[r1];
  double d = 0.0;
  if (k1 - k0 <= kStreamNnz) {
    // phase 1: products, two entries per lane per step, aligned to even k
    const int kk = k0 & ~1;
    const int npair = (k1 - kk + 1) >> 1;
    for (int p = threadIdx.x; p < npair; p += kTPB) {
      const int k = kk + 2 * p;
      const double2 a = *reinterpret_cast<const double2*>(vals + k);
      const int2 c = *reinterpret_cast<const int2*>(colidx + k);
      const double p0 = (k >= k0) ? a.x * x[c.x] : 0.0;
      const double p1 = (k + 1 < k1) ? a.y * x[c.y] : 0.0;
      *reinterpret_cast<double2*>(&prod[2 * p]) = make_double2(p0, p1);
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
      y[row] = sum;
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

int launch_spmv_csr(const CsrDev& A, const double* x, double* y, const double* x_own, double* part, int flags,
                    hipStream_t s) {
  if (A.rows == 0) return 0;
  if (flags & LZ_FLAG_SPMV_SCALAR) {
    const int grid = (int)((A.rows + kTPB - 1) / kTPB);
    hipLaunchKernelGGL(k_spmv_scalar, dim3(grid), dim3(kTPB), 0, s, A.rowptr, A.colidx, A.vals, x, x_own, y, A.rows, part);
    return grid;
  }
  const int grid = A.n_rowblk;
  if (A.fixed_k == 5)
    hipLaunchKernelGGL(k_spmv_stream<5>, dim3(grid), dim3(kTPB), 0, s, A.rowblk, A.rowptr, A.colidx, A.vals, x, x_own, y, 5, part);
  else if (A.fixed_k == 7)
    hipLaunchKernelGGL(k_spmv_stream<7>, dim3(grid), dim3(kTPB), 0, s, A.rowblk, A.rowptr, A.colidx, A.vals, x, x_own, y, 7, part);
  else
    hipLaunchKernelGGL(k_spmv_stream<0>, dim3(grid), dim3(kTPB), 0, s, A.rowblk, A.rowptr, A.colidx, A.vals, x, x_own, y,
                       A.fixed_k, part);
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
// LDS; the block then streams the same slice of rows 0..nrows-1 (row j itself
// comes from LDS), R rows at a time, each lane accumulating R partial dots, and
// reduces them once per R rows.
constexpr int kQtwR = 4;
constexpr int kQtwMaxL = 5120;  // 40 KiB of LDS -> 4 blocks per CU

template <bool SCALE>
__global__ __launch_bounds__(kTPB) void k_qtw_valu(double* __restrict__ V, int64_t ldv, int64_t len, int nrows, int j,
                                                  const double* __restrict__ r, const double* __restrict__ nrm2,
                                                  double* __restrict__ beta_slot, int64_t L, int G,
                                                  double* __restrict__ part) {
  extern __shared__ double2 sw[];
  __shared__ double red[kQtwR][kTPB / 64];
  const int64_t base = (int64_t)blockIdx.x * L;
  const int cnt2 = (int)((len - base < L ? len - base : L) >> 1);  // double2 count of this slice
  double2* vj = reinterpret_cast<double2*>(V + (int64_t)j * ldv + base);
  double self = 0.0;
  if (SCALE) {
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
    for (int t = threadIdx.x; t < cnt2; t += kTPB) {
      const double2 v = vj[t];
      sw[t] = v;
      self = fma(v.x, v.x, self);
      self = fma(v.y, v.y, self);
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i0 = 0; i0 < nrows; i0 += kQtwR) {
    const double2* row[kQtwR];
#pragma unroll
    for (int q = 0; q < kQtwR; ++q) {
      int i = i0 + q;
      if (i >= nrows) i = nrows - 1;                    // clamped duplicate, result discarded
      row[q] = reinterpret_cast<const double2*>(V + (int64_t)i * ldv + base);
    }
    double acc[kQtwR];
#pragma unroll
    for (int q = 0; q < kQtwR; ++q) acc[q] = 0.0;
#pragma unroll 2
    for (int t = threadIdx.x; t < cnt2; t += kTPB) {
      const double2 wv = sw[t];
#pragma unroll
      for (int q = 0; q < kQtwR; ++q) {
        const double2 v = row[q][t];
        acc[q] = fma(v.x, wv.x, acc[q]);
        acc[q] = fma(v.y, wv.y, acc[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < kQtwR; ++q) {
      if (i0 + q == j) acc[q] = self;  // the self term c_j = w.w from LDS-resident data (same values)
      const double s = wave_sum(acc[q]);
      if (lane == 0) red[q][w] = s;
    }
    __syncthreads();
    if (threadIdx.x < kQtwR && i0 + threadIdx.x < nrows) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < kTPB / 64; ++k) s += red[threadIdx.x][k];
      part[(int64_t)(i0 + threadIdx.x) * G + blockIdx.x] = s;
    }
    __syncthreads();
  }
}

QtwPlan plan_qtw(int64_t len) {
  QtwPlan p;
  int64_t L = round_up((len + 2047) / 2048, 512);
  if (L < 512) L = 512;
  if (L > kQtwMaxL) L = kQtwMaxL;
  p.L = L;
  p.G = (int)((len + L - 1) / L);
  return p;
}

void launch_qtw(double* V, int64_t ldv, int64_t len, int nrows, int j, const double* r, const double* nrm2,
                double* beta_slot, const QtwPlan& plan, double* part, int flags, hipStream_t s) {
  const size_t lds = (size_t)plan.L * sizeof(double);
  if (r)
    hipLaunchKernelGGL(k_qtw_valu<true>, dim3(plan.G), dim3(kTPB), lds, s, V, ldv, len, nrows, j, r, nrm2, beta_slot, plan.L,
                       plan.G, part);
  else
    hipLaunchKernelGGL(k_qtw_valu<false>, dim3(plan.G), dim3(kTPB), lds, s, V, ldv, len, nrows, j, r, nrm2, beta_slot, plan.L,
                       plan.G, part);
}

// ------------------------------------------------------------------ re-orthogonalisation pass 2
// V[j] = 2 V[j] - (((c0 V0 + c1 V1) + c2 V2) + ...): NumPy's axis-0 reduction
// order of np.sum(c[:, None] * V, axis=0) (Lanczos.py:249), products and sums
// rounded separately.  One double2 column position per lane; the row loop is
// unrolled so 8 independent 16-byte loads are in flight per lane.
__global__ __launch_bounds__(kTPB) void k_update(double* __restrict__ V, int64_t ldv, int64_t n2, int nrows, int j,
                                                const double* __restrict__ c) {
  const int64_t i = (int64_t)blockIdx.x * kTPB + threadIdx.x;
  if (i >= n2) return;
  const double2* col = reinterpret_cast<const double2*>(V) + i;
  const int64_t ld2 = ldv >> 1;
  double tx = 0.0, ty = 0.0;
  int k = 0;
  for (; k + 8 <= nrows; k += 8) {
    double2 q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) q[u] = col[(int64_t)(k + u) * ld2];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const double ck = c[k + u];
      tx = tx + ck * q[u].x;
      ty = ty + ck * q[u].y;
    }
  }
  for (; k < nrows; ++k) {
    const double2 q = col[(int64_t)k * ld2];
    const double ck = c[k];
    tx = tx + ck * q.x;
    ty = ty + ck * q.y;
  }
  double2* out = reinterpret_cast<double2*>(V) + (int64_t)j * ld2 + i;
  const double2 v = *out;
  *out = make_double2(2.0 * v.x - tx, 2.0 * v.y - ty);
}

void launch_update(double* V, int64_t ldv, int64_t len, int nrows, int j, const double* c, hipStream_t s) {
  const int64_t n2 = len >> 1;
  const int grid = (int)((n2 + kTPB - 1) / kTPB);
  hipLaunchKernelGGL(k_update, dim3(grid), dim3(kTPB), 0, s, V, ldv, n2, nrows, j, c);
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
    const double2 v = v2[i];
    x.x = x.x - v.x * a;
    x.y = x.y - v.y * a;
    if (vjm1) {
      const double2 m = m2[i];
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
typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr int kRitzMT = 2;  // 16-row m tiles per wave
constexpr int kRitzNT = 4;  // 16-col i tiles per wave

__global__ __launch_bounds__(kTPB) void k_ritz_gemm(const double* __restrict__ V, int64_t ldv, int64_t rows, int n,
                                                   const double* __restrict__ S, int npad, double* __restrict__ Y,
                                                   int64_t ldy) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  const int64_t m0 = ((int64_t)blockIdx.x * (kTPB / 64) + w) * (16 * kRitzMT);
  const int ct0 = blockIdx.y * kRitzNT;
  const int CT = npad / 16;
  if (m0 >= rows) return;  // wave-uniform
  double4_t acc[kRitzMT][kRitzNT];
#pragma unroll
  for (int a = 0; a < kRitzMT; ++a)
#pragma unroll
    for (int b = 0; b < kRitzNT; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};
  const int kpad = (n + 3) & ~3;
  for (int k0 = 0; k0 < kpad; k0 += 4) {
    int kr = k0 + lk;
    const int krv = kr < n ? kr : n - 1;  // clamp: the matching S row is zero
    double av[kRitzMT], bv[kRitzNT];
#pragma unroll
    for (int a = 0; a < kRitzMT; ++a) {
      int64_t m = m0 + 16 * a + lr;
      if (m >= rows) m = rows - 1;  // clamped duplicate, result discarded at the store
      av[a] = V[(int64_t)krv * ldv + m];
    }
#pragma unroll
    for (int b = 0; b < kRitzNT; ++b) {
      const int ct = ct0 + b < CT ? ct0 + b : CT - 1;
      bv[b] = S[(int64_t)kr * npad + 16 * ct + lr];
    }
#pragma unroll
    for (int a = 0; a < kRitzMT; ++a)
#pragma unroll
      for (int b = 0; b < kRitzNT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
  }
#pragma unroll
  for (int a = 0; a < kRitzMT; ++a)
#pragma unroll
    for (int b = 0; b < kRitzNT; ++b) {
      if (ct0 + b >= CT) continue;
      const int col = 16 * (ct0 + b) + lr;
      if (col >= n) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int64_t m = m0 + 16 * a + lk + 4 * g;
        if (m < rows) Y[m * ldy + col] = acc[a][b][g];
      }
    }
}

void launch_ritz_gemm(const double* V, int64_t ldv, int64_t rows, int n, const double* Spad, int npad, double* Y,
                      int64_t ldy, hipStream_t s) {
  const int64_t mt = 16 * kRitzMT * (kTPB / 64);  // 128 rows per block
  dim3 grid((unsigned)((rows + mt - 1) / mt), (unsigned)((npad / 16 + kRitzNT - 1) / kRitzNT));
  hipLaunchKernelGGL(k_ritz_gemm, grid, dim3(kTPB), 0, s, V, ldv, rows, n, Spad, npad, Y, ldy);
}

}  // namespace lz
