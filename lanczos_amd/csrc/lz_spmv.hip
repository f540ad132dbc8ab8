// Hand-written gfx950 SpMV kernels of the Lanczos hot path: r = A v_j with the alpha partials fused.
//
// All kernels here are HBM-bandwidth bound (<= 0.25 flop/byte), so the design
// rules are: 16-byte coalesced accesses, many independent loads in flight per
// lane, non-temporal loads for data that is streamed once, deterministic
// two-stage reductions (wave shuffle -> LDS -> per-block partial -> tiny
// second-stage kernel), and no atomics.
//
// Arithmetic contract (DESIGN.md "numerics"): element-wise results follow the
// reference CPU branch's NumPy expression order with NO fused multiply-add
// (compiled with -ffp-contract=off), so SpMV row sums, the re-orthogonalisation
// update and the three-term recurrence are bit-identical to NumPy/SciPy given
// the same scalar inputs; only the inner products differ (summation order).
#include "lz_device.h"

namespace lz {

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

template <int FIXED_K>
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
        a[i] = ld_stream<1>(reinterpret_cast<const double2*>(vals + k));
        c[i] = ld_stream<1>(reinterpret_cast<const int2*>(colidx + k));
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
        int k = a;
        for (; k + 4 <= b; k += 4) {  // four LDS reads in flight; the adds stay in CSR order
          const double p0 = prod[k], p1 = prod[k + 1], p2 = prod[k + 2], p3 = prod[k + 3];
          sum += p0;
          sum += p1;
          sum += p2;
          sum += p3;
        }
        for (; k < b; ++k) sum += prod[k];
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

// (d) ELL-ordered fixed-K rows (round 4; the default for 5-, 7- and 27-point stencils): the matrix is kept a second time
// in blocks of RB rows with entry k of every row of a block contiguous, so a LANE owns whole rows: its K value loads and
// K column loads are coalesced 8-byte (VEC == 1) or 16-byte (VEC == 2: two adjacent rows per lane) accesses, a stencil's
// k-th gathers of a wave are one contiguous run of x, and the row sum is formed in registers in CSR order (sum += a * x,
// unfused: SciPy's bits) - no LDS staging of the products, no barrier before the alpha reduction.  All of a lane's loads
// are issued before the first gather, all gathers before the first use.  Blocks are padded with (value 0, column 0).
//   SC (the device-resident partial re-orthogonalisation loop, lz_loops.hip): x = r / beta is formed per gathered entry
//   when no sweep ran on this vector (see SpmvScale), and the block stores the rows it owns to V[j].
template <typename T>
__device__ __forceinline__ T ld_nt(const T* p) {
  return __builtin_nontemporal_load(p);
}

//   CODED (round 5; "row-class coding"): a stencil matrix has a handful of distinct rows up to translation - the K offsets col - row
//   take at most 256 different tuples ("classes": interior, faces, edges, corners of a periodic grid; ell_build finds them on the device
//   from the CSR arrays and verifies every row against its class).  CODED == 1: the columns are one BYTE per row (the class; its offsets
//   sit in LDS) instead of 4 bytes per entry, the values stream as before.  Where the values repeat with the offsets (constant
//   coefficients) the whole matrix is that byte per row: k_spmv_cls below.  Products, their order and the alpha partials are exactly
//   those of the uncoded kernel: same bits.
template <int K, int RPT, int VEC, bool SC, int CODED>
__global__ __launch_bounds__(kTPB) void k_spmv_ell(const int32_t* __restrict__ ec, const double* __restrict__ ev,
                                                  const double* __restrict__ x, const double* __restrict__ xown,
                                                  double* __restrict__ y, int rows, int rows_pad, double* __restrict__ part,
                                                  SpmvScale sc, EllCode code) {
  constexpr int RB = kTPB * RPT * VEC;
  constexpr int NR = RPT * VEC;  // rows per lane
  __shared__ double sm[kTPB / 64];
  extern __shared__ double s_tab[];  // CODED == 1: [ncls * K] offsets (ints)
  const int blk = xcd_remap(blockIdx.x, gridDim.x);
  const int r0 = blk * RB;
  const int64_t e0 = (int64_t)blk * K * RB;
  int lrow[NR];
#pragma unroll
  for (int q = 0; q < NR; ++q)
    lrow[q] = VEC == 1 ? r0 + q * kTPB + (int)threadIdx.x : r0 + (q >> 1) * (2 * kTPB) + 2 * (int)threadIdx.x + (q & 1);
  double a[NR][K];
  int c[NR][K];
  if constexpr (CODED != 0) {
    int cl[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q) cl[q] = lrow[q] < rows ? (int)code.cls[lrow[q]] : -1;
    {  // the values stream as in the uncoded kernel (issued before the table is staged)
#pragma unroll
      for (int q = 0; q < RPT; ++q)
#pragma unroll
        for (int k = 0; k < K; ++k) {
          if constexpr (VEC == 1) {
            a[q][k] = ld_nt(ev + e0 + (int64_t)k * RB + q * kTPB + threadIdx.x);
          } else {
            const double2 av = ld_stream<1>(reinterpret_cast<const double2*>(ev + e0 + (int64_t)k * RB + q * (2 * kTPB) + 2 * threadIdx.x));
            a[2 * q][k] = av.x;
            a[2 * q + 1][k] = av.y;
          }
        }
    }
    const int nt = code.ncls * K;
    int* s_off = reinterpret_cast<int*>(s_tab);
    for (int i = threadIdx.x; i < nt; i += kTPB) s_off[i] = code.off[i];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NR; ++q)
#pragma unroll
      for (int k = 0; k < K; ++k)  // (rows past the end of the last block: column 0; their values are the pad slots' zeros)
        c[q][k] = cl[q] >= 0 ? lrow[q] + s_off[cl[q] * K + k] : 0;
  } else {
#pragma unroll
    for (int q = 0; q < RPT; ++q)
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if constexpr (VEC == 1) {
          const int64_t idx = e0 + (int64_t)k * RB + q * kTPB + threadIdx.x;
          a[q][k] = ld_nt(ev + idx);
          c[q][k] = ld_nt(ec + idx);
        } else {
          const int64_t idx = e0 + (int64_t)k * RB + q * (2 * kTPB) + 2 * threadIdx.x;
          const double2 av = ld_stream<1>(reinterpret_cast<const double2*>(ev + idx));
          const int2 cv = ld_stream<1>(reinterpret_cast<const int2*>(ec + idx));
          a[2 * q][k] = av.x;
          a[2 * q + 1][k] = av.y;
          c[2 * q][k] = cv.x;
          c[2 * q + 1][k] = cv.y;
        }
      }
  }
  bool scale = false;
  double beta = 1.0;
  const double* xs = x;
  const double* xo = xown;
  if constexpr (SC) {
    scale = sc.gate[0] == 0;
    if (scale) {
      beta = sqrt(sc.nrm2[0]);
      xs = sc.r;
      xo = sc.r;
      if (blockIdx.x == 0 && threadIdx.x == 0) sc.beta_slot[0] = beta;
    }
  }
  double xv[NR][K];
#pragma unroll
  for (int q = 0; q < NR; ++q)
#pragma unroll
    for (int k = 0; k < K; ++k) xv[q][k] = xs[c[q][k]];
  double own[NR];
#pragma unroll
  for (int q = 0; q < NR; ++q) own[q] = lrow[q] < rows_pad ? xo[lrow[q]] : 0.0;
  if constexpr (SC) {
    if (scale) {
#pragma unroll
      for (int q = 0; q < NR; ++q) {
#pragma unroll
        for (int k = 0; k < K; ++k) xv[q][k] = xv[q][k] / beta;
        own[q] = own[q] / beta;
        if (lrow[q] < rows_pad) sc.vj[lrow[q]] = own[q];  // (the pad of r is zero: 0 / beta keeps the pad of V[j] zero)
      }
    }
  }
  double d = 0.0;
#pragma unroll
  for (int q = 0; q < NR; ++q) {
    double sum = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) sum += a[q][k] * xv[q][k];
    if (lrow[q] < rows) {
      y[lrow[q]] = sum;
      d += own[q] * sum;
    }
  }
  d = block_sum(d, sm);
  if (threadIdx.x == 0) part[blk] = d;
}

// Offsets AND values by class: the matrix is one byte per row.  A block covers G partial units of RBU = 256 RPT rows - the alpha partial of
// a unit is formed exactly as k_spmv_ell forms a block's (lane t: rows t, t + 256, ... of the unit in order; wave sums; the waves added in
// order), so the bits do not depend on G - and a lane keeps G RPT rows in flight: the values are read from LDS only when the gathers
// have landed, which leaves registers for twice the rows of the uncoded kernel (the kernel is bound by two dependent memory round
// trips - class byte, then x - not by bytes: 17 per row).
template <int K, int RPT, int G, bool SC, bool DIAG>
__global__ __launch_bounds__(kTPB) void k_spmv_cls(EllCode code, const double* __restrict__ x, const double* __restrict__ xown, double* __restrict__ y,
                                                  int rows, int rows_pad, int nunits, double* __restrict__ part, SpmvScale sc) {
  constexpr int RBU = kTPB * RPT;
  constexpr int NR = RPT * G;
  __shared__ double sm[G][kTPB / 64];
  extern __shared__ double s_tab[];  // [ncls * K] values, then [ncls * K] offsets
  const int blk = xcd_remap(blockIdx.x, gridDim.x);
  const int r0 = blk * (RBU * G);
  int lrow[NR], cl[NR];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int q = 0; q < RPT; ++q) lrow[g * RPT + q] = r0 + g * RBU + q * kTPB + (int)threadIdx.x;
#pragma unroll
  for (int i = 0; i < NR; ++i) cl[i] = lrow[i] < rows ? (int)code.cls[lrow[i]] : -1;
  double dg[NR];  // DIAG (coding 3): the diagonal's value streams, the class holds the rest
  if constexpr (DIAG) {
#pragma unroll
    for (int i = 0; i < NR; ++i) dg[i] = lrow[i] < rows ? code.diag[lrow[i]] : 0.0;
  }
  bool scale = false;
  double beta = 1.0;
  const double* xs = x;
  const double* xo = xown;
  if constexpr (SC) {
    scale = sc.gate[0] == 0;
    if (scale) {
      beta = sqrt(sc.nrm2[0]);
      xs = sc.r;
      xo = sc.r;
      if (blockIdx.x == 0 && threadIdx.x == 0) sc.beta_slot[0] = beta;
    }
  }
  double own[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) own[i] = lrow[i] < rows_pad ? xo[lrow[i]] : 0.0;  // (does not wait for the class)
  const int nt = code.ncls * K;
  int* s_off = reinterpret_cast<int*>(s_tab + nt);
  for (int i = threadIdx.x; i < nt; i += kTPB) {
    s_off[i] = code.off[i];
    s_tab[i] = code.val[i];
  }
  __syncthreads();
  double xv[NR][K];
#pragma unroll
  for (int i = 0; i < NR; ++i)
#pragma unroll
    for (int k = 0; k < K; ++k) xv[i][k] = xs[cl[i] >= 0 ? lrow[i] + s_off[cl[i] * K + k] : 0];
  if constexpr (SC) {
    if (scale) {
#pragma unroll
      for (int i = 0; i < NR; ++i) {
#pragma unroll
        for (int k = 0; k < K; ++k) xv[i][k] = xv[i][k] / beta;
        own[i] = own[i] / beta;
        if (lrow[i] < rows_pad) sc.vj[lrow[i]] = own[i];  // (the pad of r is zero: 0 / beta keeps the pad of V[j] zero)
      }
    }
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    double d = 0.0;
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      const int i = g * RPT + q;
      if (lrow[i] < rows) {
        double sum = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          double a = s_tab[cl[i] * K + k];
          if constexpr (DIAG) a = s_off[cl[i] * K + k] == 0 ? dg[i] : a;
          sum += a * xv[i][k];
        }
        y[lrow[i]] = sum;
        d += own[i] * sum;
      }
    }
    d = wave_sum(d);
    if (lane == 0) sm[g][w] = d;
  }
  __syncthreads();
  if (threadIdx.x < G) {
    const int unit = blk * G + (int)threadIdx.x;
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < kTPB / 64; ++i) t += sm[threadIdx.x][i];
    if (unit < nunits) part[unit] = t;
  }
}

// The same with TWO ADJACENT ROWS PER LANE (the default for 5 and 7 entries per row).  rocprofv3 counters of k_spmv_cls on the headline
// matrix (profiles/r05/pmc_spmv_summary.json): waves 43 % parked on memory, 43 % stalled at ISSUE, 14 % active; 5000 vector-memory
// instructions per CU x ~16 clocks each = the kernel's 43 us - the texture-address unit takes a 64-lane 8-byte gather at 4 lanes per
// clock, and a row costs eight such instructions (K gathers, class byte, own x, y).  Rows 2t and 2t + 1 of one class gather
// x[2t + off], x[2t + 1 + off] as ONE 16-byte load per lane (8-byte aligned: fine for global loads), own x and y are 16-byte accesses,
// the two class bytes one 2-byte load: four instructions per row.  Lanes whose two rows differ in class (a grid line's end) take two
// 8-byte gathers.  The alpha partial must still be the one k_spmv_ell forms (lane L: rows L and L + 256 of the unit, in order): the
// products own * sum go through LDS to that lane mapping, then the same wave sums.  Same bits.
template <int K, bool SC, bool DIAG>
__global__ __launch_bounds__(kTPB) void k_spmv_cls2(EllCode code, const double* __restrict__ x, const double* __restrict__ xown, double* __restrict__ y,
                                                   int rows, int rows_pad, double* __restrict__ part, SpmvScale sc) {
  constexpr int RBU = 2 * kTPB;  // one alpha partial unit per workgroup
  __shared__ double sm[kTPB / 64];
  __shared__ __attribute__((aligned(16))) double sp[RBU];
  extern __shared__ double s_tab[];  // [ncls * K] values, then [ncls * K] offsets
  struct __attribute__((aligned(8))) Pair {  // two adjacent doubles at an 8-byte aligned address: one 16-byte load
    double x, y;
  };
  const int blk = xcd_remap(blockIdx.x, gridDim.x);
  const int ra = blk * RBU + 2 * (int)threadIdx.x;  // rows ra, ra + 1 (the class array is padded to whole units)
  const unsigned short cpair = *reinterpret_cast<const unsigned short*>(code.cls + ra);
  const bool live_a = ra < rows, live_b = ra + 1 < rows;
  const int ca = cpair & 0xFF, cb = cpair >> 8;
  double2 dg = make_double2(0.0, 0.0);  // DIAG (coding 3): the diagonal's values stream (the array is padded to whole units)
  if constexpr (DIAG) dg = *reinterpret_cast<const double2*>(code.diag + ra);
  bool scale = false;
  double beta = 1.0;
  const double* xs = x;
  const double* xo = xown;
  if constexpr (SC) {
    scale = sc.gate[0] == 0;
    if (scale) {
      beta = sqrt(sc.nrm2[0]);
      xs = sc.r;
      xo = sc.r;
      if (blockIdx.x == 0 && threadIdx.x == 0) sc.beta_slot[0] = beta;
    }
  }
  double2 own = make_double2(0.0, 0.0);
  if (ra + 1 < rows_pad) own = *reinterpret_cast<const double2*>(xo + ra);  // (rows_pad is even: both or neither)
  const int nt = code.ncls * K;
  int* s_off = reinterpret_cast<int*>(s_tab + nt);
  for (int i = threadIdx.x; i < nt; i += kTPB) {
    s_off[i] = code.off[i];
    s_tab[i] = code.val[i];
  }
  __syncthreads();
  double xa[K], xb[K];
  if (live_b && ca == cb) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const Pair v = *reinterpret_cast<const Pair*>(xs + (ra + s_off[ca * K + k]));
      xa[k] = v.x;
      xb[k] = v.y;
    }
  } else {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      xa[k] = xs[live_a ? ra + s_off[ca * K + k] : 0];
      xb[k] = xs[live_b ? ra + 1 + s_off[cb * K + k] : 0];
    }
  }
  if constexpr (SC) {
    if (scale) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        xa[k] = xa[k] / beta;
        xb[k] = xb[k] / beta;
      }
      own.x = own.x / beta;
      own.y = own.y / beta;
      if (ra + 1 < rows_pad) *reinterpret_cast<double2*>(sc.vj + ra) = own;  // (the pad of r is zero: 0 / beta keeps the pad of V[j] zero)
    }
  }
  double suma = 0.0, sumb = 0.0;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double a = s_tab[ca * K + k];
    if constexpr (DIAG) a = s_off[ca * K + k] == 0 ? dg.x : a;
    suma += a * xa[k];
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double a = s_tab[cb * K + k];
    if constexpr (DIAG) a = s_off[cb * K + k] == 0 ? dg.y : a;
    sumb += a * xb[k];
  }
  if (live_b)
    *reinterpret_cast<double2*>(y + ra) = make_double2(suma, sumb);
  else if (live_a)
    y[ra] = suma;
  // own * sum of every row, to the lane mapping of the one-row-per-lane kernels (lane L adds rows L, then L + 256, of the unit)
  *reinterpret_cast<double2*>(sp + 2 * threadIdx.x) = make_double2(live_a ? own.x * suma : 0.0, live_b ? own.y * sumb : 0.0);
  __syncthreads();
  double d = 0.0;
  d += sp[threadIdx.x];
  d += sp[threadIdx.x + kTPB];
  d = block_sum(d, sm);
  if (threadIdx.x == 0) part[blk] = d;
}

// ---- row classes of a fixed-K matrix, found on the device ------------------------------------------------------------------
// Two open-addressing tables of 1024 slots (keys: a 64-bit hash of the row's K offsets / of its offsets and value bits; 0 = empty).
// ctl: [0] classes by offsets, [1] classes by offsets + values, [2] / [3] "more than 256" of either, [4] a row that does not equal
// its class's representative (a hash collision: the coding is then abandoned).
constexpr int kClsSlots = 1024, kClsMax = 256;

__device__ __forceinline__ unsigned long long cls_mix(unsigned long long h, unsigned long long v) {
  h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
  h *= 0xBF58476D1CE4E5B9ull;
  h ^= h >> 31;
  return h;
}
__device__ __forceinline__ void cls_hash(const int32_t* __restrict__ colidx, const double* __restrict__ vals, int K, int row, unsigned long long* h_off,
                                         unsigned long long* h_all, unsigned long long* h_nd) {
  unsigned long long h1 = 0x243F6A8885A308D3ull, h2 = 0x13198A2E03707344ull, h3 = 0xA4093822299F31D0ull;
  const int64_t e = (int64_t)row * K;
  for (int k = 0; k < K; ++k) {
    const int off = colidx[e + k] - row;
    const unsigned long long o = (unsigned long long)(long long)off;
    const unsigned long long v = (unsigned long long)__double_as_longlong(vals[e + k]);
    h1 = cls_mix(h1, o);
    h2 = cls_mix(cls_mix(h2, o), v);
    h3 = cls_mix(cls_mix(h3, o), off == 0 ? 0x5851F42D4C957F2Dull : v);  // (the diagonal's value is not part of this key)
  }
  *h_off = h1 ? h1 : 1;
  *h_all = h2 ? h2 : 1;
  *h_nd = h3 ? h3 : 1;
}
__device__ __forceinline__ void cls_insert(unsigned long long* keys, int* rep, int* count, int* overflow, unsigned long long h, int row) {
  if (*reinterpret_cast<volatile int*>(overflow)) return;
  unsigned slot = (unsigned)(h >> 20) & (kClsSlots - 1);
  for (int p = 0; p < kClsSlots; ++p, slot = (slot + 1) & (kClsSlots - 1)) {
    unsigned long long prev = *reinterpret_cast<volatile unsigned long long*>(keys + slot);  // (a stale 0 only costs the atomic below)
    if (prev == 0) prev = atomicCAS(keys + slot, 0ull, h);
    if (prev == 0) {
      if (atomicAdd(count, 1) + 1 > kClsMax) atomicExch(overflow, 1);
      atomicMin(rep + slot, row);
      return;
    }
    if (prev == h) {
      if (row < *reinterpret_cast<volatile int*>(rep + slot)) atomicMin(rep + slot, row);  // (the representative: the class's first row)
      return;
    }
  }
  atomicExch(overflow, 1);
}
__global__ __launch_bounds__(kTPB) void k_cls_insert(const int32_t* __restrict__ colidx, const double* __restrict__ vals, int K, int rows,
                                                    unsigned long long* keys_off, int* rep_off, unsigned long long* keys_all, int* rep_all,
                                                    unsigned long long* keys_nd, int* rep_nd, int* ctl) {
  const int row = blockIdx.x * kTPB + threadIdx.x;
  if (row >= rows) return;
  unsigned long long h1, h2, h3;
  cls_hash(colidx, vals, K, row, &h1, &h2, &h3);
  cls_insert(keys_off, rep_off, ctl + 0, ctl + 2, h1, row);
  cls_insert(keys_all, rep_all, ctl + 1, ctl + 3, h2, row);
  cls_insert(keys_nd, rep_nd, ctl + 5, ctl + 6, h3, row);
}
// one block of kClsSlots threads: number the occupied slots in slot order and write the class table from the representatives
__global__ __launch_bounds__(kClsSlots) void k_cls_number(const int32_t* __restrict__ colidx, const double* __restrict__ vals, int K,
                                                         const unsigned long long* __restrict__ keys, const int* __restrict__ rep, int* __restrict__ ids,
                                                         int32_t* __restrict__ toff, double* __restrict__ tval) {
  __shared__ int scan[kClsSlots];
  const int t = threadIdx.x;
  const int occ = keys[t] != 0 ? 1 : 0;
  scan[t] = occ;
  __syncthreads();
  for (int d = 1; d < kClsSlots; d <<= 1) {
    const int v = t >= d ? scan[t - d] : 0;
    __syncthreads();
    scan[t] += v;
    __syncthreads();
  }
  const int id = scan[t] - occ;
  ids[t] = occ ? id : -1;
  if (occ && id < kClsMax) {
    const int r = rep[t];
    for (int k = 0; k < K; ++k) {
      toff[id * K + k] = colidx[(int64_t)r * K + k] - r;
      tval[id * K + k] = vals[(int64_t)r * K + k];
    }
  }
}
// with_vals: 0 the class is the offsets; 1 offsets and values; 2 offsets and the values off the diagonal (the diagonal's value goes to diag[row])
__global__ __launch_bounds__(kTPB) void k_cls_assign(const int32_t* __restrict__ colidx, const double* __restrict__ vals, int K, int rows, int with_vals,
                                                    const unsigned long long* __restrict__ keys, const int* __restrict__ ids,
                                                    const int32_t* __restrict__ toff, const double* __restrict__ tval, uint8_t* __restrict__ cls,
                                                    double* __restrict__ diag, int* ctl) {
  const int row = blockIdx.x * kTPB + threadIdx.x;
  if (row >= rows) return;
  unsigned long long h1, h2, h3;
  cls_hash(colidx, vals, K, row, &h1, &h2, &h3);
  const unsigned long long h = with_vals == 1 ? h2 : (with_vals == 2 ? h3 : h1);
  unsigned slot = (unsigned)(h >> 20) & (kClsSlots - 1);
  int id = -1;
  for (int p = 0; p < kClsSlots; ++p, slot = (slot + 1) & (kClsSlots - 1))
    if (keys[slot] == h) {
      id = ids[slot];
      break;
    }
  bool same = id >= 0 && id < kClsMax;
  double dg = 0.0;
  if (same) {
    const int64_t e = (int64_t)row * K;
    for (int k = 0; k < K; ++k) {
      const int off = colidx[e + k] - row;
      same = same && toff[id * K + k] == off;
      if (with_vals == 2 && off == 0)
        dg = vals[e + k];  // (a row with two entries on the diagonal keeps the last here and would sum both below: such a matrix hashes them
                           // into the class key as "diagonal" twice and the kernel would use one value for both - refused:)
      else if (with_vals) same = same && __double_as_longlong(tval[id * K + k]) == __double_as_longlong(vals[e + k]);
    }
    if (with_vals == 2) {
      int ndiag = 0;
      for (int k = 0; k < K; ++k) ndiag += colidx[e + k] == row;
      same = same && ndiag <= 1;
    }
  }
  if (!same) {
    atomicExch(ctl + 4, 1);
    return;
  }
  cls[row] = (uint8_t)id;
  if (with_vals == 2) diag[row] = dg;
}

__global__ __launch_bounds__(kTPB) void k_ell_build(const int32_t* __restrict__ colidx, const double* __restrict__ vals, int K, int RB,
                                                   int64_t nnz, int32_t* __restrict__ ec, double* __restrict__ ev) {
  const int64_t e = (int64_t)blockIdx.x * kTPB + threadIdx.x;
  if (e >= nnz) return;
  const int64_t row = e / K;
  const int k = (int)(e - row * K);
  const int64_t b = row / RB;
  const int lr = (int)(row - b * RB);
  const int64_t dst = (b * K + k) * RB + lr;
  if (ec) ec[dst] = colidx[e];  // (offsets-only row-class coding keeps the values, not the columns)
  ev[dst] = vals[e];
}

static int ell_rows_per_block(int K, int variant) { return K > 7 ? kTPB : 2 * kTPB; }

void ell_free(CsrDev& A) {
  if (A.ell_c) hipFree(A.ell_c);
  if (A.ell_v) hipFree(A.ell_v);
  if (A.ell_cls) hipFree(A.ell_cls);
  if (A.cls_off) hipFree(A.cls_off);
  if (A.cls_val) hipFree(A.cls_val);
  if (A.cls_diag) hipFree(A.cls_diag);
  A.cls_diag = nullptr;
  A.ell_c = nullptr;
  A.ell_v = nullptr;
  A.ell_cls = nullptr;
  A.cls_off = nullptr;
  A.cls_val = nullptr;
  A.ell_rb = 0;
  A.ell_coded = 0;
  A.ell_ncls = 0;
}

// Row classes of the fixed-K matrix A (see k_spmv_ell, CODED): returns the coding found - 2 offsets and values, 3 offsets and the values
// off the diagonal (the diagonal streams: a constant-coefficient stencil plus a potential - the reference's own operators), 1 offsets
// only, 0 none (more than 256 classes of every kind) - with A.ell_cls / cls_off / cls_val / cls_diag / ell_ncls filled in.  `want`: 1 try
// all, 2 offsets only (A/B arm).  Two short synchronisations of `s` (this is matrix set-up).
static hipError_t ell_classes(CsrDev& A, int want, hipStream_t s, int* coded_out) {
  *coded_out = 0;
  const int K = A.fixed_k;
  const int rows = (int)A.rows;
  struct Scratch {
    void* p = nullptr;
    ~Scratch() {
      if (p) hipFree(p);
    }
  } scr;
  // [keys_off | keys_all | keys_nd: 1024 u64 each | rep_off | rep_all | rep_nd | ids: 1024 ints each | ctl 8]
  const size_t bytes = 3 * kClsSlots * sizeof(unsigned long long) + (4 * kClsSlots + 8) * sizeof(int);
  hipError_t e = hipMalloc(&scr.p, bytes);
  if (e != hipSuccess) return e;
  unsigned long long* keys_off = static_cast<unsigned long long*>(scr.p);
  unsigned long long* keys_all = keys_off + kClsSlots;
  unsigned long long* keys_nd = keys_all + kClsSlots;
  int* rep_off = reinterpret_cast<int*>(keys_nd + kClsSlots);
  int* rep_all = rep_off + kClsSlots;
  int* rep_nd = rep_all + kClsSlots;
  int* ids = rep_nd + kClsSlots;
  int* ctl = ids + kClsSlots;  // [0] / [1] / [5] classes by offsets / all values / off-diagonal values; [2] / [3] / [6] their overflows; [4] mismatch
  if ((e = hipMemsetAsync(scr.p, 0, bytes, s)) != hipSuccess) return e;
  if ((e = hipMemsetAsync(rep_off, 0x7F, 3 * kClsSlots * sizeof(int), s)) != hipSuccess) return e;
  const unsigned grid = (unsigned)((rows + kTPB - 1) / kTPB);
  hipLaunchKernelGGL(k_cls_insert, dim3(grid), dim3(kTPB), 0, s, A.colidx, A.vals, K, rows, keys_off, rep_off, keys_all, rep_all, keys_nd, rep_nd, ctl);
  int h_ctl[8] = {0};
  if ((e = hipMemcpyAsync(h_ctl, ctl, sizeof h_ctl, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
  if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
  int coded = 0;
  if (want == 1 && !h_ctl[3] && h_ctl[1] <= kClsMax) coded = 2;
  else if (want == 1 && !h_ctl[6] && h_ctl[5] <= kClsMax) coded = 3;
  else if (!h_ctl[2] && h_ctl[0] <= kClsMax) coded = 1;
  if (!coded) return hipSuccess;
  const int ncls = coded == 2 ? h_ctl[1] : (coded == 3 ? h_ctl[5] : h_ctl[0]);
  const unsigned long long* keys = coded == 2 ? keys_all : (coded == 3 ? keys_nd : keys_off);
  const int* rep = coded == 2 ? rep_all : (coded == 3 ? rep_nd : rep_off);
  void* p = nullptr;
  const int RB = A.ell_rb;
  const int64_t nblk = (A.rows + RB - 1) / RB;
  if ((e = hipMalloc(&p, (size_t)nblk * RB)) != hipSuccess) return e;
  A.ell_cls = static_cast<uint8_t*>(p);
  if ((e = hipMalloc(&p, (size_t)kClsMax * K * sizeof(int32_t))) != hipSuccess) return e;
  A.cls_off = static_cast<int32_t*>(p);
  if ((e = hipMalloc(&p, (size_t)kClsMax * K * sizeof(double))) != hipSuccess) return e;
  A.cls_val = static_cast<double*>(p);
  if ((e = hipMemsetAsync(A.ell_cls, 0, (size_t)nblk * RB, s)) != hipSuccess) return e;
  if (coded == 3) {
    if ((e = hipMalloc(&p, (size_t)nblk * RB * sizeof(double))) != hipSuccess) return e;
    A.cls_diag = static_cast<double*>(p);
    if ((e = hipMemsetAsync(A.cls_diag, 0, (size_t)nblk * RB * sizeof(double), s)) != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k_cls_number, dim3(1), dim3(kClsSlots), 0, s, A.colidx, A.vals, K, keys, rep, ids, A.cls_off, A.cls_val);
  hipLaunchKernelGGL(k_cls_assign, dim3(grid), dim3(kTPB), 0, s, A.colidx, A.vals, K, rows, coded == 2 ? 1 : (coded == 3 ? 2 : 0), keys, ids, A.cls_off,
                     A.cls_val, A.ell_cls, A.cls_diag, ctl);
  if ((e = hipMemcpyAsync(h_ctl, ctl, sizeof h_ctl, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
  if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
  if ((e = hipGetLastError()) != hipSuccess) return e;
  if (h_ctl[4]) return hipSuccess;  // a hash collision (two different rows, one key), or two entries on one diagonal: no coding
  A.ell_ncls = ncls;
  *coded_out = coded;
  return hipSuccess;
}

// coding: 0 the plain ELL copy, 1 row-class coding where the matrix allows it (offsets and values, else offsets only, else plain),
// 2 at most the offsets-only coding (A/B arm); plain_fallback == false: no copy at all when the rows do not fall into classes
hipError_t ell_build(CsrDev& A, int variant, hipStream_t s, int coding, bool plain_fallback) {
  ell_free(A);
  const int K = A.fixed_k;
  if (!(K == 5 || K == 7 || K == 27) || A.rows <= 0) return hipSuccess;
  if (K == 27) variant = 0;
  const int RB = ell_rows_per_block(K, variant);
  const int64_t nblk = (A.rows + RB - 1) / RB;
  const size_t cap = (size_t)nblk * K * RB;
  A.ell_rb = RB;  // (ell_classes sizes the class bytes by it)
  hipError_t e = hipSuccess;
  int coded = 0;
  if (coding && A.rows < ((int64_t)1 << 31) - 2 * RB) {
    e = ell_classes(A, coding, s, &coded);
    if (e != hipSuccess || !coded) {
      const int rb = A.ell_rb;
      ell_free(A);
      A.ell_rb = rb;
      if (e != hipSuccess) {
        A.ell_rb = 0;
        return e;
      }
    }
  }
  if (coded == 0 && coding && !plain_fallback) {
    ell_free(A);
    return hipSuccess;
  }
  void* p = nullptr;
  if (coded == 0) {
    e = hipMalloc(&p, cap * sizeof(int32_t));
    if (e != hipSuccess) {
      ell_free(A);
      return e;
    }
    A.ell_c = static_cast<int32_t*>(p);
    e = hipMemsetAsync(A.ell_c, 0, cap * sizeof(int32_t), s);
  }
  if (e == hipSuccess && coded != 2 && coded != 3) {
    e = hipMalloc(&p, cap * sizeof(double));
    if (e == hipSuccess) {
      A.ell_v = static_cast<double*>(p);
      // only the last block has pad slots, but a memset of the whole copy is cheaper than finding them
      e = hipMemsetAsync(A.ell_v, 0, cap * sizeof(double), s);
    }
  }
  if (e != hipSuccess) {
    ell_free(A);
    return e;
  }
  if (coded != 2 && coded != 3) {
    const int64_t nnz = A.rows * K;
    hipLaunchKernelGGL(k_ell_build, dim3((unsigned)((nnz + kTPB - 1) / kTPB)), dim3(kTPB), 0, s, A.colidx, A.vals, K, RB, nnz, A.ell_c, A.ell_v);
    e = hipGetLastError();
    if (e != hipSuccess) {
      ell_free(A);
      return e;
    }
  }
  A.ell_rb = RB;
  A.ell_variant = variant;
  A.ell_coded = coded;
  return hipSuccess;
}

bool ell_usable(const CsrDev& A, int flags) {
  return A.ell_rb > 0 && !(flags & (LZ_FLAG_SPMV_SCALAR | LZ_FLAG_SPMV_STREAM)) && !A.pb;
}

template <int K, int RPT, int VEC, int CODED>
static int launch_spmv_ell_c(const CsrDev& A, const double* x, double* y, const double* x_own, double* part, hipStream_t s, const SpmvScale* sc) {
  constexpr int RB = kTPB * RPT * VEC;
  const int grid = (int)((A.rows + RB - 1) / RB);
  const int rows_pad = (int)round_up(A.rows, kPadDoubles);
  EllCode code;
  code.cls = A.ell_cls;
  code.off = A.cls_off;
  code.val = A.cls_val;
  code.ncls = A.ell_ncls;
  const size_t lds = CODED == 0 ? 0 : (size_t)A.ell_ncls * K * 4 + 8;
  if (sc)
    hipLaunchKernelGGL((k_spmv_ell<K, RPT, VEC, true, CODED>), dim3(grid), dim3(kTPB), lds, s, A.ell_c, A.ell_v, x, x_own, y, (int)A.rows, rows_pad, part, *sc,
                       code);
  else
    hipLaunchKernelGGL((k_spmv_ell<K, RPT, VEC, false, CODED>), dim3(grid), dim3(kTPB), lds, s, A.ell_c, A.ell_v, x, x_own, y, (int)A.rows, rows_pad, part,
                       SpmvScale(), code);
  return grid;
}
template <int K, int RPT, int G>
static int launch_spmv_cls(const CsrDev& A, const double* x, double* y, const double* x_own, double* part, hipStream_t s, const SpmvScale* sc) {
  constexpr int RBU = kTPB * RPT;
  const int nunits = (int)((A.rows + RBU - 1) / RBU);  // one alpha partial per unit: the ELL copy's blocks
  const int grid = (nunits + G - 1) / G;
  const int rows_pad = (int)round_up(A.rows, kPadDoubles);
  EllCode code;
  code.cls = A.ell_cls;
  code.off = A.cls_off;
  code.val = A.cls_val;
  code.ncls = A.ell_ncls;
  const size_t lds = (size_t)A.ell_ncls * K * 12 + 8;
  code.diag = A.cls_diag;
  const SpmvScale none;
  if (A.ell_coded == 3) {
    if (sc)
      hipLaunchKernelGGL((k_spmv_cls<K, RPT, G, true, true>), dim3(grid), dim3(kTPB), lds, s, code, x, x_own, y, (int)A.rows, rows_pad, nunits, part, *sc);
    else
      hipLaunchKernelGGL((k_spmv_cls<K, RPT, G, false, true>), dim3(grid), dim3(kTPB), lds, s, code, x, x_own, y, (int)A.rows, rows_pad, nunits, part, none);
  } else if (sc) {
    hipLaunchKernelGGL((k_spmv_cls<K, RPT, G, true, false>), dim3(grid), dim3(kTPB), lds, s, code, x, x_own, y, (int)A.rows, rows_pad, nunits, part, *sc);
  } else {
    hipLaunchKernelGGL((k_spmv_cls<K, RPT, G, false, false>), dim3(grid), dim3(kTPB), lds, s, code, x, x_own, y, (int)A.rows, rows_pad, nunits, part, none);
  }
  return nunits;
}
template <int K>
static int launch_spmv_cls2(const CsrDev& A, const double* x, double* y, const double* x_own, double* part, hipStream_t s, const SpmvScale* sc) {
  const int nunits = (int)((A.rows + 2 * kTPB - 1) / (2 * kTPB));
  const int rows_pad = (int)round_up(A.rows, kPadDoubles);
  EllCode code;
  code.cls = A.ell_cls;
  code.off = A.cls_off;
  code.val = A.cls_val;
  code.ncls = A.ell_ncls;
  const size_t lds = (size_t)A.ell_ncls * K * 12 + 8;
  code.diag = A.cls_diag;
  const SpmvScale none;
  if (A.ell_coded == 3) {
    if (sc)
      hipLaunchKernelGGL((k_spmv_cls2<K, true, true>), dim3(nunits), dim3(kTPB), lds, s, code, x, x_own, y, (int)A.rows, rows_pad, part, *sc);
    else
      hipLaunchKernelGGL((k_spmv_cls2<K, false, true>), dim3(nunits), dim3(kTPB), lds, s, code, x, x_own, y, (int)A.rows, rows_pad, part, none);
  } else if (sc) {
    hipLaunchKernelGGL((k_spmv_cls2<K, true, false>), dim3(nunits), dim3(kTPB), lds, s, code, x, x_own, y, (int)A.rows, rows_pad, part, *sc);
  } else {
    hipLaunchKernelGGL((k_spmv_cls2<K, false, false>), dim3(nunits), dim3(kTPB), lds, s, code, x, x_own, y, (int)A.rows, rows_pad, part, none);
  }
  return nunits;
}
template <int K, int RPT, int VEC>
static int launch_spmv_ell_t(const CsrDev& A, const double* x, double* y, const double* x_own, double* part, hipStream_t s, const SpmvScale* sc) {
  if (A.ell_coded == 1) return launch_spmv_ell_c<K, RPT, VEC, 1>(A, x, y, x_own, part, s, sc);
  return launch_spmv_ell_c<K, RPT, VEC, 0>(A, x, y, x_own, part, s, sc);
}

int launch_spmv_ell(const CsrDev& A, const double* x, double* y, const double* x_own, double* part, hipStream_t s, const SpmvScale* sc) {
  if (A.ell_coded == 2 || A.ell_coded == 3) {  // (always built with one row per lane and trip: ell_variant 0)
    // Which form runs (tools/spmv_coding_probe.py, partial_step_probe.py): 27 entries per row - one row per lane; 5 entries - two adjacent
    // rows per lane, also with the fused r / beta; 7 entries - two adjacent rows per lane for the plain SpMV, one row per lane for the
    // fused form.  Knob 23 = 1 / 3: the one-row-per-lane kernel with one / two 512-row units per workgroup (A/B: two units measured equal
    // in 2-D, 185 vs 191 us on a 300^3 grid, and 7 % slower for the fused form - 112 registers - which therefore always takes one).
    const int G = (!sc && A.cls_group == 3) ? 2 : 1;
    if (A.fixed_k == 27) return launch_spmv_cls<27, 1, 1>(A, x, y, x_own, part, s, sc);
    if (A.cls_group != 1 && A.cls_group != 3) {  // (knob 23: 1 / 3 = the one-row-per-lane forms, A/B)
      // two adjacent rows per lane: headline 43.5 -> 35.1 us, C2 9.7 -> 8.7, 300^3 7-point 187 -> 178, 464^3 674 -> 624; the fused r / beta
      // form 54.5 -> 51.0 us (5-point) but 235 -> 254 (7-point: 82 registers, five waves) - that one keeps one row per lane
      if (A.fixed_k == 5) return launch_spmv_cls2<5>(A, x, y, x_own, part, s, sc);
      if (!sc) return launch_spmv_cls2<7>(A, x, y, x_own, part, s, sc);
    }
    if (A.fixed_k == 5) return G == 1 ? launch_spmv_cls<5, 2, 1>(A, x, y, x_own, part, s, sc) : launch_spmv_cls<5, 2, 2>(A, x, y, x_own, part, s, sc);
    return G == 1 ? launch_spmv_cls<7, 2, 1>(A, x, y, x_own, part, s, sc) : launch_spmv_cls<7, 2, 2>(A, x, y, x_own, part, s, sc);
  }
  if (A.fixed_k == 27) return launch_spmv_ell_t<27, 1, 1>(A, x, y, x_own, part, s, sc);
  if (A.ell_variant == 1) {
    if (A.fixed_k == 5) return launch_spmv_ell_t<5, 1, 2>(A, x, y, x_own, part, s, sc);
    return launch_spmv_ell_t<7, 1, 2>(A, x, y, x_own, part, s, sc);
  }
  if (A.fixed_k == 5) return launch_spmv_ell_t<5, 2, 1>(A, x, y, x_own, part, s, sc);
  return launch_spmv_ell_t<7, 2, 1>(A, x, y, x_own, part, s, sc);
}

int launch_spmv_csr(const CsrDev& A, const double* x, double* y, const double* x_own, double* part, int flags,
                    hipStream_t s) {
  if (A.rows == 0) return 0;
  if (A.ell_default && ell_usable(A, flags)) return launch_spmv_ell(A, x, y, x_own, part, s);
  if (flags & LZ_FLAG_SPMV_SCALAR) {
    const int grid = (int)((A.rows + kTPB - 1) / kTPB);
    hipLaunchKernelGGL(k_spmv_scalar, dim3(grid), dim3(kTPB), 0, s, A.rowptr, A.colidx, A.vals, x, x_own, y, A.rows, part);
    return grid;
  }
  if (A.pb && !(flags & LZ_FLAG_SPMV_STREAM)) return launch_spmv_pb(A, A.pb, x, y, x_own, part, s);
  if (!(flags & LZ_FLAG_SPMV_STREAM)) {
    if (A.fixed_k == 5) return launch_spmv_fixed<5>(A, x, y, x_own, part, A.fixed_rb, s);
    if (A.fixed_k == 7) return launch_spmv_fixed<7>(A, x, y, x_own, part, A.fixed_rb, s);
  }
  const int grid = A.n_rowblk;
  const size_t lds = (size_t)(A.blk_nnz_cap + 2) * sizeof(double);
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

// ------------------------------------------------------------------ dense GEMV (row-major A)
// One wave per row, 16-byte non-temporal loads of the row (A is streamed once per matvec), x through L1/L2,
// 4 independent accumulators per lane; wave-shuffle reduction; alpha partial per block.
// rows x cols block of a row-partitioned matrix: x has `cols` entries (the whole vector, or the all-gathered padded
// layout when the matrix is split over ranks), x_own the `rows` entries this rank owns (for the alpha partial).
__global__ __launch_bounds__(kTPB) void k_gemv_dense(const double* __restrict__ A, int64_t M, int64_t cols, int64_t lda,
                                                    const double* __restrict__ x, const double* __restrict__ x_own,
                                                    double* __restrict__ y, double* __restrict__ part) {
  __shared__ double sm[kTPB / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * (kTPB / 64) + w;
  double d = 0.0;
  if (row < M) {
    const double acc = gemv_row_wave(A + row * lda, x, cols, lane);
    if (lane == 0) {
      y[row] = acc;
      d = x_own[row] * acc;
    }
  }
  d = block_sum(d, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = d;
}

int launch_gemv_dense(const double* A, int64_t M, int64_t cols, int64_t lda, const double* x, const double* x_own, double* y,
                      double* part, hipStream_t s) {
  const int grid = (int)((M + kTPB / 64 - 1) / (kTPB / 64));
  hipLaunchKernelGGL(k_gemv_dense, dim3(grid), dim3(kTPB), 0, s, A, M, cols, lda, x, x_own, y, part);
  return grid;
}

// ------------------------------------------------------------------ stencil Hamiltonian assembly on the device
// CSR of  sign * T_factor * Laplacian (+ diagonal potential)  on the reference's periodic grid
// (Python/Regular/Hamiltonian.py:73-128, there N^3; here Nx x Ny x Nz): flat index x + y Nx + z Nx Ny, neighbours wrap,
// 7-point weights (-6, 1) or 27-point weights (centre, face, edge, corner) passed in `w` exactly as the host computed them.
// One lane builds one row of the block [row0, row0 + rows_local): P (column, value) pairs, insertion-sorted by GLOBAL
// column like SciPy's sort_indices, then the columns are renumbered for a row-partitioned run (owned: col - row0; remote:
// position in the ghost tail, given as up to 16 contiguous global ranges - what a slab of a stencil needs).
// Values follow SciPy's arithmetic of `-T + V`: t = T_factor * w; entry = sign * t (+ potential on the diagonal).
// Potential: none, a per-row array, or the deuteron hard-core + well of 3Ddeuteron.py:51-61 evaluated here on
// np.linspace(-L/2, L/2, N) coordinates (device exp/pow: last-bit differences from NumPy, hence opt-in).
template <int P>
__global__ __launch_bounds__(kTPB) void k_build_stencil3d(StencilArgs a, const double* __restrict__ pot, int32_t* __restrict__ rowptr,
                                                         int32_t* __restrict__ colidx, double* __restrict__ vals) {
  const int64_t lrow = (int64_t)blockIdx.x * kTPB + threadIdx.x;
  if (lrow > a.rows_local) return;
  if (lrow == a.rows_local) {
    rowptr[lrow] = (int32_t)(lrow * P);
    return;
  }
  rowptr[lrow] = (int32_t)(lrow * P);
  const int64_t row = a.row0 + lrow;
  const int Nx = a.Nx, Ny = a.Ny, Nz = a.Nz;
  const int x = (int)(row % Nx), y = (int)((row / Nx) % Ny), z = (int)(row / ((int64_t)Nx * Ny));
  int64_t c[P];
  double v[P];
  int cnt = 0;
  for (int dz = -1; dz <= 1; ++dz)
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        const int nz = (dx != 0) + (dy != 0) + (dz != 0);
        if (P == 7 && nz > 1) continue;
        const int xx = (x + dx + Nx) % Nx, yy = (y + dy + Ny) % Ny, zz = (z + dz + Nz) % Nz;
        const int64_t col = xx + (int64_t)yy * Nx + (int64_t)zz * Nx * Ny;
        const double w = nz == 0 ? a.w[0] : (nz == 1 ? a.w[1] : (nz == 2 ? a.w[2] : a.w[3]));
        double t = a.tf * w;
        if (a.negate) t = -t;
        if (nz == 0) {
          if (a.pot_kind == 1) t = t + pot[lrow];
          if (a.pot_kind == 2) {
            const double px = -a.par[5] / 2 + x * (a.par[5] / (Nx - 1)), py = -a.par[6] / 2 + y * (a.par[6] / (Ny - 1)),
                         pz = -a.par[7] / 2 + z * (a.par[7] / (Nz - 1));
            const double r = sqrt(px * px + py * py + pz * pz);
            t = t + (a.par[0] * exp(-pow(r / a.par[1], a.par[4])) - a.par[2] * exp(-pow(r / a.par[3], a.par[4])));
          }
        }
        // insertion sort by global column (P <= 27)
        int k = cnt++;
        while (k > 0 && c[k - 1] > col) {
          c[k] = c[k - 1];
          v[k] = v[k - 1];
          --k;
        }
        c[k] = col;
        v[k] = t;
      }
#pragma unroll 1
  for (int k = 0; k < P; ++k) {
    int64_t col = c[k];
    if (a.renumber) {
      if (col >= a.row0 && col < a.row0 + a.rows_local) {
        col -= a.row0;
      } else {
        int64_t ext = 0;  // every remote column of a well-formed plan lies in exactly one range
        for (int q = 0; q < a.nranges; ++q)
          if (col >= a.gstart[q] && col < a.gstart[q] + a.glen[q]) ext = a.gext[q] + (col - a.gstart[q]);
        col = ext;
      }
    }
    colidx[lrow * P + k] = (int32_t)col;
    vals[lrow * P + k] = v[k];
  }
}

void launch_build_stencil3d(const StencilArgs& a, int points, const double* pot, int32_t* rowptr, int32_t* colidx, double* vals,
                            hipStream_t s) {
  const unsigned grid = (unsigned)((a.rows_local + 1 + kTPB - 1) / kTPB);
  if (points == 7)
    hipLaunchKernelGGL(k_build_stencil3d<7>, dim3(grid), dim3(kTPB), 0, s, a, pot, rowptr, colidx, vals);
  else
    hipLaunchKernelGGL(k_build_stencil3d<27>, dim3(grid), dim3(kTPB), 0, s, a, pot, rowptr, colidx, vals);
}

}  // namespace lz
