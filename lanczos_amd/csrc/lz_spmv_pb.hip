// Column-blocked two-phase SpMV for CSR matrices WITHOUT column locality (random graphs: BASELINE config C3).
//
// Why.  r = A v with random columns is a gather of nnz 8-byte values out of a vector that does not fit any cache
// (C3: 8e7 gathers into 80 MB).  Measured on MI355X (profiles/r01/gather_probe_random_graph.json): such gathers run at
// ~50e9/s whether v sits in HBM or in the 256 MB Infinity Cache and ~96e9/s out of an XCD's L2 - the limit is the number
// of outstanding misses, not bytes - so the row-major CSR-stream kernel needs 1.37 ms for a 1.16 GB SpMV (10.6 % of the
// HBM roofline).  The only memory on the chip that serves a random 8-byte read at full rate is the LDS.
//
// How.  Two kernels, each of which gathers out of LDS only (a reorganisation of the technique known as propagation
// blocking; the layout below is this build's own):
//   phase 1, one workgroup per COLUMN block cb (W consecutive entries of v staged into LDS with coalesced loads):
//            streams its matrix entries - stored column-block-major: values `pvals` and 16-bit local columns `pcol` -
//            and writes the products  T[t] = pvals[t] * v[cb*W + pcol[t]]  as one contiguous stream.
//   phase 2, one workgroup per ROW block rb (consecutive rows with <= kPbCap entries): copies its products - one short
//            run per column block - into LDS, then every row adds ITS products out of LDS in CSR order through the
//            16-bit map `perm` (CSR position -> LDS slot).  Same multiplications, same additions in the same order as
//            SciPy's csr_matvec: y is bit-identical to the CSR-stream kernel's (tests/test_gpu_kernels.py).
// T is ordered (column block, row block, -): phase 1 reads and writes purely sequential streams; the only fragmented
// access is phase 2's read of nCB runs of ~kPbCap/nCB products per row block.
// Traffic per SpMV: 26 bytes per entry + ~40 bytes per row (C3: 2.5 GB, all of it streamed) instead of 8e7 cache misses.
//
// The layout is built on the device at lz_set_csr time (three small integer kernels, LDS histograms); the slot an entry
// gets inside its (row block, column block) tile depends on atomic order, which is harmless: `perm` is a bijection onto
// the tile whatever that order is, so every run produces the same bits.
#include <algorithm>
#include <cstdio>
#include <vector>

#include "lz_device.h"

namespace lz {

constexpr int kPbThreads = 1024;
constexpr int kPbCapMax = 15360;   // products per row block and LDS tile of phase 2 (120 + 30 KiB); default 7168: two workgroups per CU
constexpr int kPbMaxRows = 8192;   // rows per row block
constexpr int kPbMaxW = 19968;     // doubles of v per column block: 156 KiB of LDS in phase 1

struct PbDev {
  int nCB = 0, nRB = 0;
  int W = 0;
  int cap = 0;                 // products per row block
  int64_t nnz = 0;
  int32_t* rbptr = nullptr;    // nRB + 1 row-block boundaries
  int4* rbhead = nullptr;      // nRB: {first row, rows, first entry, entries} of every row block
  int32_t* cbptr = nullptr;    // nCB + 1: T range of each column block
  int32_t* toff = nullptr;     // [nRB][nCB]: T offset of tile (cb, rb)
  uint16_t* lstart = nullptr;  // [nRB][nCB + 1]: LDS slot where tile (rb, cb) starts in phase 2
  uint16_t* perm = nullptr;    // nnz: CSR position -> LDS slot within its row block
  uint16_t* pcol = nullptr;    // nnz (T order): column - cb * W
  double* pvals = nullptr;     // nnz (T order)
  double* T = nullptr;         // nnz products
  bool wide_runs = false;      // tiles average more than 20 products: 16 lanes x 3 slots copy a tile (else 8 x 3)
  size_t lds2 = 0;             // dynamic LDS of phase 2: the product tile + this row block's tile table
};

namespace {

__global__ __launch_bounds__(256) void k_pb_hist(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                                 const int32_t* __restrict__ rbptr, int W, int nCB, int32_t* __restrict__ toff) {
  extern __shared__ int hist[];
  const int rb = blockIdx.x;
  for (int c = threadIdx.x; c < nCB; c += blockDim.x) hist[c] = 0;
  __syncthreads();
  const int k0 = rowptr[rbptr[rb]], k1 = rowptr[rbptr[rb + 1]];
  for (int k = k0 + threadIdx.x; k < k1; k += blockDim.x) atomicAdd(&hist[colidx[k] / W], 1);
  __syncthreads();
  for (int c = threadIdx.x; c < nCB; c += blockDim.x) toff[(int64_t)rb * nCB + c] = hist[c];
}

// per column block: exclusive prefix over the row blocks (in place), total to tot[cb]
__global__ __launch_bounds__(256) void k_pb_scan_rb(int32_t* __restrict__ toff, int nRB, int nCB, int32_t* __restrict__ tot) {
  const int cb = blockIdx.x * blockDim.x + threadIdx.x;
  if (cb >= nCB) return;
  int run = 0;
  for (int rb = 0; rb < nRB; ++rb) {
    const int v = toff[(int64_t)rb * nCB + cb];
    toff[(int64_t)rb * nCB + cb] = run;
    run += v;
  }
  tot[cb] = run;
}

// place every entry: T slot, 16-bit local column, and the LDS slot phase 2 will find its product in
__global__ __launch_bounds__(256) void k_pb_place(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                                  const double* __restrict__ vals, const int32_t* __restrict__ rbptr, int W, int nCB,
                                                  int nRB, const int32_t* __restrict__ cbptr, int32_t* __restrict__ toff,
                                                  uint16_t* __restrict__ lstart, uint16_t* __restrict__ perm,
                                                  uint16_t* __restrict__ pcol, double* __restrict__ pvals) {
  extern __shared__ int sm[];
  int* cursor = sm;          // nCB
  int* ls = sm + nCB;        // nCB + 1
  const int rb = blockIdx.x;
  // tile lengths of this row block = difference of the per-column-block prefixes of row blocks rb and rb + 1
  for (int c = threadIdx.x; c < nCB; c += blockDim.x) {
    const int a = toff[(int64_t)rb * nCB + c];
    const int b = rb + 1 < nRB ? toff[(int64_t)(rb + 1) * nCB + c] : cbptr[c + 1] - cbptr[c];
    cursor[c] = 0;
    ls[c + 1] = b - a;
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // setup path: a plain serial scan over <= a few thousand counters
    ls[0] = 0;
    for (int c = 0; c < nCB; ++c) ls[c + 1] += ls[c];
  }
  __syncthreads();
  for (int c = threadIdx.x; c <= nCB; c += blockDim.x) lstart[(int64_t)rb * (nCB + 1) + c] = (uint16_t)ls[c];
  const int k0 = rowptr[rbptr[rb]], k1 = rowptr[rbptr[rb + 1]];
  for (int k = k0 + threadIdx.x; k < k1; k += blockDim.x) {
    const int col = colidx[k];
    const int cb = col / W;
    const int p = atomicAdd(&cursor[cb], 1);
    const int64_t t = (int64_t)cbptr[cb] + toff[(int64_t)rb * nCB + cb] + p;
    pvals[t] = vals[k];
    pcol[t] = (uint16_t)(col - cb * W);
    perm[k] = (uint16_t)(ls[cb] + p);
  }
}

// toff[rb][cb] += cbptr[cb]: absolute T offsets for phase 2
__global__ __launch_bounds__(256) void k_pb_add(int32_t* __restrict__ toff, const int32_t* __restrict__ cbptr, int64_t total, int nCB) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) toff[i] += cbptr[i % nCB];
}

// ---- phase 1: T = pvals * v[columns], column block in LDS
template <int U>
__global__ __launch_bounds__(kPbThreads) void k_pb_products(const int32_t* __restrict__ cbptr, const double* __restrict__ pvals,
                                                           const uint16_t* __restrict__ pcol, const double* __restrict__ x,
                                                           int64_t ncols, int W, double* __restrict__ T) {
  extern __shared__ double xs[];
  const int cb = blockIdx.x;
  const int64_t c0 = (int64_t)cb * W;
  const int wn = (int)(ncols - c0 < W ? ncols - c0 : W);
  // W and c0 are multiples of 32: whole double2 lanes except possibly the very last pair of the vector
  {
    const double2* x2 = reinterpret_cast<const double2*>(x + c0);
    double2* s2 = reinterpret_cast<double2*>(xs);
    const int n2 = wn >> 1;
    for (int i = threadIdx.x; i < n2; i += kPbThreads) s2[i] = x2[i];
    if ((wn & 1) && threadIdx.x == 0) xs[wn - 1] = x[c0 + wn - 1];
  }
  __syncthreads();
  const int64_t t0 = cbptr[cb], t1 = cbptr[cb + 1];
  for (int64_t tb = t0 + threadIdx.x; tb < t1; tb += (int64_t)U * kPbThreads) {
    double a[U];
    uint16_t c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t t = tb + (int64_t)u * kPbThreads;
      const bool ok = t < t1;
      a[u] = ok ? __builtin_nontemporal_load(pvals + t) : 0.0;
      c[u] = ok ? __builtin_nontemporal_load(pcol + t) : (uint16_t)0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t t = tb + (int64_t)u * kPbThreads;
      if (t < t1) T[t] = a[u] * xs[c[u]];
    }
  }
}

// ---- phase 2: row sums out of LDS in CSR order, alpha partial per row block
// The workgroup's life is two memory round trips and no more (it is latency-, not bandwidth-critical: ~7 K products per
// workgroup): (1) the tile table of this row block -> LDS, each thread's row bounds; (2) ALL product runs (GS lanes per
// tile, three slots per lane, every load issued before the first LDS store) and the row block's whole `perm` segment,
// coalesced, into LDS; then one barrier and the sums with both operands in LDS.  Nothing after the barrier touches
// global memory except the y store: a first version fetched the perm entries of rows longer than 8 from global memory
// inside the sum loop, and the workgroup then lived as long as its longest row's chain of dependent loads (45 us; the
// timing ablations of profiles/r02/ablate_pb_rows_and_ritz.json showed 850 of its 1060 us left with every other access
// removed).  Two workgroups per CU (cap = 7168 products: 56 + 14 + 3 KiB of LDS) overlap each other's round trips.
__device__ unsigned long long g_pb_dbg[8];  // kernel-bench build: per-phase wall-clock sums (ABL & 16), 10 ns ticks
#define LZ_PB_STAMP(i)                                                                                   \
  if (ABL & 16) {                                                                                        \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                         \
    const unsigned long long tn_ = wall_clock64();                                                       \
    if (threadIdx.x == 0) atomicAdd(&g_pb_dbg[i], tn_ - tprev_);                                         \
    tprev_ = tn_;                                                                                        \
  }
template <int GS, int TPG, int ABL = 0>  // lanes that copy one tile together; tiles per lane group and trip; ABL: kernel-bench build only
__global__ __launch_bounds__(kPbThreads) __attribute__((amdgpu_waves_per_eu(8, 8)))  // <= 64 VGPRs: two workgroups per CU
void k_pb_rows(const int4* __restrict__ rbhead, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ toff,
               const uint16_t* __restrict__ lstart, const uint16_t* __restrict__ perm, const double* __restrict__ T, int nCB, int cap,
               const double* __restrict__ xown, double* __restrict__ y, double* __restrict__ part) {
  extern __shared__ double seg[];  // cap products, cap perm entries
  __shared__ double red[kPbThreads / 64];
  uint16_t* perm_s = reinterpret_cast<uint16_t*>(seg + cap);
  const int rb = blockIdx.x;
  unsigned long long tprev_ = (ABL & 16) ? wall_clock64() : 0ull;
  const int4 hd = rbhead[rb];  // {first row, rows, first entry, entries}: one (scalar) load instead of a chain of four
  const int r0 = hd.x, r1 = hd.x + hd.y, k0 = hd.z, cnt = hd.w;
  if (ABL & 8) {  // kernel-bench arm: nothing but the dispatch of the workgroup
    if (cnt == -12345) part[rb] = 0.0;
    return;
  }
  // ---- round trip 1: everything whose address is known now
  constexpr int NG = kPbThreads / GS, S = 3;
  const int g = threadIdx.x / GS, l = threadIdx.x % GS;
  const int32_t* to = toff + (int64_t)rb * nCB;
  const uint16_t* ls = lstart + (int64_t)rb * (nCB + 1);
  int off[TPG], a[TPG], len[TPG];
#pragma unroll
  for (int q = 0; q < TPG; ++q) {  // this lane group's tiles of the first (for nCB <= TPG * NG: the only) trip
    const int cb = g + q * NG;
    const bool ok = cb < nCB;
    off[q] = ok ? __builtin_nontemporal_load(to + cb) : 0;
    a[q] = ok ? __builtin_nontemporal_load(ls + cb) : 0;
    len[q] = ok ? __builtin_nontemporal_load(ls + cb + 1) - a[q] : 0;
  }
  const int row = r0 + threadIdx.x;
  int ka = 0, kb = 0;
  double xo = 0.0;
  if (row < r1) {
    ka = rowptr[row] - k0;
    kb = rowptr[row + 1] - k0;
    xo = xown[row];
  }
  constexpr int PP = 8;
  uint16_t pp[PP];
#pragma unroll
  for (int q = 0; q < PP; ++q) {
    const int i = threadIdx.x + q * kPbThreads;
    pp[q] = (!(ABL & 2) && i < cnt) ? __builtin_nontemporal_load(perm + k0 + i) : (uint16_t)(i & 1023);
  }
  LZ_PB_STAMP(0)  // round trip 1 complete (tables, row bounds, x_own, perm)
  // ---- round trip 2: the product runs.  Tile lengths scatter around their mean (Poisson): every lane takes THREE slots of
  // a tile (t = l, l + GS, l + 2 GS), all loads issued before the first LDS store; a tile longer than 3 GS (1e-4 of them)
  // takes the slow loop.
  for (int cb0 = g;; cb0 += TPG * NG) {
    double v[TPG][S];
#pragma unroll
    for (int q = 0; q < TPG; ++q)
#pragma unroll
      for (int u = 0; u < S; ++u) v[q][u] = (!(ABL & 1) && l + u * GS < len[q]) ? __builtin_nontemporal_load(T + off[q] + l + u * GS) : 0.0;
#pragma unroll
    for (int q = 0; q < TPG; ++q) {
#pragma unroll
      for (int u = 0; u < S; ++u)
        if (l + u * GS < len[q]) seg[a[q] + l + u * GS] = v[q][u];
      for (int t = l + S * GS; t < len[q]; t += GS) seg[a[q] + t] = __builtin_nontemporal_load(T + off[q] + t);
    }
    if (cb0 + TPG * NG >= nCB) break;
#pragma unroll
    for (int q = 0; q < TPG; ++q) {  // more column blocks than one trip covers (wide matrices): next trip's table entries
      const int cb = cb0 + TPG * NG + q * NG;
      const bool ok = cb < nCB;
      off[q] = ok ? to[cb] : 0;
      a[q] = ok ? ls[cb] : 0;
      len[q] = ok ? ls[cb + 1] - a[q] : 0;
    }
  }
#pragma unroll
  for (int q = 0; q < PP; ++q) {
    const int i = threadIdx.x + q * kPbThreads;
    if (i < cnt) perm_s[i] = pp[q];
  }
  for (int i = threadIdx.x + PP * kPbThreads; i < cnt; i += kPbThreads) perm_s[i] = perm[k0 + i];  // cap > 8192 only
  LZ_PB_STAMP(1)  // round trip 2 complete, LDS filled
  __syncthreads();
  LZ_PB_STAMP(2)  // barrier
  // ---- sums: both operands in LDS, CSR order, one rounding per add
  double d = 0.0;
  for (int rw = row; rw < r1; rw += kPbThreads) {  // one row per thread, except in row blocks of many short rows
    if (rw != row) {
      ka = rowptr[rw] - k0;
      kb = rowptr[rw + 1] - k0;
      xo = xown[rw];
    }
    double sum = 0.0;
    int k = ka;
    for (; k + 4 <= kb; k += 4) {  // four LDS gathers in flight
      const double p0 = (ABL & 4) ? 1.0 : seg[perm_s[k]], p1 = (ABL & 4) ? 1.0 : seg[perm_s[k + 1]], p2 = (ABL & 4) ? 1.0 : seg[perm_s[k + 2]],
                   p3 = (ABL & 4) ? 1.0 : seg[perm_s[k + 3]];
      sum += p0;
      sum += p1;
      sum += p2;
      sum += p3;
    }
    for (; k < kb; ++k) sum += (ABL & 4) ? 1.0 : seg[perm_s[k]];
    y[rw] = sum;
    d += xo * sum;
  }
  LZ_PB_STAMP(3)  // sums + y store
  d = wave_sum(d);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < kPbThreads / 64; ++i) t += red[i];
    part[rb] = t;
  }
  LZ_PB_STAMP(4)  // block reduction
}

template <class T>
hipError_t pb_alloc(T*& p, size_t count) {
  void* q = nullptr;
  hipError_t e = hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T));
  p = static_cast<T*>(q);
  return e;
}

}  // namespace

void pb_free(PbDev*& pb) {
  if (!pb) return;
  hipFree(pb->rbptr);
  hipFree(pb->rbhead);
  hipFree(pb->cbptr);
  hipFree(pb->toff);
  hipFree(pb->lstart);
  hipFree(pb->perm);
  hipFree(pb->pcol);
  hipFree(pb->pvals);
  hipFree(pb->T);
  delete pb;
  pb = nullptr;
}

// Build the two-phase layout for the device CSR matrix A.  Returns hipSuccess with *out == nullptr when the matrix does
// not qualify (a single row longer than the LDS tile, or too few columns to be worth blocking).
hipError_t pb_build(const CsrDev& A, const int32_t* rowptr_host, PbDev** out, hipStream_t s, int cap_knob) {
  *out = nullptr;
  // products per row block: 7168 by default (two workgroups per CU); rows longer than that (up to 16384 entries) get the larger tile
  int cap = cap_knob > 0 ? std::min(std::max(cap_knob, 1024), kPbCapMax) : 7168;
  if (A.max_row_nnz > cap) cap = kPbCapMax;
  if (A.rows <= 0 || A.nnz <= 0 || A.max_row_nnz > cap) return hipSuccess;
  // Column blocks: as few as the LDS allows (a tile run is ~kPbCap / nCB products: fewer blocks, longer runs), and a
  // multiple of 256 workgroups where that matters for balance.
  int64_t W = round_up((A.ncols + 511) / 512, kPadDoubles);
  if (W < 256) W = 256;
  if (W > kPbMaxW) W = kPbMaxW;
  const int nCB = (int)((A.ncols + W - 1) / W);
  // row blocks: consecutive rows, <= kPbCap entries, <= kPbMaxRows rows
  std::vector<int32_t> rb;
  rb.push_back(0);
  for (int64_t r = 0; r < A.rows;) {
    int64_t e = r;
    const int64_t k0 = rowptr_host[r];
    while (e < A.rows && e - r < kPbMaxRows && (int64_t)rowptr_host[e + 1] - k0 <= cap) ++e;
    rb.push_back((int32_t)e);  // e > r: no row is longer than cap
    r = e;
  }
  const int nRB = (int)rb.size() - 1;
  PbDev* pb = new PbDev();
  pb->nCB = nCB;
  pb->nRB = nRB;
  pb->W = (int)W;
  pb->cap = cap;
  pb->nnz = A.nnz;
  hipError_t e = hipSuccess;
  auto chk = [&](hipError_t x) {
    if (e == hipSuccess && x != hipSuccess) e = x;
  };
  int32_t* tot = nullptr;
  chk(pb_alloc(pb->rbptr, (size_t)nRB + 1));
  chk(pb_alloc(pb->rbhead, (size_t)nRB));
  std::vector<int4> head((size_t)nRB);
  for (int b = 0; b < nRB; ++b)
    head[(size_t)b] = make_int4(rb[(size_t)b], rb[(size_t)b + 1] - rb[(size_t)b], rowptr_host[rb[(size_t)b]], rowptr_host[rb[(size_t)b + 1]] - rowptr_host[rb[(size_t)b]]);
  if (e == hipSuccess) chk(hipMemcpyAsync(pb->rbhead, head.data(), head.size() * sizeof(int4), hipMemcpyHostToDevice, s));
  chk(pb_alloc(pb->cbptr, (size_t)nCB + 1));
  chk(pb_alloc(pb->toff, (size_t)nRB * nCB));
  chk(pb_alloc(pb->lstart, (size_t)nRB * (nCB + 1)));
  chk(pb_alloc(pb->perm, (size_t)A.nnz));
  chk(pb_alloc(pb->pcol, (size_t)A.nnz));
  chk(pb_alloc(pb->pvals, (size_t)A.nnz));
  chk(pb_alloc(pb->T, (size_t)A.nnz));
  chk(pb_alloc(tot, (size_t)nCB));
  if (e == hipSuccess) chk(hipMemcpyAsync(pb->rbptr, rb.data(), rb.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_pb_hist, dim3(nRB), dim3(256), (size_t)nCB * sizeof(int), s, A.rowptr, A.colidx, pb->rbptr, (int)W, nCB, pb->toff);
    hipLaunchKernelGGL(k_pb_scan_rb, dim3((nCB + 255) / 256), dim3(256), 0, s, pb->toff, nRB, nCB, tot);
    chk(hipGetLastError());
  }
  std::vector<int32_t> cbp((size_t)nCB + 1, 0);
  if (e == hipSuccess) chk(hipMemcpyAsync(cbp.data() + 1, tot, (size_t)nCB * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  if (e == hipSuccess) chk(hipStreamSynchronize(s));
  if (e == hipSuccess) {
    for (int c = 0; c < nCB; ++c) cbp[(size_t)c + 1] += cbp[(size_t)c];
    if (cbp[(size_t)nCB] != A.nnz) e = hipErrorUnknown;  // cannot happen: every entry was counted once
  }
  if (e == hipSuccess) chk(hipMemcpyAsync(pb->cbptr, cbp.data(), cbp.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_pb_place, dim3(nRB), dim3(256), (size_t)(2 * nCB + 1) * sizeof(int), s, A.rowptr, A.colidx, A.vals, pb->rbptr,
                       (int)W, nCB, nRB, pb->cbptr, pb->toff, pb->lstart, pb->perm, pb->pcol, pb->pvals);
    const int64_t total = (int64_t)nRB * nCB;
    hipLaunchKernelGGL(k_pb_add, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, pb->toff, pb->cbptr, total, nCB);
    chk(hipGetLastError());
    chk(hipStreamSynchronize(s));
  }
  hipFree(tot);
  pb->wide_runs = (double)pb->nnz / ((double)nRB * (double)nCB) > 20.0;
  pb->lds2 = (size_t)cap * sizeof(double) + (size_t)cap * sizeof(uint16_t);
  // both phases may need more than the default 64 KiB of dynamic LDS: allowed once per kernel, here, so that the
  // launches themselves have no failure mode
  if (e == hipSuccess && pb->lds2 > 160 * 1024) e = hipErrorInvalidValue;  // only with thousands of column blocks AND the large tile
  if (e == hipSuccess && W * sizeof(double) > 65536)
    chk(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pb_products<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(W * sizeof(double))));
  if (e == hipSuccess && pb->lds2 > 65536) {
    chk(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pb_rows<16, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pb->lds2));
    chk(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pb_rows<8, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pb->lds2));
  }
  if (e != hipSuccess) {
    pb_free(pb);
    return e;
  }
  *out = pb;
  return hipSuccess;
}

int pb_num_partials(const PbDev* pb) { return pb->nRB; }

// y = A x (both phases, asynchronous on s); part[rb] = sum over the rows of row block rb of x_own[i] * y[i].
// Returns the number of partials.
int launch_spmv_pb(const CsrDev& A, const PbDev* pb, const double* x, double* y, const double* x_own, double* part, hipStream_t s) {
  const size_t lds1 = (size_t)pb->W * sizeof(double);
  hipLaunchKernelGGL(k_pb_products<8>, dim3(pb->nCB), dim3(kPbThreads), lds1, s, pb->cbptr, pb->pvals, pb->pcol, x, A.ncols, pb->W, pb->T);
#ifdef LZ_KBENCH  // timing-only ablation arms (wrong results): 1 no product loads, 2 no perm loads, 4 no LDS gathers
  if (A.ablation && !pb->wide_runs) {
    auto go = [&](auto kern) {
      hipLaunchKernelGGL(kern, dim3(pb->nRB), dim3(kPbThreads), pb->lds2, s, pb->rbhead, A.rowptr, pb->toff, pb->lstart, pb->perm, pb->T, pb->nCB,
                         pb->cap, x_own, y, part);
    };
    switch (A.ablation) {
      case 1: go(k_pb_rows<8, 4, 1>); break;
      case 2: go(k_pb_rows<8, 4, 2>); break;
      case 3: go(k_pb_rows<8, 4, 3>); break;
      case 4: go(k_pb_rows<8, 4, 4>); break;
      case 7: go(k_pb_rows<8, 4, 7>); break;
      case 8: go(k_pb_rows<8, 4, 8>); break;
      case 16: {
        unsigned long long z[8] = {0};
        hipMemcpyToSymbolAsync(HIP_SYMBOL(g_pb_dbg), z, sizeof z, 0, hipMemcpyHostToDevice, s);
        go(k_pb_rows<8, 4, 16>);
        hipMemcpyFromSymbolAsync(z, HIP_SYMBOL(g_pb_dbg), sizeof z, 0, hipMemcpyDeviceToHost, s);
        hipStreamSynchronize(s);
        fprintf(stderr, "[k_pb_rows phases, us per workgroup (wave 0)] rt1 %.2f  rt2+lds %.2f  barrier %.2f  sums %.2f  reduce %.2f  (%d workgroups)\n",
                z[0] * 0.01 / pb->nRB, z[1] * 0.01 / pb->nRB, z[2] * 0.01 / pb->nRB, z[3] * 0.01 / pb->nRB, z[4] * 0.01 / pb->nRB, pb->nRB);
        break;
      }
      case 9: hipLaunchKernelGGL((k_pb_rows<8, 4, 8>), dim3(pb->nRB), dim3(kPbThreads), 0, s, pb->rbhead, A.rowptr, pb->toff, pb->lstart, pb->perm, pb->T, pb->nCB,
                                 pb->cap, x_own, y, part); break;  // ... without the 72 KiB of LDS
      case 10: hipLaunchKernelGGL((k_pb_rows<8, 4, 8>), dim3(pb->nRB * 4), dim3(256), 0, s, pb->rbhead, A.rowptr, pb->toff, pb->lstart, pb->perm, pb->T, pb->nCB,
                                  pb->cap, x_own, y, part); break;  // ... as four times as many 256-thread workgroups
      default: go(k_pb_rows<8, 4>); break;
    }
    return pb->nRB;
  }
#endif
  if (pb->wide_runs)
    hipLaunchKernelGGL((k_pb_rows<16, 4>), dim3(pb->nRB), dim3(kPbThreads), pb->lds2, s, pb->rbhead, A.rowptr, pb->toff, pb->lstart, pb->perm,
                       pb->T, pb->nCB, pb->cap, x_own, y, part);
  else
    hipLaunchKernelGGL((k_pb_rows<8, 4>), dim3(pb->nRB), dim3(kPbThreads), pb->lds2, s, pb->rbhead, A.rowptr, pb->toff, pb->lstart, pb->perm,
                       pb->T, pb->nCB, pb->cap, x_own, y, part);
  return pb->nRB;
}

}  // namespace lz
