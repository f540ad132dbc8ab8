// Column-blocked two-phase SpMV for CSR matrices WITHOUT column locality (random graphs: BASELINE config C3).
// Selected by lz_set_csr for such matrices (lz_matrix.hip; lz_set_tuning(h, 14, 1) switches it off, 2 forces it); y is
// bit-identical to SciPy's csr_matvec (tests/test_gpu_kernels.py).
//
// Why.  r = A v with random columns is a gather of nnz 8-byte values out of a vector that fits no cache (C3: 8e7 gathers
// into 80 MB).  Such gathers run at ~50-65e9/s on MI355X whether v sits in HBM or in the Infinity Cache and ~96e9/s out of
// an L2 (profiles/r01/gather_probe_random_graph.json) - the limit is the rate of cache-missing accesses, not bytes - so
// the row-major CSR-stream kernel needs 1.24-1.37 ms for a 1.16 GB SpMV.  The only memory on the chip that serves a random
// 8-byte read at full rate is the LDS.
//
// How.  Two kernels, each of which gathers out of LDS only:
//   phase 1, one workgroup per COLUMN block cb (W consecutive entries of v staged into LDS with coalesced loads):
//            streams its matrix entries - stored column-block-major: values `pvals`, 16-bit local columns `pcol` and one
//            destination `gdst` per group of 8 entries - and SCATTERS the products
//            T2[gdst[t / 8] + t % 8] = pvals[t] * v[cb*W + pcol[t]].
//            T2 is ordered (row block, column block, -): the products of one tile (row block x column block) are
//            contiguous, every tile is padded to a multiple of 8 products and starts on a 64-byte boundary, and the pad
//            slots are real (zero-valued) entries of the stream - so every store covers whole 64-byte sectors, and a
//            store is fire-and-forget: nothing in this phase waits for a scattered access.
//   phase 2, one persistent workgroup per CU walking the ROW blocks: the products of a row block are ONE contiguous
//            segment of T2 - a plain coalesced stream into LDS, together with the 16-bit map `perm` (CSR position -> slot in
//            the segment) - then every row adds its products out of LDS in CSR order.  The loads of the NEXT row block are
//            issued into registers before the sums of the current one, so the CU (whose LDS holds one segment only) always
//            has a stream in flight.  Same multiplications, same additions in the same order as SciPy.
// (The first version of this file kept the products column-block-major and let phase 2 fetch 512 short runs per row block:
// two dependent scattered round trips per workgroup, 0.99 ms - slower than the gather it replaced.  Measurements:
// profiles/r02/ablate_pb_rows_and_ritz.json)
//
// Round 3 - fewer bytes (3.02 GB per SpMV against 1.16 GB of CSR-algorithmic bytes):
//   * the DIAGONAL never takes the round trip through T2.  x[row] is an access phase 2 makes anyway (coalesced, for the fused
//     alpha = x . A x), so phase 2 forms a_ii x_i itself from a per-row array `dvals` and parks it in LDS behind the
//     segment; `perm` points the row's sum at it, at the diagonal's position in CSR order - same products, same additions.
//     For D - Adj that is every eighth entry: 28.5 -> 8 bytes each.  (Square single-rank matrices only: in a row-block
//     partition a row's own column is not its local index.)
//   * matrix values travel as fp32 where EVERY value is exactly representable in fp32 (Laplacians, adjacency and stencil
//     weights are small integers): the kernel widens them back - bit-identical products, 4 bytes less per entry.
//
// The layout is built on the device at lz_set_csr time (integer kernels with LDS histograms); the slot an entry gets
// inside its tile depends on atomic order, which is harmless: `perm` is a bijection onto the tile whatever that order
// is, so every run produces the same bits.
#include <algorithm>
#include <cstdio>
#include <type_traits>
#include <vector>

#include <mutex>

#include "lz_device.h"

namespace lz {

constexpr int kPbThreads = 1024;
#ifndef LZ_PB_PAD
#define LZ_PB_PAD 8
#endif
constexpr int kPbPad = LZ_PB_PAD;  // tiles are padded to a multiple of it: 8 products = one 64-byte sector
constexpr int kPbPadLog = kPbPad == 8 ? 3 : kPbPad == 4 ? 2 : 1;
static_assert(kPbPad == (1 << kPbPadLog) && kPbPad >= 2, "tile padding: 2, 4 or 8 products");
constexpr int kPbMaxRows = 2048;   // rows per row block: two per thread of phase 2, whose bounds travel with the prefetch
constexpr int kPbMaxW = 19968;     // doubles of v per column block: 156 KiB of LDS in phase 1
constexpr int kPbLdsMax = 158 * 1024;

struct PbDev {
  int nCB = 0, nRB = 0;
  int W = 0;
  int cap = 0;                 // real products per row block
  int64_t nnz = 0, np = 0;     // entries; entries + pad slots
  int32_t* rbptr = nullptr;    // nRB + 1 row-block boundaries
  int4* rbhead = nullptr;      // nRB: {first row, rows, first CSR entry, entries}
  int2* rbseg = nullptr;       // nRB: {first slot in T2, padded length}
  int32_t* cbptr = nullptr;    // nCB + 1: range of each column block in the padded stream
  uint16_t* perm = nullptr;    // nnz: CSR position -> slot within its row block's segment
  uint16_t* pcol = nullptr;    // np (stream order): column - cb * W
  uint32_t* gdst = nullptr;    // np / 8: slot in T2 of every group of 8 stream entries
  double* pvals = nullptr;     // np (values that need fp64)
  float* pvals32 = nullptr;    // np (every value exactly representable in fp32: half the bytes, same products)
  bool constv = false;         // every off-diagonal value is the SAME number (graph Laplacians D - Adj: -1): no value stream at all
  double cval = 0.0;
  double* dvals = nullptr;     // rows: the diagonal entry of every row (0 where a row has none); nullptr: diagonal not split off
  int maxrows = 0;             // longest row block (rows)
  // A/B arm of round 5 (knob 22 = G > 1; DESIGN.md section 4, "closed"): the two phases interleaved over G groups of row blocks -
  // products(g), rows(g), products(g + 1), ... - so that a group's segment of T2 (+ x) may stay in the Infinity Cache between its
  // two kernels.  grp_rb: G + 1 row-block boundaries (host); grp_ptr[g * nCB + cb]: where group g starts in column block cb's stream.
  int groups = 0;
  std::vector<int> grp_rb;
  int32_t* grp_ptr = nullptr;
  double* T2 = nullptr;        // np products
  int segmax = 0;              // longest padded segment (products)
  int ncu = 256;               // compute units of the device: phase 2's persistent grid
  size_t lds2 = 0;             // dynamic LDS of phase 2: the longest segment + an aligned window of cap perm entries
};

namespace {

__host__ __device__ __forceinline__ int pad8(int n) { return (n + kPbPad - 1) & ~(kPbPad - 1); }

// per row block: tile lengths (entries per column block) and the padded segment length.  `diag`: entries with
// column == row are not part of the stream (phase 2 forms them itself)
__global__ __launch_bounds__(256) void k_pb_hist(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                                 const int32_t* __restrict__ rbptr, int W, int nCB, int diag, int32_t* __restrict__ len,
                                                 int32_t* __restrict__ segtot) {
  extern __shared__ int hist[];
  __shared__ int tot;
  const int rb = blockIdx.x;
  for (int c = threadIdx.x; c < nCB; c += blockDim.x) hist[c] = 0;
  if (threadIdx.x == 0) tot = 0;
  __syncthreads();
  for (int row = rbptr[rb] + threadIdx.x; row < rbptr[rb + 1]; row += blockDim.x)
    for (int k = rowptr[row]; k < rowptr[row + 1]; ++k) {
      const int col = colidx[k];
      if (!(diag && col == row)) atomicAdd(&hist[col / W], 1);
    }
  __syncthreads();
  int mine = 0;
  for (int c = threadIdx.x; c < nCB; c += blockDim.x) {
    len[(int64_t)rb * nCB + c] = hist[c];
    mine += pad8(hist[c]);
  }
  atomicAdd(&tot, mine);
  __syncthreads();
  if (threadIdx.x == 0) segtot[rb] = tot;
}

// per column block: exclusive prefix over the row blocks of the PADDED tile lengths, total to tot[cb]
__global__ __launch_bounds__(256) void k_pb_scan_rb(const int32_t* __restrict__ len, int32_t* __restrict__ toff, int nRB, int nCB,
                                                    int32_t* __restrict__ tot) {
  const int cb = blockIdx.x * blockDim.x + threadIdx.x;
  if (cb >= nCB) return;
  int run = 0;
  for (int rb = 0; rb < nRB; ++rb) {
    toff[(int64_t)rb * nCB + cb] = run;
    run += pad8(len[(int64_t)rb * nCB + cb]);
  }
  tot[cb] = run;
}

// place every entry (and every pad slot): stream position, 16-bit local column, destination in T2, CSR -> slot map
template <class VT>
__global__ __launch_bounds__(256) void k_pb_place(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                                  const double* __restrict__ vals, const int32_t* __restrict__ rbptr, int W, int nCB,
                                                  const int32_t* __restrict__ cbptr, const int32_t* __restrict__ len,
                                                  const int32_t* __restrict__ toff, const int2* __restrict__ rbseg,
                                                  uint16_t* __restrict__ perm, uint16_t* __restrict__ pcol,
                                                  uint32_t* __restrict__ gdst, VT* __restrict__ pvals, double* __restrict__ dvals) {
  extern __shared__ int sm[];
  int* cursor = sm;          // nCB
  int* ls = sm + nCB;        // nCB + 1: first slot of every tile inside the segment
  const int rb = blockIdx.x;
  for (int c = threadIdx.x; c < nCB; c += blockDim.x) {
    cursor[c] = 0;
    ls[c + 1] = pad8(len[(int64_t)rb * nCB + c]);
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // setup path: a plain serial scan over <= a few thousand counters
    ls[0] = 0;
    for (int c = 0; c < nCB; ++c) ls[c + 1] += ls[c];
  }
  __syncthreads();
  const uint32_t segbase = (uint32_t)rbseg[rb].x;
  const int dbase = rbseg[rb].y;  // the diagonal products sit right behind this row block's padded segment in phase 2's LDS image
  const int r0 = rbptr[rb];
  for (int row = r0 + threadIdx.x; row < rbptr[rb + 1]; row += blockDim.x)
    for (int k = rowptr[row]; k < rowptr[row + 1]; ++k) {
      const int col = colidx[k];
      if (dvals != nullptr && col == row) {  // the diagonal: phase 2 forms vals[k] * x[row] itself, in LDS slot dbase + local row
        dvals[row] = vals[k];
        perm[k] = (uint16_t)(dbase + (row - r0));
        continue;
      }
      const int cb = col / W;
      const int p = atomicAdd(&cursor[cb], 1);
      const int64_t t = (int64_t)cbptr[cb] + toff[(int64_t)rb * nCB + cb] + p;
      if (pvals) pvals[t] = (VT)vals[k];
      pcol[t] = (uint16_t)(col - cb * W);
      perm[k] = (uint16_t)(ls[cb] + p);
    }
  for (int c = threadIdx.x; c < nCB; c += blockDim.x) {
    const int n = len[(int64_t)rb * nCB + c];
    const int64_t t0 = (int64_t)cbptr[c] + toff[(int64_t)rb * nCB + c];  // a multiple of 8: every tile is padded
    for (int p = n; p < pad8(n); ++p) {  // pad slots: zero-valued entries, so that whole sectors are written
      if (pvals) pvals[t0 + p] = (VT)0;  // (without a value stream a pad slot receives cval * v[c0]: never read - perm maps real entries only)
      pcol[t0 + p] = 0;
    }
    for (int g = 0; g < pad8(n) / kPbPad; ++g) gdst[(t0 >> kPbPadLog) + g] = segbase + (uint32_t)(ls[c] + kPbPad * g);
  }
}

// A/B arm (knob 22): grp_ptr[g * nCB + cb] = stream position where row-block group g starts inside column block cb
__global__ __launch_bounds__(256) void k_pb_group_ptr(const int32_t* __restrict__ cbptr, const int32_t* __restrict__ toff, int nRB, int nCB, int G,
                                                     const int32_t* __restrict__ grb, int32_t* __restrict__ grp_ptr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (G + 1) * nCB) return;
  const int g = i / nCB, cb = i - g * nCB;
  const int rb = grb[g];
  grp_ptr[i] = rb < nRB ? cbptr[cb] + toff[(int64_t)rb * nCB + cb] : cbptr[cb + 1];
}

// ---- phase 1: T2[gdst[t / 8] + t % 8] = pvals[t] * v[columns], column block in LDS.  A lane takes PAIRS of entries (one
// 16-byte value load, one 4-byte column load, a 16-byte product store); four lanes share a group's destination.
__device__ __forceinline__ double2 pb_ld_vals(const double* pv, int64_t pair) { return ld_stream<1>(reinterpret_cast<const double2*>(pv) + pair); }
__device__ __forceinline__ double2 pb_ld_vals(const float* pv, int64_t pair) {
  typedef float f2v_t __attribute__((ext_vector_type(2)));
  const f2v_t v = __builtin_nontemporal_load(reinterpret_cast<const f2v_t*>(pv) + pair);
  return make_double2((double)v.x, (double)v.y);  // exact: the layout keeps fp32 only where every value round-trips
}

// no value stream (PbDev::constv): every product is cval * v[column]
struct PbConst {
  double c;
};
__device__ __forceinline__ double2 pb_ld_vals(const PbConst* pv, int64_t) { return make_double2(0.0, 0.0); }

template <int U, class VT>
__global__ __launch_bounds__(kPbThreads) void k_pb_products(const int32_t* __restrict__ cbptr, const VT* __restrict__ pvals,
                                                           const uint16_t* __restrict__ pcol, const uint32_t* __restrict__ gdst,
                                                           const double* __restrict__ x, int64_t ncols, int W,
                                                           double* __restrict__ T2, double cval, const int32_t* __restrict__ grp_ptr, int grp) {
  constexpr bool kConst = std::is_same<VT, PbConst>::value;
  extern __shared__ double xs[];
  const int cb = blockIdx.x;
  const int64_t c0 = (int64_t)cb * W;
  const int wn = (int)(ncols - c0 < W ? ncols - c0 : W);
  // W and c0 are multiples of 32: whole double2 lanes except possibly the very last pair of the vector.
  // The column block of v -> LDS with ALL of a thread's loads in flight before its first LDS store (W <= 19 968 doubles: at most
  // ten 16-byte loads per thread); as a load -> wait -> store loop this was ten dependent round trips per workgroup (round 3).
  {
    constexpr int NX = (kPbMaxW / 2 + kPbThreads - 1) / kPbThreads;
    const double2* x2 = reinterpret_cast<const double2*>(x + c0);
    double2* s2 = reinterpret_cast<double2*>(xs);
    const int n2 = wn >> 1;
    double2 tmp[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int q = threadIdx.x + i * kPbThreads;
      tmp[i] = q < n2 ? x2[q] : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int q = threadIdx.x + i * kPbThreads;
      if (q < n2) s2[q] = tmp[i];
    }
    if ((wn & 1) && threadIdx.x == 0) xs[wn - 1] = x[c0 + wn - 1];
  }
  __syncthreads();
  // stream positions are multiples of 8.  grp_ptr (A/B arm, knob 22): only the part of this column block's stream that belongs to
  // row-block group `grp` (a column block's stream is ordered by row block: k_pb_place)
  const int64_t p0 = (int64_t)(grp_ptr ? grp_ptr[(int64_t)grp * gridDim.x + cb] : cbptr[cb]) >> 1;
  const int64_t p1 = (int64_t)(grp_ptr ? grp_ptr[(int64_t)(grp + 1) * gridDim.x + cb] : cbptr[cb + 1]) >> 1;
  if (p1 <= p0) return;  // (uniform over the block; before any barrier-dependent work: the staging above is complete)
  const uint32_t* pc2 = reinterpret_cast<const uint32_t*>(pcol);
  constexpr int64_t B = (int64_t)U * kPbThreads;  // pairs per batch
  // One batch: lane p takes the pairs p, p + 1024, ...: a value pair, a column pair, the destination of its group of 8.
  // FULL batches are unconditional and software-pipelined - the loads of batch b + 1 are issued BEFORE the scattered stores of
  // batch b - for the sake of the wait counts: vmcnt counts loads and stores alike, in issue order, so a load issued after a
  // store cannot be waited for without waiting for that store's acknowledgement from HBM first (a non-temporal, scattered
  // 64-byte sector: microseconds).  In the old loop every batch sat that out (s_waitcnt vmcnt(0) once per batch).
  auto load = [&](int64_t base, double2 (&a)[U], uint32_t (&c)[U], uint32_t (&d)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t p = base + (int64_t)u * kPbThreads;
      a[u] = kConst ? make_double2(cval, cval) : pb_ld_vals(pvals, p);
      c[u] = __builtin_nontemporal_load(pc2 + p);
      d[u] = __builtin_nontemporal_load(gdst + (p >> (kPbPadLog - 1)));
    }
  };
  auto store = [&](int64_t base, const double2 (&a)[U], const uint32_t (&c)[U], const uint32_t (&d)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t p = base + (int64_t)u * kPbThreads;
      // streamed once, read back by phase 2 from HBM: non-temporal
      st_stream<1>(reinterpret_cast<double2*>(T2 + d[u] + (((uint32_t)p & (uint32_t)(kPbPad / 2 - 1)) << 1)),
                   make_double2(a[u].x * xs[c[u] & 0xffffu], a[u].y * xs[c[u] >> 16]));
    }
  };
  const int64_t nfull = (p1 - p0) / B;
  int64_t base = p0 + threadIdx.x;
  if (nfull > 0) {
    double2 a[U], an[U];
    uint32_t c[U], d[U], cn[U], dn[U];
    load(base, a, c, d);
    for (int64_t b = 1; b < nfull; ++b) {
      load(base + B, an, cn, dn);
      store(base, a, c, d);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        a[u] = an[u];
        c[u] = cn[u];
        d[u] = dn[u];
      }
      base += B;
    }
    store(base, a, c, d);
    base += B;
  }
  // the ragged rest (less than one batch)
  {
    double2 a[U];
    uint32_t c[U], d[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t p = base + (int64_t)u * kPbThreads;
      const bool ok = p < p1;
      a[u] = ok ? (kConst ? make_double2(cval, cval) : pb_ld_vals(pvals, p)) : make_double2(0.0, 0.0);
      c[u] = ok ? __builtin_nontemporal_load(pc2 + p) : 0u;
      d[u] = ok ? __builtin_nontemporal_load(gdst + (p >> (kPbPadLog - 1))) : 0u;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t p = base + (int64_t)u * kPbThreads;
      if (p < p1)
        st_stream<1>(reinterpret_cast<double2*>(T2 + d[u] + (((uint32_t)p & (uint32_t)(kPbPad / 2 - 1)) << 1)),
                     make_double2(a[u].x * xs[c[u] & 0xffffu], a[u].y * xs[c[u] >> 16]));
    }
  }
}

// ---- phase 2: one contiguous segment of products + its perm entries -> LDS, row sums in CSR order, alpha partial.
// Persistent: block b takes the row blocks b, b + gridDim.x, ...; a tile travels HBM -> registers -> LDS and the
// registers of the next tile are in flight while the current one is summed.
constexpr int kPbNQ = 10;  // double2 per thread: 20480 products >= the longest segment the LDS can hold
constexpr int kPbNP = 2;   // uint4 (8 perm entries) per thread: 16384 >= cap + 16
constexpr int kPbNR = kPbMaxRows / kPbThreads;  // rows per thread: their bounds are prefetched with the tile

typedef unsigned int u4v_t __attribute__((ext_vector_type(4)));

struct PbTile {
  double2 q[kPbNQ];
  u4v_t pm[kPbNP];
  int ka[kPbNR], kb[kPbNR];
  double xo[kPbNR];
  double dv[kPbNR];  // the rows' diagonal entries (dvals), 0 without the diagonal split
};

__device__ __forceinline__ void pb_tile_load(PbTile& t, int4 hd, int2 sg, const int32_t* __restrict__ rowptr,
                                             const uint16_t* __restrict__ perm, const double* __restrict__ T2,
                                             const double* __restrict__ xown, const double* __restrict__ dvals) {
  const double2* src = reinterpret_cast<const double2*>(T2 + sg.x);  // 64-byte aligned, a multiple of 8 products long
  const int n2 = sg.y >> 1;
#pragma unroll
  for (int i = 0; i < kPbNQ; ++i) {
    const int p = threadIdx.x + i * kPbThreads;
    if (p < n2) t.q[i] = ld_stream<1>(src + p);
  }
  const int kbase = hd.z & ~7;  // aligned window of perm: 16-byte loads
  const int n8 = (hd.z + hd.w - kbase + 7) >> 3;
  const u4v_t* pm8 = reinterpret_cast<const u4v_t*>(perm + kbase);
#pragma unroll
  for (int i = 0; i < kPbNP; ++i) {
    const int p = threadIdx.x + i * kPbThreads;
    if (p < n8) t.pm[i] = __builtin_nontemporal_load(pm8 + p);
  }
#pragma unroll
  for (int i = 0; i < kPbNR; ++i) {
    const int row = hd.x + threadIdx.x + i * kPbThreads;
    t.ka[i] = t.kb[i] = 0;
    t.xo[i] = 0.0;
    t.dv[i] = 0.0;
    if (row < hd.x + hd.y) {
      t.ka[i] = rowptr[row];  // raw: nothing here may wait for a load (the consumer subtracts the window base)
      t.kb[i] = rowptr[row + 1];
      t.xo[i] = xown[row];
      if (dvals != nullptr) t.dv[i] = __builtin_nontemporal_load(dvals + row);
    }
  }
}

__global__ __launch_bounds__(kPbThreads) void k_pb_rows(const int4* __restrict__ rbhead, const int2* __restrict__ rbseg,
                                                       const int32_t* __restrict__ rowptr, const uint16_t* __restrict__ perm,
                                                       const double* __restrict__ T2, int segcap, int nRB,
                                                       const double* __restrict__ xown, double* __restrict__ y,
                                                       double* __restrict__ part, const double* __restrict__ dvals, int rb0) {
  extern __shared__ double seg[];  // per row block: its padded segment of products | one diagonal product per row | the aligned window of its perm entries
  __shared__ double red[kPbThreads / 64];
  int rb = rb0 + blockIdx.x;  // row blocks [rb0, nRB): the grid is never larger than their number
  int4 hd = rbhead[rb];
  int2 sg = rbseg[rb];
  PbTile t;
  pb_tile_load(t, hd, sg, rowptr, perm, T2, xown, dvals);
  for (;;) {
    // registers -> LDS.  The explicit vmcnt(0) tells the compiler's wait-count pass, on every path, that nothing is
    // outstanding from here on - otherwise the predicated loads below make it wait in the middle of the next prefetch.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    const int dbase = sg.y;  // (a multiple of 8)
    uint16_t* perm_s = reinterpret_cast<uint16_t*>(seg + ((dbase + (dvals != nullptr ? hd.y : 0) + 1) & ~1));
    {
      double2* dst = reinterpret_cast<double2*>(seg);
      const int n2 = sg.y >> 1;
#pragma unroll
      for (int i = 0; i < kPbNQ; ++i) {
        const int p = threadIdx.x + i * kPbThreads;
        if (p < n2) dst[p] = t.q[i];
      }
      const int n8 = (hd.z + hd.w - (hd.z & ~7) + 7) >> 3;
      u4v_t* pd = reinterpret_cast<u4v_t*>(perm_s);
#pragma unroll
      for (int i = 0; i < kPbNP; ++i) {
        const int p = threadIdx.x + i * kPbThreads;
        if (p < n8) pd[p] = t.pm[i];
      }
    }
    int ka[kPbNR], kb[kPbNR];
    double xo[kPbNR];
#pragma unroll
    for (int i = 0; i < kPbNR; ++i) {
      ka[i] = t.ka[i];
      kb[i] = t.kb[i];
      xo[i] = t.xo[i];
      // the diagonal's product a_ii * x_i, formed here (the same fp64 multiply phase 1 would have made) and parked behind the
      // segment, where this row's perm entry for the diagonal points
      if (dvals != nullptr && threadIdx.x + i * kPbThreads < hd.y) seg[dbase + threadIdx.x + i * kPbThreads] = t.dv[i] * xo[i];
      // keep the compiler from folding "- kbase" into the prefetch (it would wait for the loads right where they are issued)
      asm volatile("" : "+v"(ka[i]), "+v"(kb[i]));
    }
    const int r0 = hd.x, r1 = hd.x + hd.y, kbase = hd.z & ~7;
    const int cur = rb;
    __syncthreads();
    // the next row block's stream goes out before this one is summed
    rb += gridDim.x;
    const bool more = rb < nRB;  // uniform over the block
    if (more) {
      hd = rbhead[rb];
      sg = rbseg[rb];
      pb_tile_load(t, hd, sg, rowptr, perm, T2, xown, dvals);
    }
    double d = 0.0;
#pragma unroll
    for (int i = 0; i < kPbNR; ++i) {  // no global load in here: the prefetch above stays in flight until the next trip
      const int rw = r0 + threadIdx.x + i * kPbThreads;
      if (rw < r1) {
        const int a = ka[i] - kbase, b = kb[i] - kbase;
        double sum = 0.0;
        int k = a;
        for (; k + 4 <= b; k += 4) {  // four LDS gathers in flight; the adds stay in CSR order, one rounding each
          const double p0 = seg[perm_s[k]], p1 = seg[perm_s[k + 1]], p2 = seg[perm_s[k + 2]], p3 = seg[perm_s[k + 3]];
          sum += p0;
          sum += p1;
          sum += p2;
          sum += p3;
        }
        for (; k < b; ++k) sum += seg[perm_s[k]];
        y[rw] = sum;
        d += xo[i] * sum;
      }
    }
    d = wave_sum(d);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
    __syncthreads();  // every thread is done with this segment (and red is complete)
    if (threadIdx.x == 0) {
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < kPbThreads / 64; ++i) s += red[i];
      part[cur] = s;
    }
    if (!more) break;
  }
}

// once per process, DEVICE and kernel: dynamic-LDS limit = the CU's 160 KiB (result remembered per device ordinal, returned on
// every call).  hipFuncSetAttribute applies to the current device's function object only, so a second handle on another GPU of
// the same process (Lanczos.device_id = 1 after 0) needs its own raise; the table is guarded by a mutex.
constexpr size_t kPbLdsCuMax = 160 * 1024 - 1024;  // the CU's 160 KiB less the kernels' static LDS (the attribute counts dynamic bytes only)
hipError_t pb_raise_lds_limits() {
  constexpr int kMaxDev = 64;
  static std::mutex mu;
  static hipError_t done[kMaxDev];
  static bool init = false;
  int dev = 0;
  const hipError_t ge = hipGetDevice(&dev);
  if (ge != hipSuccess) return ge;
  std::lock_guard<std::mutex> lk(mu);
  if (!init) {
    for (auto& d : done) d = hipErrorNotReady;
    init = true;
  }
  const bool tracked = dev >= 0 && dev < kMaxDev;
  if (tracked && done[dev] != hipErrorNotReady) return done[dev];
  hipError_t e = hipSuccess;
  auto up = [&](const void* k) {
    const hipError_t x = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPbLdsCuMax);
    if (e == hipSuccess) e = x;
  };
  up(reinterpret_cast<const void*>(k_pb_products<4, double>));
  up(reinterpret_cast<const void*>(k_pb_products<4, float>));
  up(reinterpret_cast<const void*>(k_pb_products<4, PbConst>));
  up(reinterpret_cast<const void*>(k_pb_rows));
  if (tracked) done[dev] = e;
  return e;
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
  if (pb) hipFree(pb->grp_ptr);
  if (!pb) return;
  hipFree(pb->rbptr);
  hipFree(pb->rbhead);
  hipFree(pb->rbseg);
  hipFree(pb->cbptr);
  hipFree(pb->perm);
  hipFree(pb->pcol);
  hipFree(pb->gdst);
  hipFree(pb->pvals);
  hipFree(pb->pvals32);
  hipFree(pb->dvals);
  hipFree(pb->T2);
  delete pb;
  pb = nullptr;
}

// Build the two-phase layout for the device CSR matrix A.  Returns hipSuccess with *out == nullptr when the matrix does
// not qualify (a single row longer than the LDS tile).  A.host_colidx / A.host_vals (the caller's arrays, valid during
// lz_set_csr) switch on the diagonal split and the fp32 value stream where they apply.
hipError_t pb_build(const CsrDev& A, const int32_t* rowptr_host, PbDev** out, hipStream_t s, int cap_knob, int groups) {
  *out = nullptr;
  if (A.rows <= 0 || A.nnz <= 0) return hipSuccess;
  // Column blocks: as few as phase 1's LDS allows (a tile holds ~cap / nCB products: fewer blocks, longer tiles, less padding).
  int64_t W = round_up((A.ncols + 511) / 512, kPadDoubles);
  if (W < 256) W = 256;
  if (W > kPbMaxW) W = kPbMaxW;
  const int nCB = (int)((A.ncols + W - 1) / W);
  // The diagonal split needs a row's own column to be its local index: the whole square matrix on this rank.
  bool diag = A.host_colidx != nullptr && A.rows == A.ncols;
  bool f32 = A.host_vals != nullptr;
  bool constv = A.host_vals != nullptr && A.host_colidx != nullptr;
  double cval = 0.0;
  std::vector<uint8_t> ndiag;  // per row: how many of its entries sit on the diagonal (row-block planning below)
  if (A.host_colidx) {
    // One sweep over the caller's arrays on a few host threads (round 3: three single-thread sweeps over all nnz):
    //  * a row that stores its diagonal TWICE (unsummed duplicates are legal CSR) has one slot for two products: no split then
    //  * fp32 value stream only where EVERY value round-trips through float (NaN fails too: such a matrix keeps fp64)
    //  * no value stream at all where every OFF-diagonal value is one and the same number (graph Laplacians: -1)
    ndiag.assign((size_t)A.rows, 0);
    bool have_c = false;
    for (int64_t r = 0; r < A.rows && !have_c; ++r)
      for (int64_t k = rowptr_host[r]; k < rowptr_host[r + 1]; ++k)
        if (A.host_colidx[k] != r) {
          cval = A.host_vals ? A.host_vals[k] : 0.0;
          have_c = true;
          break;
        }
    struct Part {
      bool dup = false, not32 = false, notconst = false;
    } parts[kMaxHostThreads];
    const bool square = A.rows == A.ncols;
    parallel_ranges(A.rows, 1 << 16, [&](int t, int64_t lo, int64_t hi) {
      Part p;
      for (int64_t r = lo; r < hi; ++r) {
        int nd = 0;
        for (int64_t k = rowptr_host[r]; k < rowptr_host[r + 1]; ++k) {
          const bool on = square && A.host_colidx[k] == r;
          nd += on;
          if (A.host_vals) {
            const double v = A.host_vals[k];
            if ((double)(float)v != v) p.not32 = true;
            if (!on && !(v == cval)) p.notconst = true;
          }
        }
        if (nd > 1) p.dup = true;
        ndiag[(size_t)r] = (uint8_t)std::min(nd, 255);
      }
      parts[t] = p;
    });
    for (const Part& p : parts) {
      if (p.dup) diag = false;
      if (p.not32) f32 = false;
      if (p.notconst) constv = false;
    }
    if (!have_c || !diag) constv = false;  // (without the diagonal split the diagonal values ride in the stream too)
  }
  // Row blocks.  Phase 2 keeps in LDS: the padded segment (off-diagonal products + at most 7 pad slots per column block),
  // one diagonal product per row, and an aligned window of the block's perm entries (2 bytes per CSR entry).
  const int64_t lds_budget = kPbLdsMax - 32 - (int64_t)(kPbPad - 1) * nCB * 8 - 32;
  const int ent_max = kPbNP * 8 * kPbThreads - 16;                       // perm window held in registers by the prefetch
  const int off_max = kPbNQ * 2 * kPbThreads - (kPbPad - 1) * nCB - 16;  // products held in registers by the prefetch
  if (lds_budget < 4096 || off_max < 256) return hipSuccess;
  int cap = cap_knob > 0 ? cap_knob : ent_max;  // entries per row block: as many as fit (longer tiles, less padding: measured best)
  if (cap > ent_max) cap = ent_max;
  if (cap < 256) return hipSuccess;
  std::vector<int32_t> rb;
  rb.push_back(0);
  int maxrows = 0, maxent = 0;
  for (int64_t r = 0; r < A.rows;) {
    int64_t e = r, ent = 0, off = 0;
    while (e < A.rows && e - r < kPbMaxRows) {
      const int64_t nr = (int64_t)rowptr_host[e + 1] - rowptr_host[e];
      const int64_t nd = diag ? ndiag[(size_t)e] : 0;
      const int64_t ent2 = ent + nr, off2 = off + nr - nd, rows2 = e - r + 1;
      if (ent2 > cap || off2 > off_max || 8 * off2 + (diag ? 8 * rows2 : 0) + 2 * (ent2 + 16) > lds_budget) break;
      ent = ent2;
      off = off2;
      ++e;
    }
    if (e == r) return hipSuccess;  // a single row does not fit: not applicable (the CSR-stream kernel handles it)
    rb.push_back((int32_t)e);
    maxrows = std::max(maxrows, (int)(e - r));
    maxent = std::max(maxent, (int)ent);
    r = e;
  }
  const int nRB = (int)rb.size() - 1;
  PbDev* pb = new PbDev();
  pb->nCB = nCB;
  pb->nRB = nRB;
  pb->W = (int)W;
  pb->cap = maxent;
  pb->nnz = A.nnz;
  pb->maxrows = maxrows;
  hipError_t e = hipSuccess;
  {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
      pb->ncu = cus;
  }
  auto chk = [&](hipError_t x) {
    if (e == hipSuccess && x != hipSuccess) e = x;
  };
  int32_t *len = nullptr, *toff = nullptr, *tot = nullptr, *segtot = nullptr;
  chk(pb_alloc(pb->rbptr, (size_t)nRB + 1));
  chk(pb_alloc(pb->rbhead, (size_t)nRB));
  chk(pb_alloc(pb->rbseg, (size_t)nRB));
  chk(pb_alloc(pb->cbptr, (size_t)nCB + 1));
  chk(pb_alloc(pb->perm, (size_t)A.nnz + 16));  // phase 2 reads whole aligned 16-byte groups
  chk(pb_alloc(len, (size_t)nRB * nCB));
  chk(pb_alloc(toff, (size_t)nRB * nCB));
  chk(pb_alloc(tot, (size_t)nCB));
  chk(pb_alloc(segtot, (size_t)nRB));
  if (diag) {
    chk(pb_alloc(pb->dvals, (size_t)A.rows));
    if (e == hipSuccess) chk(hipMemsetAsync(pb->dvals, 0, (size_t)A.rows * sizeof(double), s));
  }
  std::vector<int4> head((size_t)nRB);
  for (int b = 0; b < nRB; ++b)
    head[(size_t)b] = make_int4(rb[(size_t)b], rb[(size_t)b + 1] - rb[(size_t)b], rowptr_host[rb[(size_t)b]],
                                rowptr_host[rb[(size_t)b + 1]] - rowptr_host[rb[(size_t)b]]);
  if (e == hipSuccess) chk(hipMemcpyAsync(pb->rbhead, head.data(), head.size() * sizeof(int4), hipMemcpyHostToDevice, s));
  if (e == hipSuccess) chk(hipMemcpyAsync(pb->rbptr, rb.data(), rb.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_pb_hist, dim3(nRB), dim3(256), (size_t)nCB * sizeof(int), s, A.rowptr, A.colidx, pb->rbptr, (int)W, nCB, diag ? 1 : 0, len,
                       segtot);
    hipLaunchKernelGGL(k_pb_scan_rb, dim3((nCB + 255) / 256), dim3(256), 0, s, len, toff, nRB, nCB, tot);
    chk(hipGetLastError());
  }
  std::vector<int32_t> cbp((size_t)nCB + 1, 0), st((size_t)nRB, 0);
  std::vector<int2> seg((size_t)nRB);
  if (e == hipSuccess) chk(hipMemcpyAsync(cbp.data() + 1, tot, (size_t)nCB * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  if (e == hipSuccess) chk(hipMemcpyAsync(st.data(), segtot, (size_t)nRB * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  if (e == hipSuccess) chk(hipStreamSynchronize(s));
  int64_t np = 0;
  int segmax = 0;
  if (e == hipSuccess) {
    int64_t run = 0;
    for (int c = 0; c < nCB; ++c) {
      run += cbp[(size_t)c + 1];
      if (run >= ((int64_t)1 << 31)) e = hipErrorInvalidValue;
      cbp[(size_t)c + 1] = (int32_t)run;
    }
    for (int b = 0; b < nRB; ++b) {
      seg[(size_t)b] = make_int2((int)np, st[(size_t)b]);
      np += st[(size_t)b];
      segmax = std::max(segmax, st[(size_t)b]);
    }
    if (np != run) e = hipErrorUnknown;  // cannot happen: both count every tile's padded length once
  }
  pb->np = np;
  pb->segmax = segmax;
  // phase 2's LDS image of row block b: [0, seg_b) products | [seg_b, seg_b + rows_b) diagonal products | perm window of ent_b + 16 entries
  size_t lds2 = 0;
  for (int b = 0; b < nRB; ++b) {
    const size_t dprod = diag ? (size_t)head[(size_t)b].y : 0;
    lds2 = std::max(lds2, ((st[(size_t)b] + dprod + 1) & ~(size_t)1) * sizeof(double) + (size_t)(head[(size_t)b].w + 16 + 8) * sizeof(uint16_t));
  }
  pb->lds2 = lds2;
  chk(pb_alloc(pb->pcol, (size_t)np + 8));
  chk(pb_alloc(pb->gdst, (size_t)np / kPbPad + 8));
  pb->constv = constv;
  pb->cval = cval;
  if (constv) {
  } else if (f32)
    chk(pb_alloc(pb->pvals32, (size_t)np + 8));
  else
    chk(pb_alloc(pb->pvals, (size_t)np + 8));
  chk(pb_alloc(pb->T2, (size_t)np + 8));
  if (e == hipSuccess) chk(hipMemcpyAsync(pb->cbptr, cbp.data(), cbp.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
  if (e == hipSuccess) chk(hipMemcpyAsync(pb->rbseg, seg.data(), seg.size() * sizeof(int2), hipMemcpyHostToDevice, s));
  if (e == hipSuccess) {
    const size_t sm = (size_t)(2 * nCB + 1) * sizeof(int);
    if (constv)
      hipLaunchKernelGGL(k_pb_place<float>, dim3(nRB), dim3(256), sm, s, A.rowptr, A.colidx, A.vals, pb->rbptr, (int)W, nCB, pb->cbptr, len, toff,
                         pb->rbseg, pb->perm, pb->pcol, pb->gdst, (float*)nullptr, pb->dvals);
    else if (f32)
      hipLaunchKernelGGL(k_pb_place<float>, dim3(nRB), dim3(256), sm, s, A.rowptr, A.colidx, A.vals, pb->rbptr, (int)W, nCB, pb->cbptr, len, toff,
                         pb->rbseg, pb->perm, pb->pcol, pb->gdst, pb->pvals32, pb->dvals);
    else
      hipLaunchKernelGGL(k_pb_place<double>, dim3(nRB), dim3(256), sm, s, A.rowptr, A.colidx, A.vals, pb->rbptr, (int)W, nCB, pb->cbptr, len, toff,
                         pb->rbseg, pb->perm, pb->pcol, pb->gdst, pb->pvals, pb->dvals);
    chk(hipGetLastError());
    chk(hipStreamSynchronize(s));
  }
  if (e == hipSuccess && groups > 1 && nRB >= 2 * groups) {
    // row-block groups of (nearly) equal segment length
    pb->groups = groups;
    pb->grp_rb.assign((size_t)groups + 1, nRB);
    pb->grp_rb[0] = 0;
    int b = 0;
    for (int g = 1; g < groups; ++g) {
      const int64_t want = np * g / groups;
      while (b < nRB && seg[(size_t)b].x < want) ++b;
      pb->grp_rb[(size_t)g] = b;
    }
    int32_t* grb = nullptr;
    chk(pb_alloc(grb, (size_t)groups + 1));
    chk(pb_alloc(pb->grp_ptr, (size_t)(groups + 1) * nCB));
    if (e == hipSuccess) chk(hipMemcpyAsync(grb, pb->grp_rb.data(), ((size_t)groups + 1) * sizeof(int32_t), hipMemcpyHostToDevice, s));
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_pb_group_ptr, dim3(((groups + 1) * nCB + 255) / 256), dim3(256), 0, s, pb->cbptr, toff, nRB, nCB, groups, grb, pb->grp_ptr);
      chk(hipGetLastError());
      chk(hipStreamSynchronize(s));
    }
    hipFree(grb);
  }
  hipFree(len);
  hipFree(toff);
  hipFree(tot);
  hipFree(segtot);
  if (e == hipSuccess && segmax > kPbNQ * 2 * kPbThreads) e = hipErrorInvalidValue;  // cannot happen: the row blocks were cut to fit
  // Both phases may need more than the default 64 KiB of dynamic LDS.  The limit is a property of the KERNEL, shared by
  // every layout in the process (H and H^T of a two-sided run, a second handle, ...): it is raised once to what the CU
  // has - never to one matrix's need, which a later, smaller layout would lower again under the earlier one's launches.
  if (e == hipSuccess && (pb->lds2 > kPbLdsCuMax || W * sizeof(double) > kPbLdsCuMax)) e = hipErrorInvalidValue;
  if (e == hipSuccess) chk(pb_raise_lds_limits());
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
  const int segcap = pb->segmax;
  auto products = [&](const int32_t* grp_ptr, int g) {
    if (pb->constv)
      hipLaunchKernelGGL((k_pb_products<4, PbConst>), dim3(pb->nCB), dim3(kPbThreads), lds1, s, pb->cbptr, (const PbConst*)nullptr, pb->pcol, pb->gdst, x,
                         A.ncols, pb->W, pb->T2, pb->cval, grp_ptr, g);
    else if (pb->pvals32)
      hipLaunchKernelGGL((k_pb_products<4, float>), dim3(pb->nCB), dim3(kPbThreads), lds1, s, pb->cbptr, pb->pvals32, pb->pcol, pb->gdst, x, A.ncols,
                         pb->W, pb->T2, 0.0, grp_ptr, g);
    else
      hipLaunchKernelGGL((k_pb_products<4, double>), dim3(pb->nCB), dim3(kPbThreads), lds1, s, pb->cbptr, pb->pvals, pb->pcol, pb->gdst, x, A.ncols,
                         pb->W, pb->T2, 0.0, grp_ptr, g);
  };
  auto rows = [&](int rb0, int rb1) {  // one segment fills a CU's LDS: one persistent workgroup per CU
    const int grid = std::min(rb1 - rb0, pb->ncu);
    hipLaunchKernelGGL(k_pb_rows, dim3(grid), dim3(kPbThreads), pb->lds2, s, pb->rbhead, pb->rbseg, A.rowptr, pb->perm, pb->T2, segcap, rb1, x_own, y, part,
                       pb->dvals, rb0);
  };
  if (pb->groups > 1) {  // A/B arm (knob 22): the phases interleaved over groups of row blocks
    for (int g = 0; g < pb->groups; ++g) {
      if (pb->grp_rb[(size_t)g + 1] <= pb->grp_rb[(size_t)g]) continue;
      products(pb->grp_ptr, g);
      rows(pb->grp_rb[(size_t)g], pb->grp_rb[(size_t)g + 1]);
    }
    return pb->nRB;
  }
  products(nullptr, 0);
  rows(0, pb->nRB);
  return pb->nRB;
}

// what the layout moves per SpMV, for the traffic accounting in DESIGN.md / bench.py: {stream entries incl. padding, bytes per
// stream entry of phase 1's input, 1 if the diagonal is split off}
void pb_layout_info(const PbDev* pb, int64_t* np, int* in_bytes_x2, int* diag) {
  *np = pb->np;
  *in_bytes_x2 = (pb->constv ? 0 : pb->pvals32 ? 8 : 16) + 4 + 1;  // twice (value + 2-byte column + 4 bytes of destination per 8 entries)
  *diag = pb->dvals != nullptr;
}

}  // namespace lz
