// Small-problem engine: the whole Lanczos run (Lanczos.py:104-119 + reorthogonalize :247-249) as ONE cooperative kernel.
//
// Why.  The reference's own scripts are small (1Dbox.py: N = 500, 1Ddeuteron.py: N = n = 1001, BASELINE config C1: dense
// 512 x 512, k = 20).  There the six kernels of a Lanczos step do microseconds of work each and the step costs what six
// dependent launches cost (~32 us, profiles/r01/bench_c1_dense_M512_k20.json); a hipGraph would still dispatch every
// node, and has to be captured per (n, allocation).  Here a persistent grid walks all n steps by itself: three grid
// barriers per step, no launch in the loop, scalars never leave the chip.
//
// How a step is split (grid of NB blocks x 256 threads, co-resident by hipLaunchCooperativeKernel):
//   [every block]  alpha_j by the SAME reduction tree as the multi-kernel path, r = (A v_j - alpha_j v_j) - beta v_{j-1}
//                  for ALL positions into the block's own LDS (the vector is at most 10 KB: redundant, but it spares a
//                  barrier), then pass 1 of the re-orthogonalisation: 8-row tiles of the basis dealt round-robin to the
//                  waves of the whole grid, each tile the same v_mfma_f64_4x4x4 sequence as k_qtw_mfma4   -> barrier
//   [positions]    beta = sqrt(r.r), V[j] = 2 r/beta - sum_i c_i V[i]  (NumPy's order, no FMA)            -> barrier
//   [rows]         r = A V[j], per-row products V[j]_i (A V[j])_i                                           -> barrier
// Arithmetic contract: every floating-point operation, and the order of every sum, is that of the multi-kernel path for
// the same input (same partial-sum tree for alpha, same MFMA sequence and lane reduction for c, same element-wise
// expressions) - tests/test_gpu_small.py holds H_eff and V to np.array_equal against it.  That is why the engine is
// limited to what keeps those trees small: one rank, fused-norm mode, full re-orthogonalisation, rows_pad <= 1280 (at most
// three 512-element slices of pass 1, the whole residual in 10 KB of LDS), CSR rows no longer than the CSR-stream tile.
#include "lz_device.h"

namespace lz {

constexpr int kSmallMaxPad = 1280;   // doubles of the residual every block keeps in LDS
constexpr int kSmallMaxParts = 1024; // alpha partials (k_final_sum takes them in one pass of its 1024 threads)

namespace {

// Data that one block writes and OTHER blocks read within the kernel (they may sit on other XCDs, i.e. behind other,
// mutually incoherent L2s):
//   * rewritten every step (SpMV output y, per-row products, raw coefficient sums): written with agent-scope atomic stores
//     (write-through to the memory side) and read with agent-scope atomic loads (served from there, past L1 and L2);
//   * written once (a basis row): atomic stores, then PLAIN loads - no reader can hold a stale copy of a row that did not
//     exist before (rows are 256-byte aligned; the start vector lives in a scratch vector, so row 0 is not read before
//     step 0 writes it).
// With that no cache has to be written back or invalidated at a barrier - the agent-scope release/acquire fences that
// would do it (buffer_wbl2 / buffer_inv over a whole L2) cost ~8 us per barrier here, more than the kernel launches
// the engine replaces.
// LOCAL only changes WHERE the participating blocks run (all on one XCD); the accesses stay device-scope.  Tried and
// measured (profiles/r02/small_engine.json): workgroup-scope atomics are served by the CU's own L1 outside threadgroup-
// split mode (a polling load spins on a stale line: the bounded spin below turned that into a clean fallback, not a
// hang), sc0 buffer loads likewise, L2 atomics used as loads (fetch_add 0) are coherent but serialise (tens of thousands
// per step), release/acquire fences cost ~8 us per barrier.  Device-scope loads and write-through stores are the
// cheapest correct protocol - ~2 us per dependent round trip, wherever the blocks sit.
// COH 0 / 1: the one-kernel engine (plain grid / one XCD), device-scope accesses as described above.  COH 2: the per-step
// kernels (k_small_step): writer and reader are separated by a kernel boundary, plain loads and stores.
template <int COH>
__device__ __forceinline__ double ld_sh(const double* base, int idx) {
  if (COH == 2) return base[idx];
  return __hip_atomic_load(base + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int COH>
__device__ __forceinline__ void st_sh(double* p, double v) {
  if (COH == 2)
    *p = v;
  else
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Monotonic-counter grid barrier over the nb participating blocks.  Every participant calls it the same number of times
// (all loop bounds are grid-uniform).  Each wave first waits for its own write-through stores to be acknowledged.  The
// spin is BOUNDED: if the count is not reached within ~4 M polls the block gives up, flags the run as failed (the host
// then repeats it on the multi-kernel path) and returns false - and so does, one by one, every other block, so the grid
// always drains.
template <bool LOCAL>
__device__ __forceinline__ bool grid_sync(unsigned* bar, unsigned nb, unsigned* status, unsigned zero) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __shared__ int ok_s;
  if (threadIdx.x == 0) {
    unsigned old, target, seen = 0;
    old = __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    target = (old / nb + 1u) * nb;
    int ok = 0;
    for (int it = 0; it < (1 << 22); ++it) {
      seen = __hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + zero;
      if (seen >= target) {
        ok = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (!ok) __hip_atomic_store(status, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ok_s = ok;
  }
  __syncthreads();
  return ok_s != 0;
}

// alpha = V[j] . (A V[j]) from the per-row products, by the partial-sum tree of the multi-kernel path:
//   dense : k_gemv_dense   - blocks of 4 rows, part = (((0 + d0) + d1) + d2) + d3
//   CSR   : k_spmv_stream / k_spmv_fixed - thread t of a row block adds its rows r0 + t + 256 q, block_sum over 4 waves
//   then k_final_sum over the partials (1024 threads = 16 waves, one partial per thread, shuffle tree, 16 sequential adds).
template <int LOCAL>
__device__ double small_alpha(const SmallArgs& a, double* parts, double* sm16) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (a.kind == 2) {
    for (int b = threadIdx.x; b < a.nparts; b += kTPB) {
      double t = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) t += (4 * b + q < a.rows) ? ld_sh<LOCAL>(a.drow, 4 * b + q) : 0.0;
      parts[b] = t;
    }
  } else {
    for (int b = w; b < a.nparts; b += kTPB / 64) {
      const int r0 = a.rowblk ? a.rowblk[b] : b * 512;
      int r1 = a.rowblk ? a.rowblk[b + 1] : (b + 1) * 512;
      if (r1 > a.rows) r1 = a.rows;
      double t = 0.0;
      for (int vw = 0; vw < kTPB / 64; ++vw) {
        double d = 0.0;
        for (int row = r0 + vw * 64 + lane; row < r1; row += kTPB) d += ld_sh<LOCAL>(a.drow, row);
        t += wave_sum(d);  // lane 0: ((0 + s0) + s1) + s2) + s3
      }
      if (lane == 0) parts[b] = t;
    }
  }
  __syncthreads();
  for (int vw = w; vw < 16; vw += kTPB / 64) {
    const int i = vw * 64 + lane;
    const double a0 = i < a.nparts ? 0.0 + parts[i] : 0.0;
    const double s = wave_sum((a0 + 0.0) + (0.0 + 0.0));
    if (lane == 0) sm16[vw] = s;
  }
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) t += sm16[k];
  return t;  // every thread holds alpha
}

// r = A V[j] on the rows dealt to this block, per-row products for alpha
template <int LOCAL>
__device__ void small_spmv(const SmallArgs& a, const double* __restrict__ x, int bid, int nb) {
  const int lane = threadIdx.x & 63;
  if (a.kind == 2) {
    const int gw = bid * (kTPB / 64) + (threadIdx.x >> 6), nw = nb * (kTPB / 64);
    for (int row = gw; row < a.rows; row += nw) {
      const double acc = gemv_row_wave(a.dense + (int64_t)row * a.lda, x, a.rows, lane);
      if (lane == 0) {
        st_sh<LOCAL>(a.y + row, acc);
        st_sh<LOCAL>(a.drow + row, x[row] * acc);
      }
    }
  } else {
    for (int row = bid * kTPB + threadIdx.x; row < a.rows; row += nb * kTPB) {
      double sum = 0.0;
      for (int k = a.rowptr[row]; k < a.rowptr[row + 1]; ++k) sum += a.vals[k] * x[a.colidx[k]];  // SciPy's csr_matvec order
      st_sh<LOCAL>(a.y + row, sum);
      st_sh<LOCAL>(a.drow + row, x[row] * sum);
    }
  }
}

}  // namespace

template <bool LOCAL>
__global__ __launch_bounds__(kTPB) void k_small_run(SmallArgs a) {
  __shared__ double2 sw[kSmallMaxPad / 2 + 64];  // this block's copy of the residual r (all positions; + slack for masked-off steps)
  __shared__ double parts[kSmallMaxParts];
  __shared__ double sm16[16];
  __shared__ double selfw[kTPB / 64];
  __shared__ double pcs[kSmallMaxPad];  // the raw coefficient sums of the current step
  __shared__ int same_s;
  // LOCAL: the grid has 8 x nb blocks and only every eighth takes part - workgroups are dealt to the 8 XCDs round-robin,
  // so those nb blocks share one XCD and its L2.  That is verified before anything depends on it (over device-scope
  // atomics): every participant publishes the XCC_ID it runs on; unless all are equal the run is flagged and left.
  if (LOCAL && (blockIdx.x & 7) != 0) return;
  const int bid = LOCAL ? blockIdx.x >> 3 : blockIdx.x;
  const unsigned nb = LOCAL ? gridDim.x >> 3 : gridDim.x;
  unsigned* bar_dev = a.bar;       // device-scope counter (placement handshake; every barrier when !LOCAL)
  unsigned* bar = LOCAL ? a.bar + 1 : a.bar;
  unsigned* status = a.bar + 2;
  if (LOCAL) {
    if (threadIdx.x == 0) {
      const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;  // HW_REG_XCC_ID[3:0]
      __hip_atomic_store(a.xcc + bid, xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!grid_sync<false>(bar_dev, nb, status, 0u)) return;
    if (threadIdx.x == 0) {
      const unsigned first = __hip_atomic_load(a.xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int same = 1;
      for (unsigned q = 1; q < nb; ++q) same &= __hip_atomic_load(a.xcc + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == first;
      if (!same && bid == 0) __hip_atomic_store(status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      same_s = same;
    }
    __syncthreads();
    if (!same_s) return;  // every participant read the same list: a uniform decision
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n = a.n, cnt2 = a.rows_pad >> 1;
  const int64_t ld2 = a.ldv >> 1;
  double2* V2 = reinterpret_cast<double2*>(a.V);
  // warm-up (Lanczos.py:108-110): r = A v0; alpha0 = r . v0; r = r - alpha0 v0
  small_spmv<LOCAL>(a, a.x0, bid, (int)nb);  // the start vector's scratch copy: basis row 0 itself is not read before step 0 rewrites it
  if (!grid_sync<LOCAL>(bar, nb, status, (unsigned)a.zero)) return;
  double beta_prev = 0.0;
  for (int j = -1; j < n; ++j) {
    // ---- alpha of the vector just multiplied (j == -1: the warm-up's v0), three-term recurrence into LDS
    const int vj = j < 0 ? 0 : j;
    const double al = small_alpha<LOCAL>(a, parts, sm16);
    if (bid == 0 && threadIdx.x == 0) a.alpha[vj] = al;
    if (j == n - 1) break;  // the reference forms one more residual after the last alpha; nothing reads it
    {
      const double2* v2 = j < 0 ? reinterpret_cast<const double2*>(a.x0) : V2 + (int64_t)vj * ld2;
      const double2* m2 = j > 0 ? V2 + (int64_t)(j - 1) * ld2 : nullptr;  // j == 0: the reference's V[-1] is the zero row
      for (int p = threadIdx.x; p < cnt2; p += kTPB) {
        double2 x = make_double2(ld_sh<LOCAL>(a.y, 2 * p), ld_sh<LOCAL>(a.y, 2 * p + 1));
        const double2 v = v2[p];
        x.x = x.x - v.x * al;
        x.y = x.y - v.y * al;
        if (m2) {
          const double2 m = m2[p];
          x.x = x.x - m.x * beta_prev;
          x.y = x.y - m.y * beta_prev;
        }
        sw[p] = x;
      }
    }
    __syncthreads();
    // ---- pass 1 for the NEXT vector (row jn): c_i = V_i . r (i < jn) by 8-row tiles, c_jn = r . r.
    // Grouping of the multi-kernel path at these sizes (plan_qtw: slices of L = 512 elements, one block each; a wave owns a
    // 128-element quarter): per row and slice ((q0 + q1) + q2) + q3 over the four quarters, then the slices in order
    // starting from 0.0 (k_final_rows_t).
    const int jn = j + 1, nrows = jn + 1;
    const int nslices = (a.rows_pad + 511) >> 9;
    double rr = 0.0;
    for (int b = 0; b < nslices; ++b) {
      // r . r of slice b with the per-thread / per-wave grouping of qtw_stage_w<2> (one double2 per thread)
      const int c2 = ((a.rows_pad - 512 * b < 512 ? a.rows_pad - 512 * b : 512)) >> 1;
      double self = 0.0;
      if ((int)threadIdx.x < c2) {
        const double2 v = sw[256 * b + threadIdx.x];
        self = fma(v.x, v.x, self);
        self = fma(v.y, v.y, self);
      }
      self = wave_sum(self);
      __syncthreads();
      if (lane == 0) selfw[w] = self;
      __syncthreads();
      rr = rr + (((selfw[0] + selfw[1]) + selfw[2]) + selfw[3]);
    }
    {
      const int li = lane & 3, blk = (lane >> 2) & 3, lk = lane >> 4;
      const int eoff = 8 * blk + 2 * lk;
      const int gw = bid * (kTPB / 64) + w, nw = (int)nb * (kTPB / 64);
      const int ntiles = jn > 0 ? (nrows + 7) >> 3 : 0;  // step 0: no rows yet (and basis row 0 must not be read, see ld_sh)
      for (int k = gw; k < ntiles; k += nw) {
        const int i0 = 8 * k;
        const double* rowp[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          int i = i0 + 4 * t + li;
          if (i >= nrows) i = nrows - 1;
          if (i == jn) i = jn > 0 ? jn - 1 : (nrows > 1 ? 1 : 0);  // row jn is r itself (not formed yet): its sum is r . r
          rowp[t] = a.V + (int64_t)i * a.ldv;
        }
        double c[2] = {0.0, 0.0};
        for (int b = 0; b < nslices; ++b) {
          double sb[2] = {0.0, 0.0};
          for (int q = 0; q < 4; ++q) {  // the four waves of the multi-kernel block, one quarter each
            const int m_lo = 512 * b + 128 * q;
            int m_hi = m_lo + 128;
            if (m_hi > a.rows_pad) m_hi = a.rows_pad;
            const int nsteps = m_hi > m_lo ? (m_hi - m_lo) >> 5 : 0;
            const double2* swl = sw + ((m_lo + eoff) >> 1);
            double acc[2] = {0.0, 0.0};
            double2 av[2][4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
              for (int t = 0; t < 2; ++t)
                av[t][u] = u < nsteps ? reinterpret_cast<const double2*>(rowp[t] + m_lo + eoff)[16 * u] : make_double2(0.0, 0.0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const double2 bv = u < nsteps ? swl[16 * u] : make_double2(0.0, 0.0);
#pragma unroll
              for (int t = 0; t < 2; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[t][u].x, bv.x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[t][u].y, bv.y, acc[t], 0, 0, 0);
              }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              double v = acc[t];
              v += __shfl_xor(v, 4, 64);
              v += __shfl_xor(v, 8, 64);
              sb[t] = q == 0 ? v : sb[t] + v;
            }
          }
#pragma unroll
          for (int t = 0; t < 2; ++t) c[t] = c[t] + sb[t];
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int row = i0 + 4 * t + lk;
          if ((lane & 15) == 0 && row < nrows && row != jn) st_sh<LOCAL>(a.pc + row, c[t]);
        }
      }
    }
    if (!grid_sync<LOCAL>(bar, nb, status, (unsigned)a.zero)) return;
    // ---- pass 2: beta, V[jn] = 2 r/beta - sum_{i <= jn} c_i V[i]  (k_update_slice<FUSED, raw sums>: NumPy's order)
    const double bnorm = sqrt(rr);
    const int bidx = (jn + n - 2) % (n - 1);  // beta[jn - 1] with Python's negative index at jn = 0
    if (bid == 0 && threadIdx.x == 0) a.beta[bidx] = bnorm;
    for (int k = threadIdx.x; k < jn; k += kTPB) pcs[k] = ld_sh<LOCAL>(a.pc, k);
    __syncthreads();
    {
      const double* sr = reinterpret_cast<const double*>(sw);
      const int e = bid * kTPB + threadIdx.x;  // one ELEMENT per thread: 64 basis rows in flight per lane
      if (e < a.rows_pad) {
        const double wv = sr[e] / bnorm;
        double tx = 0.0;
        constexpr int RU = 64;
        for (int k = 0; k < nrows; k += RU) {
          double q[RU];
#pragma unroll
          for (int u = 0; u < RU; ++u)
            if (k + u < nrows) q[u] = (k + u == jn) ? wv : a.V[(int64_t)(k + u) * a.ldv + e];
#pragma unroll
          for (int u = 0; u < RU; ++u)
            if (k + u < nrows) {
              double ck = (k + u == jn) ? rr : pcs[k + u];
              ck = (k + u == jn) ? ck / (bnorm * bnorm) : ck / bnorm;
              tx = tx + ck * q[u];
            }
        }
        st_sh<LOCAL>(a.V + (int64_t)jn * a.ldv + e, 2.0 * wv - tx);
      }
    }
    beta_prev = bnorm;
    if (!grid_sync<LOCAL>(bar, nb, status, (unsigned)a.zero)) return;
    // ---- r = A V[jn], per-row products
    small_spmv<LOCAL>(a, a.V + (int64_t)jn * a.ldv, bid, (int)nb);
    if (!grid_sync<LOCAL>(bar, nb, status, (unsigned)a.zero)) return;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Per-step kernel: ONE launch per Lanczos step, no grid barrier.  The default for the smallest problems (rows_pad <= 1280,
// n <= 64: BASELINE config C1, 1Dbox.py) - the three-launch path spends 5.2 us per dependent kernel there
// (profiles/r02/launch_bound_probe.txt).  The only data a step needs from OTHER blocks is the SpMV result, so every block
// redoes the vector work of the step for itself - alpha, the three-term recurrence, both re-orthogonalisation passes over
// ALL positions, from its own LDS - then multiplies its share of the rows by the new vector and leaves y and the per-row
// products for the next launch.  Block 0 publishes alpha, beta and the new basis row.  Same arithmetic contract as the
// engine above (same trees, same MFMA sequence, same expressions): bit-identical to the multi-kernel paths.
// MODE 0: r = A x0 only (first launch); 1: a whole step; 2: the last alpha only.
template <int MODE>
__global__ __launch_bounds__(kTPB) void k_small_step(SmallArgs a, int j) {
  __shared__ double2 sw[kSmallMaxPad / 2 + 64];
  __shared__ double vnew[kSmallMaxPad];
  __shared__ double parts[kSmallMaxParts];
  __shared__ double sm16[16];
  __shared__ double selfw[kTPB / 64];
  __shared__ double pcs[kSmallMaxPad];
  const int bid = blockIdx.x, nb = gridDim.x;
  if (MODE == 0) {
    small_spmv<2>(a, a.x0, bid, nb);
    return;
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n = a.n, cnt2 = a.rows_pad >> 1;
  const int64_t ld2 = a.ldv >> 1;
  double2* V2 = reinterpret_cast<double2*>(a.V);
  const int vj = j < 0 ? 0 : j;
  const double al = small_alpha<2>(a, parts, sm16);
  if (bid == 0 && threadIdx.x == 0) a.alpha[vj] = al;
  if (MODE == 2) return;
  // beta of the previous step (the norm that formed V[j]): block 0 of the previous launch stored it
  const double beta_prev = j > 0 ? a.beta[(j + n - 2) % (n - 1)] : 0.0;
  {
    const double2* v2 = j < 0 ? reinterpret_cast<const double2*>(a.x0) : V2 + (int64_t)vj * ld2;
    const double2* m2 = j > 0 ? V2 + (int64_t)(j - 1) * ld2 : nullptr;  // j == 0: the reference's V[-1] is the zero row
    const double2* y2 = reinterpret_cast<const double2*>(a.y);
    for (int p = threadIdx.x; p < cnt2; p += kTPB) {
      double2 x = y2[p];
      const double2 v = v2[p];
      x.x = x.x - v.x * al;
      x.y = x.y - v.y * al;
      if (m2) {
        const double2 m = m2[p];
        x.x = x.x - m.x * beta_prev;
        x.y = x.y - m.y * beta_prev;
      }
      sw[p] = x;
    }
  }
  __syncthreads();
  // pass 1 for row jn: the grouping of the multi-kernel path (see k_small_run); every block does every tile
  const int jn = j + 1, nrows = jn + 1;
  const int nslices = (a.rows_pad + 511) >> 9;
  double rr = 0.0;
  for (int b = 0; b < nslices; ++b) {
    const int c2 = ((a.rows_pad - 512 * b < 512 ? a.rows_pad - 512 * b : 512)) >> 1;
    double self = 0.0;
    if ((int)threadIdx.x < c2) {
      const double2 v = sw[256 * b + threadIdx.x];
      self = fma(v.x, v.x, self);
      self = fma(v.y, v.y, self);
    }
    self = wave_sum(self);
    __syncthreads();
    if (lane == 0) selfw[w] = self;
    __syncthreads();
    rr = rr + (((selfw[0] + selfw[1]) + selfw[2]) + selfw[3]);
  }
  {
    const int li = lane & 3, blk = (lane >> 2) & 3, lk = lane >> 4;
    const int eoff = 8 * blk + 2 * lk;
    const int ntiles = jn > 0 ? (nrows + 7) >> 3 : 0;
    for (int k = w; k < ntiles; k += kTPB / 64) {
      const int i0 = 8 * k;
      const double* rowp[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        int i = i0 + 4 * t + li;
        if (i >= nrows) i = nrows - 1;
        if (i == jn) i = jn > 0 ? jn - 1 : (nrows > 1 ? 1 : 0);  // row jn is r itself (not formed yet): its sum is r . r
        rowp[t] = a.V + (int64_t)i * a.ldv;
      }
      double c[2] = {0.0, 0.0};
      for (int b = 0; b < nslices; ++b) {
        double sb[2] = {0.0, 0.0};
        for (int q = 0; q < 4; ++q) {
          const int m_lo = 512 * b + 128 * q;
          int m_hi = m_lo + 128;
          if (m_hi > a.rows_pad) m_hi = a.rows_pad;
          const int nsteps = m_hi > m_lo ? (m_hi - m_lo) >> 5 : 0;
          const double2* swl = sw + ((m_lo + eoff) >> 1);
          double acc[2] = {0.0, 0.0};
          double2 av[2][4];
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < 2; ++t)
              av[t][u] = u < nsteps ? reinterpret_cast<const double2*>(rowp[t] + m_lo + eoff)[16 * u] : make_double2(0.0, 0.0);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const double2 bv = u < nsteps ? swl[16 * u] : make_double2(0.0, 0.0);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              acc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[t][u].x, bv.x, acc[t], 0, 0, 0);
              acc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[t][u].y, bv.y, acc[t], 0, 0, 0);
            }
          }
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            double v = acc[t];
            v += __shfl_xor(v, 4, 64);
            v += __shfl_xor(v, 8, 64);
            sb[t] = q == 0 ? v : sb[t] + v;
          }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) c[t] = c[t] + sb[t];
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int row = i0 + 4 * t + lk;
        if ((lane & 15) == 0 && row < nrows && row != jn) pcs[row] = c[t];
      }
    }
  }
  __syncthreads();
  // pass 2: beta, V[jn] = 2 r/beta - sum_{i <= jn} c_i V[i] for ALL positions, into LDS (block 0 also publishes the row)
  const double bnorm = sqrt(rr);
  if (bid == 0 && threadIdx.x == 0) a.beta[(jn + n - 2) % (n - 1)] = bnorm;
  {
    const double* sr = reinterpret_cast<const double*>(sw);
    for (int e = threadIdx.x; e < a.rows_pad; e += kTPB) {
      const double wv = sr[e] / bnorm;
      double tx = 0.0;
      constexpr int RU = 64;
      for (int k = 0; k < nrows; k += RU) {
        double q[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u)
          if (k + u < nrows) q[u] = (k + u == jn) ? wv : a.V[(int64_t)(k + u) * a.ldv + e];
#pragma unroll
        for (int u = 0; u < RU; ++u)
          if (k + u < nrows) {
            double ck = (k + u == jn) ? rr : pcs[k + u];
            ck = (k + u == jn) ? ck / (bnorm * bnorm) : ck / bnorm;
            tx = tx + ck * q[u];
          }
      }
      const double vn = 2.0 * wv - tx;
      vnew[e] = vn;
      if (bid == 0) a.V[(int64_t)jn * a.ldv + e] = vn;
    }
  }
  __syncthreads();
  // r = A V[jn] on this block's rows, per-row products for the next launch's alpha
  small_spmv<2>(a, vnew, bid, nb);
}

// participating blocks: one per CU of the one XCD they share (32 CUs; the cooperative grid of 8 x 32 blocks is the most
// this kernel's registers allow) - every phase is a latency chain per wave (a dense row, an 8-row tile, an element's walk
// over the basis), so what counts is the number of waves
int small_grid(int rows_pad) {
  (void)rows_pad;
  return 32;
}

// local: nb participants out of a grid of 8 nb blocks (one XCD); else a grid of nb blocks with device-scope coherence
hipError_t launch_small_run(const SmallArgs& a, int nb, bool local, hipStream_t s) {
  SmallArgs copy = a;
  void* args[] = {&copy};
  if (local) return hipLaunchCooperativeKernel(reinterpret_cast<const void*>(k_small_run<true>), dim3(8 * nb), dim3(kTPB), args, 0, s);
  return hipLaunchCooperativeKernel(reinterpret_cast<const void*>(k_small_run<false>), dim3(nb), dim3(kTPB), args, 0, s);
}

// the per-step kernels: mode 0 (first SpMV), 1 (step j), 2 (last alpha)
hipError_t launch_small_step(const SmallArgs& a, int mode, int j, int nb, hipStream_t s) {
  if (mode == 0)
    hipLaunchKernelGGL(k_small_step<0>, dim3(nb), dim3(kTPB), 0, s, a, j);
  else if (mode == 1)
    hipLaunchKernelGGL(k_small_step<1>, dim3(nb), dim3(kTPB), 0, s, a, j);
  else
    hipLaunchKernelGGL(k_small_step<2>, dim3(1), dim3(kTPB), 0, s, a, j);
  return hipGetLastError();
}

}  // namespace lz
