// Hand-written gfx950 vector kernels of the Lanczos hot path: full re-orthogonalisation (Q^T w, update),
// three-term recurrence, second-stage reductions, halo pack.
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
#include <algorithm>
#include <cmath>

#include "lz_device.h"

namespace lz {

// ------------------------------------------------------------------ second-stage reductions
constexpr int kFinalThreads = 1024;
// sum(part[0..n)) by one block of 1024 threads: four strided accumulators per thread (a full group of four while
// i + 3 * 1024 < n, the remainder into the first), wave shuffle tree, 16 wave sums added in order.  The grouping is part of
// the contract (final_sum_emulated in lz_device.h replays it).  The result is valid in thread 0.
// (Round 4 tried issuing all of a thread's loads - up to 24 - before its first add, on the theory that the ~4.7 us this
// one-block kernel takes are eight dependent round trips to the memory side: measured 4.85 us instead of 4.69, k_omega 6.0
// instead of 5.0 (profiles/r04/partial_loop_kernel_stats*.csv) - a one-block kernel this short pays for its instruction
// fetch, not for its loads; the plain loop is the smaller code and stays.)
__device__ __forceinline__ double final_sum_1024(const double* __restrict__ part, int n, double* sm) {
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int i = threadIdx.x;
  for (; i + 3 * kFinalThreads < n; i += 4 * kFinalThreads) {
    a0 += part[i];
    a1 += part[i + kFinalThreads];
    a2 += part[i + 2 * kFinalThreads];
    a3 += part[i + 3 * kFinalThreads];
  }
  for (; i < n; i += kFinalThreads) a0 += part[i];
  const double acc = wave_sum((a0 + a1) + (a2 + a3));
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) sm[w] = acc;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < kFinalThreads / 64; ++k) t += sm[k];
  }
  return t;
}

__global__ __launch_bounds__(kFinalThreads) void k_final_sum(const double* __restrict__ part, int n, double* __restrict__ out) {
  __shared__ double sm[kFinalThreads / 64];
  const double t = final_sum_1024(part, n, sm);
  if (threadIdx.x == 0) out[0] = t;
}

// ------------------------------------------------------------------ partial re-orthogonalisation: the decision, on the device
// Simon's omega-recurrence (H. D. Simon 1984): omega_{j,k} estimates v_j . v_k from alpha and beta alone; a sweep is due when
// max_k |omega_{j,k}| exceeds sqrt(eps), on that vector and the next.  One block; called once per step, AFTER the three-term
// kernel of step jn - 1 (optionally folding that kernel's ||r||^2 partials first, with k_final_sum's exact grouping), it leaves
// the gate of step jn in ist[0].  The sweep kernels and the SpMV of step jn read the gate: lz_run never waits for the device.
// Arithmetic: expression for expression the host loop this replaces (run_loop_six, lz_loops.hip; both compiled with
// -ffp-contract=off, IEEE sqrt and division), every k is independent and the maximum does not depend on the order, so the
// decisions - and with them every coefficient - are bit-identical to the host-decided run (tests/test_gpu_lanczos.py).
// State layout: see omega_state_doubles (lz_internal.h).
// c_clear (row-block partition): the coefficient buffer the host all-reduces EVERY step (it cannot skip a collective the device may
// need) is zeroed when the coming step does not sweep - stale coefficients would otherwise be summed over the ranks again and again
// (x world per step: inf after a few hundred steps; never read, but a trap for any later consumer - ADVICE r4).
__global__ __launch_bounds__(kFinalThreads) void k_omega(const double* __restrict__ part, int np, double* __restrict__ nrm2,
                                                        const double* __restrict__ alpha, int jn, int n, double* __restrict__ st,
                                                        int* __restrict__ ist, double* __restrict__ c_clear) {
  __shared__ double sm[kFinalThreads / 64];
  __shared__ double s_nrm2;
  __shared__ int s_sweep;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (part) {  // k_final_sum's own code
    const double t = final_sum_1024(part, np, sm);
    if (threadIdx.x == 0) {
      nrm2[0] = t;
      s_nrm2 = t;
    }
  } else if (threadIdx.x == 0) {
    s_nrm2 = nrm2[0];
  }
  __syncthreads();
  const double eps = 2.220446049250313e-16, thresh = 1.4901161193847656e-08;
  double* hb = st + 2;
  double* W = hb + (n + 2);
  const int ldw = n + 1;
  const double hbj = sqrt(s_nrm2);
  if (jn == 0) {  // after the warm-up: clear the state; step 0 always sweeps (a one-row sweep)
    for (int i = threadIdx.x; i < 3 * ldw; i += kFinalThreads) W[i] = i == 0 ? 1.0 : 0.0;  // omega_{0,0} = v_0 . v_0
    for (int i = threadIdx.x; i < n + 2; i += kFinalThreads) hb[i] = i == 0 ? hbj : 0.0;
    if (threadIdx.x == 0) {
      st[0] = 0.0;
      st[1] = 0.0;
      ist[0] = 1;
      ist[1] = 1;
      ist[2] = 1;
    }
    return;
  }
  // beta_j omega_{j,k} = beta_{k+1} omega_{j-1,k+1} + (alpha_k - alpha_{j-1}) omega_{j-1,k} + beta_k omega_{j-1,k-1}
  //                      - beta_{j-1} omega_{j-2,k}  (+ rounding of size eps ||A||),   k <= j-2
  const double a_last = alpha[jn - 1];
  const double hb_prev = hb[jn - 1];
  double normA = st[0];
  {
    const double cand = fabs(a_last) + hb_prev + hbj;
    if (cand > normA) normA = cand;
  }
  const double* cur = W + (size_t)((jn + 2) % 3) * ldw;   // omega_{jn-1,:}
  const double* prev = W + (size_t)((jn + 1) % 3) * ldw;  // omega_{jn-2,:}
  double* nw = W + (size_t)(jn % 3) * ldw;
  double worst = 0.0;
  for (int k = threadIdx.x; k <= n; k += kFinalThreads) {
    double v = 0.0;
    if (k == jn) {
      v = 1.0;
    } else if (k == jn - 1) {
      v = eps;
    } else if (k + 2 <= jn) {
      double t = hb[k + 1] * cur[k + 1] + (alpha[k] - a_last) * cur[k] - hb_prev * prev[k];
      if (k > 0) t += hb[k] * cur[k - 1];
      t += (t < 0 ? -1.0 : 1.0) * 2.0 * eps * normA;
      v = t / hbj;
      const double av = fabs(v);
      if (av > worst) worst = av;
    }
    nw[k] = v;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_down(worst, off, 64);
    if (o > worst) worst = o;
  }
  __syncthreads();  // (sm is reused)
  if (lane == 0) sm[w] = worst;
  __syncthreads();
  if (threadIdx.x == 0) {
    double m = 0.0;
#pragma unroll
    for (int k = 0; k < kFinalThreads / 64; ++k)
      if (sm[k] > m) m = sm[k];
    const bool due = m > thresh;
    const int sweep = (due || st[1] != 0.0) ? 1 : 0;  // a due sweep also covers the next vector (both feed the recurrence)
    st[1] = due ? 1.0 : 0.0;
    st[0] = normA;
    hb[jn] = hbj;
    ist[0] = sweep;
    ist[1] += sweep;
    ist[2 + jn] = sweep;
    s_sweep = sweep;
  }
  __syncthreads();
  if (s_sweep)
    for (int k = threadIdx.x; k < jn; k += kFinalThreads) nw[k] = eps;
  else if (c_clear)
    for (int k = threadIdx.x; k <= jn; k += kFinalThreads) c_clear[k] = 0.0;
}
void launch_omega(const double* part, int np, double* nrm2, const double* alpha, int jn, int n, double* st, int* ist, hipStream_t s, double* c_clear) {
  hipLaunchKernelGGL(k_omega, dim3(1), dim3(kFinalThreads), 0, s, part, np, nrm2, alpha, jn, n, st, ist, c_clear);
}

// ---- the same decision for the loop with ONE all-reduce per step (LZ_FLAG_REORTH_PARTIAL | LZ_FLAG_ONE_REDUCE) --------------
// On a row-block partition the sums a step needs - alpha of the newest vector u = v_{j-1}, the three self terms that give
// ||r||^2, and (when a sweep runs) the basis dots against r'' and u - travel in ONE buffer and one all-reduce (run_loop_partial_
// onereduce, lz_loops.hip).  The sweep's dots must be in that buffer BEFORE alpha_{j-1} and beta_j - the inputs of omega_{j,:} -
// exist on any rank, so the gate of step j cannot be Simon's exact test of omega_{j,:}; it is a ONE-STEP LOOK-AHEAD, taken by this
// kernel right after the all-reduce of step j - 1:
//   * omega_{j,:} is advanced exactly (k_omega's recurrence, expression for expression) once alpha_{j-1}, beta_j are known;
//   * omega_{j+1,:} is PREDICTED with the same recurrence, the two coefficients that do not exist yet replaced by the newest
//     ones (alpha_j ~ alpha_{j-1}, beta_{j+1} ~ beta_j); a sweep of v_{j+1} (and of v_{j+2}: Simon's pair) is due when
//     kappa * max_k |pred| > sqrt(eps) (kappa = 4 by default: a sweep comes a step early rather than late - an early sweep costs
//     nothing but its place in the schedule);
//   * the exact row is the check on the prediction: a vector that was not swept although its exact omega exceeds sqrt(eps) is a
//     MISS - counted (lz_last_sweep_misses), and the next two vectors are swept.
// Every input is an all-reduced sum or derived from one: every rank takes the same decisions.  One block; it also finishes the
// all-reduced buffer like k_onereduce_prepare (alpha, c_i = p_i - alpha q_i, ||r||^2 by the three-sum form + its cancellation
// guard) and clears the buffer of the next step (coefficient slots of a step without a sweep must be zero, not stale: they are
// summed over the ranks every step).
// State: st / W as k_omega; ist[0], ist[1] = gate of the even / odd steps, ist[2] = sweeps, ist[3] = misses, ist[4 + j] = log.
__global__ __launch_bounds__(kFinalThreads) void k_partial_onered_post(double* __restrict__ buf, double* __restrict__ bufn, int nzero_next, int m,
                                                                       int ldp, double* __restrict__ alpha_slot, double* __restrict__ nrm2,
                                                                       const double* __restrict__ alpha, int j, int n, double* __restrict__ st,
                                                                       int* __restrict__ ist, double kappa) {
  __shared__ double sm[kFinalThreads / 64];
  __shared__ double s_v;
  __shared__ int s_flag;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (j < 0) {  // before the warm-up: step 0 always sweeps (the reference's one-row sweep)
    for (int i = threadIdx.x; i < n + 5; i += kFinalThreads) ist[i] = (i == 0 || i == 2 || i == 4) ? 1 : 0;
    return;
  }
  const int g = ist[j & 1];
  const double a = buf[ldp + m + 2];
  if (bufn)
    for (int i = threadIdx.x; i < nzero_next; i += kFinalThreads) bufn[i] = 0.0;
  if (g)
    for (int i = threadIdx.x; i < m; i += kFinalThreads) buf[i] = buf[i] - a * buf[ldp + i];
  if (threadIdx.x == 0) {
    const double rr = buf[m], uu = buf[ldp + m], ur = buf[ldp + m + 1];
    const double v = (rr - 2.0 * a * ur) + a * a * uu;
    if (!(v > 1e-4 * rr)) nrm2[1] = 1.0;  // cancellation guard, as k_onereduce_prepare
    buf[m] = v;
    nrm2[0] = v;
    alpha_slot[0] = a;
    s_v = v;
  }
  __syncthreads();
  const double eps = 2.220446049250313e-16, thresh = 1.4901161193847656e-08;
  double* hb = st + 2;
  double* W = hb + (n + 2);
  const int ldw = n + 1;
  const double hbj = sqrt(s_v);
  if (j == 0) {
    for (int i = threadIdx.x; i < 3 * ldw; i += kFinalThreads) W[i] = i == 0 ? 1.0 : 0.0;  // omega_{0,0} = v_0 . v_0
    for (int i = threadIdx.x; i < n + 2; i += kFinalThreads) hb[i] = i == 0 ? hbj : 0.0;
    if (threadIdx.x == 0) {
      st[0] = 0.0;
      st[1] = 0.0;
      ist[1] = 0;  // step 1: nothing to be orthogonal to but v_0, which the recurrence itself takes care of
      if (n > 1) ist[4 + 1] = 0;
    }
    return;
  }
  // exact row j (k_omega's code): alpha_{j-1} = a, beta_j = hbj
  const double hb_prev = hb[j - 1];
  double normA = st[0];
  {
    const double cand = fabs(a) + hb_prev + hbj;
    if (cand > normA) normA = cand;
  }
  const double* cur = W + (size_t)((j + 2) % 3) * ldw;   // omega_{j-1,:}
  const double* prev = W + (size_t)((j + 1) % 3) * ldw;  // omega_{j-2,:}
  double* nw = W + (size_t)(j % 3) * ldw;
  double worst = 0.0;
  for (int k = threadIdx.x; k <= n; k += kFinalThreads) {
    double v = 0.0;
    if (k == j) {
      v = 1.0;
    } else if (k == j - 1) {
      v = eps;
    } else if (k + 2 <= j) {
      const double ak = k == j - 1 ? a : alpha[k];
      double t = hb[k + 1] * cur[k + 1] + (ak - a) * cur[k] - hb_prev * prev[k];
      if (k > 0) t += hb[k] * cur[k - 1];
      t += (t < 0 ? -1.0 : 1.0) * 2.0 * eps * normA;
      v = t / hbj;
      const double av = fabs(v);
      if (av > worst) worst = av;
    }
    if (g && k < j) v = eps;  // this vector is being swept
    nw[k] = v;
  }
  auto block_max = [&](double x) -> double {  // valid in thread 0
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const double o = __shfl_down(x, off, 64);
      if (o > x) x = o;
    }
    __syncthreads();
    if (lane == 0) sm[w] = x;
    __syncthreads();
    double mx = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
      for (int k = 0; k < kFinalThreads / 64; ++k)
        if (sm[k] > mx) mx = sm[k];
    }
    return mx;
  };
  const double worst_exact = block_max(worst);
  if (threadIdx.x == 0) s_flag = (!g && worst_exact > thresh) ? 1 : 0;
  __syncthreads();  // (also: row j complete before the prediction reads it)
  const int miss = s_flag;
  // predicted row j + 1: k <= j - 1; unknown alpha_j ~ a, beta_{j+1} ~ hbj; beta_j = hbj exact; hb[j] is not stored yet
  double worst_p = 0.0;
  if (j + 1 < n) {
    for (int k = threadIdx.x; k + 2 <= j + 1; k += kFinalThreads) {
      const double ak = k == j - 1 ? a : alpha[k];
      const double hk1 = k + 1 == j ? hbj : hb[k + 1];
      double t = hk1 * nw[k + 1] + (ak - a) * nw[k] - hbj * cur[k];
      if (k > 0) t += hb[k] * nw[k - 1];
      t += (t < 0 ? -1.0 : 1.0) * 2.0 * eps * normA;
      const double av = fabs(t / hbj);
      if (av > worst_p) worst_p = av;
    }
  }
  const double wp = block_max(worst_p);
  if (threadIdx.x == 0) {
    const bool due = kappa * wp > thresh || miss;
    const int gn = (due || st[1] != 0.0) ? 1 : 0;
    st[1] = (due && !g) ? 1.0 : 0.0;  // Simon's pair: the vector after a newly due one is swept too
    st[0] = normA;
    hb[j] = hbj;
    ist[3] += miss;
    if (j + 1 < n) {
      ist[(j + 1) & 1] = gn;
      ist[2] += gn;
      ist[4 + j + 1] = gn;
    }
  }
}
void launch_partial_onered_post(double* buf, double* bufn, int nzero_next, int m, int ldp, double* alpha_slot, double* nrm2, const double* alpha,
                                int j, int n, double* st, int* ist, double kappa, hipStream_t s) {
  hipLaunchKernelGGL(k_partial_onered_post, dim3(1), dim3(kFinalThreads), 0, s, buf, bufn, nzero_next, m, ldp, alpha_slot, nrm2, alpha, j, n, st, ist,
                     kappa);
}

// up to four independent k_final_sum's in one launch (block q adds its own run into its own slot)
__global__ __launch_bounds__(kFinalThreads) void k_final_sum_multi(FinalMulti fm) {
  __shared__ double sm[kFinalThreads / 64];
  const double t = final_sum_1024(fm.part[blockIdx.x], fm.n[blockIdx.x], sm);
  if (threadIdx.x == 0) fm.out[blockIdx.x][0] = t;
}
void launch_final_sum_multi(const FinalMulti& fm, int count, hipStream_t s) {
  hipLaunchKernelGGL(k_final_sum_multi, dim3(count), dim3(kFinalThreads), 0, s, fm);
}

__global__ __launch_bounds__(kTPB) void k_final_rows(const double* __restrict__ part, int G, double* __restrict__ c) {
  __shared__ double sm[kTPB / 64];
  const double* p = part + (int64_t)blockIdx.x * G;
  double acc = 0.0;
  for (int i = threadIdx.x; i < G; i += kTPB) acc += p[i];
  acc = block_sum(acc, sm);
  if (threadIdx.x == 0) c[blockIdx.x] = acc;
}

// Transposed partial layout of the 4x4x4 MFMA kernel: part[pid * ldp + row].  One block owns 8 rows (one 64-byte
// sector of every pid's run): 128 pid lanes x 8 row lanes, fixed summation order.
__global__ __launch_bounds__(kFinalThreads) void k_final_rows_t(const double* __restrict__ part, int P, int ldp, int nrows,
                                                              double* __restrict__ c, const int* __restrict__ gate) {
  if (gate && gate[0] == 0) return;  // device-resident partial re-orthogonalisation: no sweep on this vector
  __shared__ double sm[kFinalThreads / 8][8];
  const int rl = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int row = blockIdx.x * 8 + rl;  // < ldp (a multiple of 8); rows >= nrows hold stale values and are never stored
  const double* p = part + row;
  constexpr int S = kFinalThreads / 8;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int pid = pl;
  for (; pid + 3 * S < P; pid += 4 * S) {
    a0 += p[(int64_t)pid * ldp];
    a1 += p[(int64_t)(pid + S) * ldp];
    a2 += p[(int64_t)(pid + 2 * S) * ldp];
    a3 += p[(int64_t)(pid + 3 * S) * ldp];
  }
  for (; pid < P; pid += S) a0 += p[(int64_t)pid * ldp];
  sm[pl][rl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (threadIdx.x < 8 && row < nrows) {
    double t = 0.0;
    for (int k = 0; k < S; ++k) t += sm[k][threadIdx.x];
    c[row] = t;
  }
}

void launch_final_sum(const double* part, int n, double* out, hipStream_t s) {
  hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(kFinalThreads), 0, s, part, n, out);
}
// explicit run length: c[i] = sum_b part[b * ldp + i], i < nout (one-reduce mode: two coefficient runs per block)
void launch_final_rows_t(const double* part, int G, int ldp, int nout, double* c, hipStream_t s, const int* gate) {
  if (nout <= 0) return;
  hipLaunchKernelGGL(k_final_rows_t, dim3((nout + 7) / 8), dim3(kFinalThreads), 0, s, part, G, ldp, nout, c, gate);
}
void launch_final_rows(const double* part, int nrows, int G, double* c, hipStream_t s, bool transposed, const int* gate) {
  if (nrows <= 0) return;
  if (transposed)
    hipLaunchKernelGGL(k_final_rows_t, dim3((nrows + 7) / 8), dim3(kFinalThreads), 0, s, part, G, qtw_ldp(nrows), nrows, c, gate);
  else
    hipLaunchKernelGGL(k_final_rows, dim3(nrows), dim3(kTPB), 0, s, part, G, c);
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
      st_stream<1>(vj + t, v);  // next read of V[j] is a whole pass away: keep it out of L2
      sw[t] = v;
      self = fma(v.x, v.x, self);
      self = fma(v.y, v.y, self);
    }
  } else {
    const double2* src = SCALE == 2 ? reinterpret_cast<const double2*>(r + base) : vj;
    // five positions per trip (a full 5120-slice is two trips): all loads of a trip are issued before the first use -
    // at launch every resident block stages at once and nothing else hides the latency
    for (int t0 = threadIdx.x; t0 < cnt2; t0 += 5 * kTPB) {
      double2 v[5];
#pragma unroll
      for (int u = 0; u < 5; ++u) v[u] = (t0 + u * kTPB < cnt2) ? src[t0 + u * kTPB] : make_double2(0.0, 0.0);
#pragma unroll
      for (int u = 0; u < 5; ++u)
        if (t0 + u * kTPB < cnt2) {
          sw[t0 + u * kTPB] = v[u];
          self = fma(v[u].x, v[u].x, self);
          self = fma(v[u].y, v[u].y, self);
        }
    }
  }
  return self;
}

// One-reduce mode (SCALE == 3): two columns are staged - the two-term residual r'' = A u - beta v_{j-2} (sw) and the newest
// basis vector u = v_{j-1} (su) - and their three mutual dots come back: self[0] = r''.r'', self[1] = u.r'', self[2] = u.u.
__device__ __forceinline__ void qtw_stage_two(const double* __restrict__ r, const double* __restrict__ u, int64_t base, int cnt2,
                                              double2* sw, double2* su, double (&self)[3]) {
  const double2* rr = reinterpret_cast<const double2*>(r + base);
  const double2* uu = reinterpret_cast<const double2*>(u + base);
  self[0] = self[1] = self[2] = 0.0;
  for (int t0 = threadIdx.x; t0 < cnt2; t0 += 4 * kTPB) {
    double2 a[4], b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool ok = t0 + q * kTPB < cnt2;
      a[q] = ok ? rr[t0 + q * kTPB] : make_double2(0.0, 0.0);
      b[q] = ok ? uu[t0 + q * kTPB] : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (t0 + q * kTPB < cnt2) {
        sw[t0 + q * kTPB] = a[q];
        su[t0 + q * kTPB] = b[q];
        self[0] = fma(a[q].x, a[q].x, self[0]);
        self[0] = fma(a[q].y, a[q].y, self[0]);
        self[1] = fma(a[q].x, b[q].x, self[1]);
        self[1] = fma(a[q].y, b[q].y, self[1]);
        self[2] = fma(b[q].x, b[q].x, self[2]);
        self[2] = fma(b[q].y, b[q].y, self[2]);
      }
  }
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
  const int i_top = ((nrows - 1) / R) * R;
  for (int i0 = i_top; i0 >= 0; i0 -= R) {
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

#ifdef LZ_KBENCH  // retired A/B arm (LZ_FLAG_QTW_MFMA, the 16x16x4 tiling: 20 % slower): its kernel lives in a kernel-bench-only file
#include "lz_reorth_kbench.h"
#endif

// MFMA variant 2: v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 blocks per instruction).  Lane layout measured on
// gfx950 (tools/probes/mfma_f64_4x4x4_layout.hip): A[blk][i][k] lives in lane 16k + 4blk + i, B[blk][k][j] in lane
// 16k + 4blk + j, D[blk][i][j] in lane 16i + 4blk + j.  Here the 4 blocks are four adjacent 64-byte chunks of the SAME
// four basis rows, so one wave-load covers 4 rows x 256 contiguous bytes (whole 128-byte lines, like the VALU kernel)
// instead of the 16 rows x 64 bytes the 16x16x4 shape forces.  B = the matching w entries broadcast over j; the four
// block results of a row are added with two cross-lane steps once per tile.  No barrier in the main loop.
template <int SCALE, int U, int T>
__global__ __launch_bounds__(kTPB) void k_qtw_mfma4(double* __restrict__ V, int64_t ldv, int64_t len, int nrows, int j,
                                                   const double* __restrict__ r, const double* __restrict__ nrm2,
                                                   double* __restrict__ beta_slot, int64_t L, int ldp,
                                                   double* __restrict__ part, QtwFuse fz) {
  if (fz.gate && fz.gate[0] == 0) return;  // device-resident partial re-orthogonalisation: no sweep on this vector
  extern __shared__ double2 sw[];
  // Per-wave coefficients are parked in LDS (after the slice of w) and written when the wave is done, as one contiguous
  // run part[pid][0..nrows): 8-byte partial stores trickling into the read stream cost 6 % of the pass
  // (profiles/r01/ab_qtw_tile_epilogue.json: 1258 -> 1186 us without them), full lines at the end of a block's life do not.
  // SCALE == 3 (one-reduce mode): two LDS slices - r'' (argument `r`) and u = basis row `j` - and coefficient runs twice
  // as long: [0, ldp) the dots with r'', [ldp, 2 ldp) the dots with u.  Rows [0, nrows) are ALL streamed (u's own row
  // included, nothing is skipped); nrows may be 0 (first step: only the three self terms come back).
  constexpr int NCOL = SCALE == 3 ? 2 : 1;
  double* keep = reinterpret_cast<double*>(sw) + NCOL * L + (threadIdx.x >> 6) * (NCOL * ldp);
  const int64_t base = (int64_t)blockIdx.x * L;
  const int cnt = (int)(len - base < L ? len - base : L);
  double self = 0.0, self3[3] = {0.0, 0.0, 0.0};
  if constexpr (SCALE == 3) {
    qtw_stage_two(r, V + (int64_t)j * ldv, base, cnt >> 1, sw, sw + (L >> 1), self3);
#pragma unroll
    for (int q = 0; q < 3; ++q) self3[q] = wave_sum(self3[q]);
  } else if constexpr (SCALE == 4) {
    // fused small-problem mode: finish the previous step here (see QtwFuse), then stage and dot r like SCALE == 2
    // np > 0 (small problems): alpha is added up here from the SpMV's block partials; np == 0 (any size): alpha has been
    // reduced (and all-reduced) already and is read back from its slot
    __shared__ double sm16[16];
    double al;
    if (fz.np > 0) {
      al = final_sum_emulated(fz.apart, fz.np, sm16);
      if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) fz.alpha_out[0] = al;
    } else {
      al = fz.alpha_out[0];
    }
    const double be = fz.jprev2 >= 0 ? fz.beta_prev[0] : 0.0;
    const double2* y2 = reinterpret_cast<const double2*>(r + base);
    double2* ro2 = reinterpret_cast<double2*>(fz.r_out + base);
    const double2* v2 = reinterpret_cast<const double2*>(V + (int64_t)fz.jprev * ldv + base);
    const double2* m2 = fz.jprev2 >= 0 ? reinterpret_cast<const double2*>(V + (int64_t)fz.jprev2 * ldv + base) : nullptr;
    const int cnt2 = cnt >> 1;
    for (int t0 = threadIdx.x; t0 < cnt2; t0 += 5 * kTPB) {
      double2 x[5];
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const int t = t0 + u * kTPB;
        x[u] = make_double2(0.0, 0.0);
        if (t < cnt2) {
          x[u] = y2[t];
          const double2 v = v2[t];
          x[u].x = x[u].x - v.x * al;  // k_three_term's expression, element for element
          x[u].y = x[u].y - v.y * al;
          if (m2) {
            const double2 m = m2[t];
            x[u].x = x[u].x - m.x * be;
            x[u].y = x[u].y - m.y * be;
          }
          if (blockIdx.y == 0) ro2[t] = x[u];  // (row split: every block of the slice computes it, one stores it)
        }
      }
#pragma unroll
      for (int u = 0; u < 5; ++u)
        if (t0 + u * kTPB < cnt2) {
          sw[t0 + u * kTPB] = x[u];
          self = fma(x[u].x, x[u].x, self);
          self = fma(x[u].y, x[u].y, self);
        }
    }
    self = wave_sum(self);
  } else {
    self = qtw_stage_w<SCALE>(V, ldv, j, r, nrm2, beta_slot, base, cnt >> 1, sw);
    self = wave_sum(self);  // this wave's share of w.w (lane 0); replaces the streamed row-j result below
  }
  const int jskip = SCALE == 3 ? -1 : j;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int li = lane & 3, blk = (lane >> 2) & 3, lk = lane >> 4;
  const int sub = (int)(L >> 2);  // elements per wave (multiple of 128)
  const int m_lo = w * sub;
  int m_hi = m_lo + sub;
  if (m_hi > cnt) m_hi = cnt;
  const int nsteps = m_hi > m_lo ? (m_hi - m_lo) >> 5 : 0;  // 32 elements (256 B per row) per step
  const int eoff = 8 * blk + 2 * lk;                          // this lane's double2 within a step
  // B operand: column jj = lane & 3 of each 4x4 block; one-reduce mode gives the odd columns the second vector
  const double2* swl = sw + ((m_lo + eoff) >> 1) + ((SCALE == 3 && (lane & 1)) ? (L >> 1) : 0);  // + 16 per step
  const int i_top = nrows > 0 ? ((nrows - 1) / (4 * T)) * (4 * T) : 0;
  // Row pointers of tile k (tiles run from the newest rows down).
  auto tile_rows = [&](int i0, const double2* (&a)[T]) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      int i = i0 + 4 * t + li;
      if (i >= nrows) i = nrows - 1;  // clamped duplicate, discarded at the store
      if (i == jskip) i = j > 0 ? j - 1 : (nrows > 1 ? 1 : 0);  // row j is w itself: its coefficient comes from `self`
      a[t] = reinterpret_cast<const double2*>(V + (int64_t)i * ldv + base + m_lo + eoff);
    }
  };
  auto load_batch = [&](const double2* const (&a)[T], int s0, double2 (&av)[T][U]) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < T; ++t) av[t][u] = (s0 + u < nsteps) ? ld_stream<1>(a[t] + 16 * (s0 + u)) : make_double2(0.0, 0.0);
  };
  auto mma_batch = [&](int s0, const double2 (&av)[T][U], double (&acc)[T]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const double2 bv = (s0 + u < nsteps) ? swl[16 * (s0 + u)] : make_double2(0.0, 0.0);
#pragma unroll
      for (int t = 0; t < T; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[t][u].x, bv.x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[t][u].y, bv.y, acc[t], 0, 0, 0);
      }
    }
  };
  // The first batch of loads of the NEXT tile is issued before the current tile's results are reduced and stored: the
  // end of a tile otherwise drains the wave's whole load queue (s_waitcnt 0 before the cross-lane adds), a bubble that
  // cost 6 % of the pass (profiles/r01/ab_qtw_tile_epilogue.json).
  const int ntiles = nrows > 0 ? i_top / (4 * T) + 1 : 0;  // one-reduce mode at j = 0: no rows yet, only the self terms
  // Row split (gridDim.y > 1, short vectors with many basis rows - 1Ddeuteron.py: M = n = 1001): the blocks
  // (blockIdx.x, 0..Y-1) share slice blockIdx.x and deal its row tiles round-robin; every (row, slice, quarter) dot is
  // still one wave's same MFMA sequence and every block writes exactly its own rows of the slice's run: same bits, Y
  // times the parallelism.
  const int ky = blockIdx.y, kstep = gridDim.y;
  const double2* a[T];
  double2 av0[T][U];
  if (ky < ntiles) {
    tile_rows(i_top - ky * 4 * T, a);
    load_batch(a, 0, av0);
  }
  for (int k = ky; k < ntiles; k += kstep) {
    const int i0 = i_top - k * 4 * T;
    double acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = 0.0;
    mma_batch(0, av0, acc);
    for (int s0 = U; s0 < nsteps; s0 += U) {
      double2 av[T][U];
      load_batch(a, s0, av);
      mma_batch(s0, av, acc);
    }
    if (k + kstep < ntiles) {
      tile_rows(i_top - (k + kstep) * 4 * T, a);
      load_batch(a, 0, av0);
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
      double v = acc[t];                 // D[blk][i][jj] in lane 16 i + 4 blk + jj
      v += __shfl_xor(v, 4, 64);         // add the four blocks (adjacent 64-byte chunks)
      v += __shfl_xor(v, 8, 64);
      const int row = i0 + 4 * t + lk;   // lane 16 i (+0) holds row i
      if ((lane & 15) == 0 && row < nrows) keep[row] = v;
      if (SCALE == 3 && (lane & 15) == 1 && row < nrows) keep[ldp + row] = v;  // column 1: dots with u
    }
  }
  if (SCALE == 3) {
    if (lane == 0) {  // slots behind the streamed rows: [nrows] = r''.r'', [ldp + nrows] = u.u, [ldp + nrows + 1] = u.r''
      keep[nrows] = self3[0];
      keep[ldp + nrows] = self3[2];
      keep[ldp + nrows + 1] = self3[1];
    }
  } else if (lane == 0 && j < nrows) {
    keep[j] = self;  // c_j = w.w from the LDS-resident values (row j itself is not streamed)
  }
  __syncthreads();  // the only barrier after staging: the four waves' runs are added (fixed order) and leave as one
  const int run = NCOL * ldp;
  const int nout = SCALE == 3 ? ldp + nrows + 2 : nrows;
  const double* k0 = reinterpret_cast<const double*>(sw) + NCOL * L;
  double* mine = part + (int64_t)blockIdx.x * run;
  const int t_top = i_top / (4 * T);
  for (int i = threadIdx.x; i < nout; i += kTPB) {
    if (kstep > 1) {  // row split: this block owns the tiles k == ky (mod Y) counted from the top, and block 0 row j's self term
      const bool mine_row = i == j ? ky == 0 : (t_top - i / (4 * T)) % kstep == ky;
      if (!mine_row) continue;
    }
    __builtin_nontemporal_store(((k0[i] + k0[run + i]) + k0[2 * run + i]) + k0[3 * run + i], mine + i);
  }
  // Device-resident partial re-orthogonalisation (fz.ticket != nullptr; SCALE == 1 or 3, no row split): the block that finishes LAST
  // adds up every block's run itself - k_final_rows_t's additions in k_final_rows_t's order, so c has the same bits - which
  // saves that kernel's launch in the 99 % of steps where this kernel returns at its first line.  The release / acquire
  // pair around the ticket costs more than the launch it replaces (the two-sided links measured 2-4x), but only a step that
  // really sweeps pays it.
  if constexpr (SCALE == 1 || SCALE == 3) {
    if (fz.ticket != nullptr) {
      __shared__ unsigned s_last;
      __threadfence();  // this block's run is visible device-wide before its ticket is
      __syncthreads();  // (and every wave is done with the LDS image)
      if (threadIdx.x == 0) s_last = atomicAdd(fz.ticket, 1u) == gridDim.x * gridDim.y - 1 ? 1u : 0u;
      __syncthreads();
      if (s_last) {
        __threadfence();  // the other blocks' runs
        double* smf = reinterpret_cast<double*>(sw);  // [128][8] doubles of the (now free) slice image
        constexpr int S = kFinalThreads / 8;
        const int P = (int)gridDim.x;
        const int ldrun = NCOL * ldp;  // (one-reduce mode: runs of 2 ldp doubles, ldp + nrows + 2 of them meaningful)
        for (int g8 = 0; g8 < (nout + 7) / 8; ++g8) {
#pragma unroll
          for (int q = 0; q < kFinalThreads / kTPB; ++q) {  // a 256-thread block plays k_final_rows_t's 1024 threads
            const int vt = threadIdx.x + q * kTPB;
            const int rl = vt & 7, pl = vt >> 3;
            const double* p = part + (g8 * 8 + rl);
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            int pid = pl;
            for (; pid + 3 * S < P; pid += 4 * S) {
              a0 += p[(int64_t)pid * ldrun];
              a1 += p[(int64_t)(pid + S) * ldrun];
              a2 += p[(int64_t)(pid + 2 * S) * ldrun];
              a3 += p[(int64_t)(pid + 3 * S) * ldrun];
            }
            for (; pid < P; pid += S) a0 += p[(int64_t)pid * ldrun];
            smf[pl * 8 + rl] = (a0 + a1) + (a2 + a3);
          }
          __syncthreads();
          if (threadIdx.x < 8 && g8 * 8 + (int)threadIdx.x < nout) {
            double t = 0.0;
            for (int k = 0; k < S; ++k) t += smf[k * 8 + threadIdx.x];
            fz.c_out[g8 * 8 + threadIdx.x] = t;
          }
          __syncthreads();
        }
        if (threadIdx.x == 0) fz.ticket[0] = 0u;  // ready for the next sweep
      }
    }
  }
}

// LDS bytes left for ONE slice of w in the default (4x4x4 MFMA) kernel: a block stages ncol slices and parks ncol sets of four
// coefficient runs of qtw_ldp(rows) doubles (one-reduce mode: two columns, and two more "rows" for the self terms)
static int64_t qtw_room(int flags, int nrows_max) {
  const int ncol = (flags & LZ_FLAG_ONE_REDUCE) ? 2 : 1;
  return (int64_t)155 * 1024 / ncol - (int64_t)(kTPB / 64) * qtw_ldp(nrows_max + (ncol == 2 ? 2 : 0)) * 8;
}

QtwPlan plan_qtw(int64_t len, int flags, const int* tune, int nrows_max) {
  QtwPlan p;
  int64_t target = (tune && tune[0] > 0) ? tune[0] : 0;
  // Two effects set the slice length (profiles/r01/ab_qtw_slice_balance.json): (a) a CU streams at a fixed share of the
  // HBM rate, so the pass finishes when the CU with the most blocks does - G blocks on 256 CUs run at
  // (G/256) / ceil(G/256) of the balanced rate (G = 612 or 815: 80 %, G = 489, 977, 1223: 95 %); (b) every tile ends in a
  // small reduction + partial stores, which favours long slices (L = 512: -9 %, 1024: -3 %, >= 2560: < 1 %).
  int64_t L = 512;
  if (target > 0) {
    L = round_up(target, 512);
  } else {
    // the longest slice that still gives every CU about two blocks and an even share; else the best-balanced one
    double best = -1.0;
    bool found = false;
    // the default kernel also parks four coefficient runs of qtw_ldp(n) doubles in LDS: with thousands of basis rows the
    // slice has to shrink to stay inside the 160 KiB of a CU (n = 4000 -> L <= 3072)
    int64_t lmax = (flags & LZ_FLAG_ONE_REDUCE) ? kQtwMaxL / 2 : kQtwMaxL;  // one-reduce mode stages two slices per block
    const int64_t room = qtw_room(flags, nrows_max);
    if (room / 8 < lmax) lmax = std::max<int64_t>(512, room / 8 / 512 * 512);
    for (int64_t cand = lmax; cand >= 1024 && !found; cand -= 512) {
      const int64_t G = (len + cand - 1) / cand;
      if (G < 480) continue;
      const double g = (double)G / kNumCU, bal = g / std::ceil(g);
      if (bal >= 0.93) {
        L = cand;
        found = true;
      } else if (bal * (double)cand / (double)(cand + 48) > best) {
        best = bal * (double)cand / (double)(cand + 48);
        L = cand;
      }
    }
  }
  if (L > kQtwMaxL) L = kQtwMaxL;
  p.L = L;
  p.G = (int)((len + L - 1) / L);
  // kernel family: 2 = 4x4x4 MFMA (default: matrix cores, line-coalesced loads), 1 = 16x16x4 MFMA (A/B arm),
  // 0 = VALU + shuffle reductions (LZ_FLAG_QTW_VALU)
  p.family = (flags & LZ_FLAG_QTW_VALU) ? 0 : ((flags & LZ_FLAG_QTW_MFMA) ? 1 : 2);
  if (p.family == 2 && qtw_room(flags, nrows_max) < L * 8)
    p.family = 0;  // ~5000 basis rows and more: the coefficient runs no longer fit next to any slice - VALU kernel, partials in HBM
  p.mfma = p.family != 0;
  p.variant = tune ? tune[1] : 0;
  p.P = p.family == 1 ? p.G * (kTPB / 64) : p.G;  // 16x16x4 kernel: per-wave partials; the others: one run per block
  return p;
}

template <int SCALE>
static hipError_t launch_qtw_t(double* V, int64_t ldv, int64_t len, int nrows, int j, const double* r, const double* nrm2,
                         double* beta_slot, const QtwPlan& plan, double* part, hipStream_t s, const QtwFuse& fz) {
  const size_t lds = (size_t)plan.L * sizeof(double);
  const dim3 grid(plan.G), block(kTPB);
#define LZ_QTW_ARGS V, ldv, len, nrows, j, r, nrm2, beta_slot, plan.L, plan.P, part
  if (plan.family == 2) {
    const int ncol = SCALE == 3 ? 2 : 1;
    const int ldp = qtw_ldp(SCALE == 3 ? nrows + 2 : nrows);
    size_t lds4 = ncol * (lds + (size_t)(kTPB / 64) * ldp * sizeof(double));  // slice(s) of w + the four waves' coefficient runs
    if (fz.ticket) lds4 = std::max<size_t>(lds4, (size_t)kFinalThreads * sizeof(double));  // the folded second stage's [128][8] image
    // row split for short vectors (see the kernel): only where the slices alone leave most of the chip idle, and only in
    // the modes whose staging stores nothing that other blocks of the same slice would race on
    dim3 grid2 = grid;
    if ((SCALE == 2 || SCALE == 4) && plan.G <= 64 && nrows > 8) {
      const int ntiles = (nrows + 7) / 8;
      int Y = 256 / plan.G;
      if (Y > ntiles) Y = ntiles;
      if (Y > 1) grid2 = dim3(plan.G, Y);
    }
    hipError_t err = hipSuccess;
    auto go = [&](auto kern) {
      // more than 64 KiB of dynamic LDS (long slices with many hundred basis rows) has to be allowed per kernel
      if (lds4 > 65536)
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
      if (err == hipSuccess)
        hipLaunchKernelGGL(kern, grid2, block, lds4, s, V, ldv, len, nrows, j, r, nrm2, beta_slot, plan.L, ldp, part, fz);
    };
    switch (plan.variant) {
      case 10: go(k_qtw_mfma4<SCALE, 4, 4>); break;
      case 11: go(k_qtw_mfma4<SCALE, 2, 8>); break;
      case 13: go(k_qtw_mfma4<SCALE, 8, 2>); break;
      default: go(k_qtw_mfma4<SCALE, 4, 2>); break;     // measured best: 2 tiles of 4 rows x 4 steps = 8 loads in flight per lane
    }
    return err;
  }
  if constexpr (SCALE == 3 || SCALE == 4) return hipErrorInvalidValue;  // these modes exist for the default (4x4x4 MFMA) family only
#ifdef LZ_KBENCH
  if constexpr (SCALE != 2 && SCALE != 3 && SCALE != 4) {
    if (plan.family == 1) return kb_launch_qtw_mfma<SCALE>(V, ldv, len, nrows, j, r, nrm2, beta_slot, plan, part, s);
  }
#else
  if (plan.family == 1) return hipErrorInvalidValue;  // (lz_set_options refuses LZ_FLAG_QTW_MFMA in the product library)
#endif
  switch (plan.variant) {
    case 1: hipLaunchKernelGGL((k_qtw_valu<SCALE, 4, 2>), grid, block, lds, s, LZ_QTW_ARGS); break;
    case 7: hipLaunchKernelGGL((k_qtw_valu<SCALE, 8, 2, 0>), grid, block, lds, s, LZ_QTW_ARGS); break;  // plain (cached) loads
    default:  // measured best on MI355X (profiles/r01): 8 rows x 2 positions = 16 loads in flight per lane
      if (nrows > 4)
        hipLaunchKernelGGL((k_qtw_valu<SCALE, 8, 2>), grid, block, lds, s, LZ_QTW_ARGS);
      else
        hipLaunchKernelGGL((k_qtw_valu<SCALE, 4, 2>), grid, block, lds, s, LZ_QTW_ARGS);
      break;
  }
#undef LZ_QTW_ARGS
  return hipSuccess;
}

hipError_t launch_qtw(double* V, int64_t ldv, int64_t len, int nrows, int j, const double* r, const double* nrm2,
                      double* beta_slot, const QtwPlan& plan, double* part, int mode, hipStream_t s, const QtwFuse* fuse) {
  const QtwFuse fz = fuse ? *fuse : QtwFuse();
  if (mode == 4) return launch_qtw_t<4>(V, ldv, len, nrows, j, r, nrm2, beta_slot, plan, part, s, fz);
  if (mode == 3) return launch_qtw_t<3>(V, ldv, len, nrows, j, r, nrm2, beta_slot, plan, part, s, fz);
  if (mode == 2) return launch_qtw_t<2>(V, ldv, len, nrows, j, r, nrm2, beta_slot, plan, part, s, fz);
  if (mode == 1) return launch_qtw_t<1>(V, ldv, len, nrows, j, r, nrm2, beta_slot, plan, part, s, fz);
  return launch_qtw_t<0>(V, ldv, len, nrows, j, r, nrm2, beta_slot, plan, part, s, fz);
}

// ------------------------------------------------------------------ re-orthogonalisation pass 2
// V[j] = 2 V[j] - (((c0 V0 + c1 V1) + c2 V2) + ...): NumPy's axis-0 reduction
// order of np.sum(c[:, None] * V, axis=0) (Lanczos.py:249), products and sums
// rounded separately.  One double2 column position per lane; the row loop is
// unrolled so 8 independent 16-byte loads are in flight per lane.
template <bool FUSED, int VAR = 0, int UN = 8>
__global__ __launch_bounds__(kTPB) void k_update(double* __restrict__ V, int64_t ldv, int64_t p0, int64_t n2, int nrows, int j,
                                                const double* __restrict__ c, const double* __restrict__ r,
                                                const double* __restrict__ beta) {
  const int64_t i = p0 + (int64_t)blockIdx.x * kTPB + threadIdx.x;  // double2 positions [p0, n2)
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
  if (k < nrows) {  // tail (< UN rows): still one batch of independent loads, not one round trip per row
    const int rem = nrows - k;  // block-uniform
    double2 q[UN];
#pragma unroll
    for (int u = 0; u < UN - 1; ++u)
      if (u < rem) q[u] = (FUSED && k + u == j) ? w : ld_stream<VAR>(col + (int64_t)(k + u) * ld2);
#pragma unroll
    for (int u = 0; u < UN - 1; ++u)
      if (u < rem) {
        const double ck = c[k + u];
        tx = tx + ck * q[u].x;
        ty = ty + ck * q[u].y;
      }
  }
  const double2 v = FUSED ? w : ld_stream<VAR>(out);
  st_stream<VAR>(out, make_double2(2.0 * v.x - tx, 2.0 * v.y - ty));
}

// Default pass-2 kernel: "slice owner".  A block owns 256*P consecutive double2 positions of the row and keeps their
// running sums in registers while it walks ALL basis rows, RU rows x P positions = RU*P independent 16-byte loads in
// flight per lane.  Same per-element arithmetic and order as k_update; the difference is WHEN V[j] is written: every
// block of a residency round finishes at about the same time, so the 8M bytes of stores arrive as a few bursts at the
// end instead of trickling in between the reads for the whole launch.  A 1 % trickle of writes costs ~15 % of HBM read
// throughput on MI355X (tools/probes/hbm_read_peak.hip: 6.8 -> 5.8 TB/s), presumably read<->write bus turnarounds.
// MODE 1 (one-reduce partial loop; FUSED, raw_c == 1): w = (r'' - alpha u) / beta is formed here from the two-term residual r'' (`r`),
// the newest basis vector u (`usub`) and alpha (`asub`) - k_three_term's expression, then the division - and the gate selects
// between the sweep (V[j] = 2 w - sum c_i V_i) and the plain V[j] = w: one launch per step does either.
template <bool FUSED, int P, int RU, int MODE = 0>
__global__ __launch_bounds__(kTPB) void k_update_slice(double* __restrict__ V, int64_t ldv, int64_t p0, int64_t n2, int nrows,
                                                      int j, const double* __restrict__ c, const double* __restrict__ r,
                                                      double* __restrict__ beta, int raw_c, int nblk_a, int64_t p0b,
                                                      int64_t n2b, int cG, int cldp, const int* __restrict__ gate,
                                                      const double* __restrict__ usub, const double* __restrict__ asub) {
  bool sweep = true;
  if (MODE == 1) {
    sweep = gate[0] != 0;
  } else if (gate && gate[0] == 0) {
    return;  // device-resident partial re-orthogonalisation: no sweep on this vector
  }
  // blocks [0, nblk_a) cover positions [p0, n2); any further blocks cover a second range [p0b, n2b) (the two faces of a
  // slab in overlap mode leave in one launch)
  int bx = blockIdx.x;
  if (bx >= nblk_a) {
    bx -= nblk_a;
    p0 = p0b;
    n2 = n2b;
  }
  const int64_t base = p0 + (int64_t)bx * (kTPB * P) + threadIdx.x;
  const int64_t ld2 = ldv >> 1;
  const double2* V2 = reinterpret_cast<const double2*>(V);
  int64_t pos[P];
  bool ok[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    pos[p] = base + (int64_t)p * kTPB;
    ok[p] = pos[p] < n2;
    if (!ok[p]) pos[p] = n2 - 1;  // valid address, result discarded
  }
  double2 w[P];
#pragma unroll
  for (int p = 0; p < P; ++p) w[p] = make_double2(0.0, 0.0);
  // raw_c (fused-norm mode inside lz_run): c still holds the reduced sums [V_0.r, ..., V_{j-1}.r, r.r]; beta and the
  // coefficients of w = r / beta are formed here with k_fused_prepare's arithmetic (one tiny launch less per step).
  // raw_c == 2 (fused small-problem mode): c is pass 1's block partials (cG runs of cldp doubles); a coefficient is their
  // sum in k_final_rows_t's order (from 0.0, block by block) - the second-stage launch is folded in here: the block adds
  // them up once, into LDS (nrows doubles of dynamic shared memory).
  extern __shared__ double cs[];
  if (FUSED && raw_c == 2) {
    for (int k = threadIdx.x; k < nrows; k += kTPB) {
      double t = 0.0;
      for (int b = 0; b < cG; ++b) t += c[(int64_t)b * cldp + k];
      cs[k] = t;
    }
    __syncthreads();
  }
  double bnorm = 1.0;
  if (FUSED && raw_c) {
    bnorm = sqrt(raw_c == 2 ? cs[j] : c[j]);
    if (blockIdx.x == 0 && threadIdx.x == 0) beta[0] = bnorm;
  }
  if (FUSED) {
    const double b = raw_c ? bnorm : beta[0];
    const double a = MODE == 1 ? asub[0] : 0.0;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      w[p] = reinterpret_cast<const double2*>(r)[pos[p]];
      if (MODE == 1) {
        const double2 uu = reinterpret_cast<const double2*>(usub)[pos[p]];
        w[p].x = w[p].x - uu.x * a;
        w[p].y = w[p].y - uu.y * a;
      }
      w[p].x = w[p].x / b;
      w[p].y = w[p].y / b;
    }
  }
  if (MODE == 1 && !sweep) nrows = 0;  // no sweep on this vector: V[j] = w
  double tx[P], ty[P];
#pragma unroll
  for (int p = 0; p < P; ++p) tx[p] = ty[p] = 0.0;
  for (int k = 0; k < nrows; k += RU) {
    double2 q[RU][P];
#pragma unroll
    for (int u = 0; u < RU; ++u)
      if (k + u < nrows) {  // block-uniform
        const double2* row = V2 + (int64_t)(k + u) * ld2;
#pragma unroll
        for (int p = 0; p < P; ++p) q[u][p] = (FUSED && k + u == j) ? w[p] : ld_stream<1>(row + pos[p]);
      }
#pragma unroll
    for (int u = 0; u < RU; ++u)
      if (k + u < nrows) {
        double ck = (FUSED && raw_c == 2) ? cs[k + u] : c[k + u];
        if (FUSED && raw_c) ck = (k + u == j) ? ck / (bnorm * bnorm) : ck / bnorm;
        if (!FUSED && k + u == j) {  // row j is V[j] itself (j < nrows): keep it for the final 2 v - t instead of re-reading it
#pragma unroll
          for (int p = 0; p < P; ++p) w[p] = q[u][p];
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
          tx[p] = tx[p] + ck * q[u][p].x;
          ty[p] = ty[p] + ck * q[u][p].y;
        }
      }
  }
  double2* out = reinterpret_cast<double2*>(V) + (int64_t)j * ld2;
#pragma unroll
  for (int p = 0; p < P; ++p)
    if (ok[p]) st_stream<1>(out + pos[p], (MODE == 1 && !sweep) ? w[p] : make_double2(2.0 * w[p].x - tx[p], 2.0 * w[p].y - ty[p]));
}

// Pass 2 for SHORT vectors with many basis rows (1Ddeuteron.py: M = n = 1001): the walk over the rows is a latency chain
// per element and there are only len elements to overlap chains with, so one ELEMENT (not a double2) per lane, 32 rows per
// trip, and the next trip's loads in flight while the current one is added up (two register buffers): four times the
// loads in flight of k_update_slice<.., 1, 32>.  Same per-element arithmetic and order (NumPy's), fused-norm mode only.
__global__ __launch_bounds__(kTPB) void k_update_elem(double* __restrict__ V, int64_t ldv, int64_t len, int nrows, int j,
                                                     const double* __restrict__ c, const double* __restrict__ r,
                                                     double* __restrict__ beta, int raw_c, int cG, int cldp) {
  extern __shared__ double cs[];
  if (raw_c == 2) {  // fused small-problem mode: add pass 1's block partials in k_final_rows_t's order, once per block
    for (int k = threadIdx.x; k < nrows; k += kTPB) {
      double t = 0.0;
      for (int b = 0; b < cG; ++b) t += c[(int64_t)b * cldp + k];
      cs[k] = t;
    }
    __syncthreads();
  }
  const double* cc = raw_c == 2 ? cs : c;
  double bnorm = 1.0;
  if (raw_c) {
    bnorm = sqrt(cc[j]);
    if (blockIdx.x == 0 && threadIdx.x == 0) beta[0] = bnorm;
  }
  const int64_t e0 = (int64_t)blockIdx.x * kTPB + threadIdx.x;
  const bool ok = e0 < len;
  const int64_t e = ok ? e0 : len - 1;  // valid address, result discarded
  const double b = raw_c ? bnorm : beta[0];
  const double w = r[e] / b;
  constexpr int RU = 32;
  double q[2][RU];
  auto load = [&](int k, double (&dst)[RU]) {
#pragma unroll
    for (int u = 0; u < RU; ++u)
      if (k + u < nrows) dst[u] = (k + u == j) ? w : __builtin_nontemporal_load(V + (int64_t)(k + u) * ldv + e);
  };
  double tx = 0.0;
  auto add = [&](int k, const double (&src)[RU]) {
#pragma unroll
    for (int u = 0; u < RU; ++u)
      if (k + u < nrows) {
        double ck = cc[k + u];
        if (raw_c) ck = (k + u == j) ? ck / (bnorm * bnorm) : ck / bnorm;
        tx = tx + ck * src[u];
      }
  };
  load(0, q[0]);
  for (int k = 0; k < nrows; k += 2 * RU) {
    if (k + RU < nrows) load(k + RU, q[1]);
    add(k, q[0]);
    if (k + 2 * RU < nrows) load(k + 2 * RU, q[0]);
    if (k + RU < nrows) add(k + RU, q[1]);
  }
  if (ok) V[(int64_t)j * ldv + e] = 2.0 * w - tx;
}

template <bool FUSED, int P, int RU>
static void launch_update_slice(double* V, int64_t ldv, int64_t p0, int64_t n2, int nrows, int j, const double* c, const double* r,
                                double* beta, int raw_c, hipStream_t s, int64_t p0b = 0, int64_t n2b = 0, int cG = 0, int cldp = 0,
                                const int* gate = nullptr, const double* usub = nullptr, const double* asub = nullptr) {
  const int grid = (int)((n2 - p0 + kTPB * P - 1) / (kTPB * P));
  const int grid_b = n2b > p0b ? (int)((n2b - p0b + kTPB * P - 1) / (kTPB * P)) : 0;
  if constexpr (FUSED) {
    if (usub) {  // one-reduce partial loop: the three-term subtraction and the sweep / no-sweep choice live in the kernel
      hipLaunchKernelGGL((k_update_slice<true, P, RU, 1>), dim3(grid + grid_b), dim3(kTPB), 0, s, V, ldv, p0, n2, nrows, j, c, r, beta, raw_c, grid, p0b, n2b,
                         cG, cldp, gate, usub, asub);
      return;
    }
  }
  hipLaunchKernelGGL((k_update_slice<FUSED, P, RU>), dim3(grid + grid_b), dim3(kTPB), raw_c == 2 ? (size_t)nrows * sizeof(double) : 0, s, V, ldv, p0,
                     n2, nrows, j, c, r, beta, raw_c, grid, p0b, n2b, cG, cldp, gate, (const double*)nullptr, (const double*)nullptr);
}

void launch_update(double* V, int64_t ldv, int64_t len, int nrows, int j, const double* c, const double* r_fused,
                   double* beta, int variant, hipStream_t s, int64_t pos_lo, int64_t pos_hi, int raw_c, int64_t pos_lo_b,
                   int64_t pos_hi_b, int cG, int cldp, const int* gate, const double* usub, const double* asub) {
  // gate (device-resident partial re-orthogonalisation): the default slice-owner kernels only
  // double2 positions [pos_lo, pos_hi) of the row (default: the whole row)
  const int64_t n2 = pos_hi >= 0 ? pos_hi : (len >> 1);
  const int64_t p0 = pos_lo > 0 ? pos_lo : 0;
  if (n2 <= p0) return;
  const int grid = (int)((n2 - p0 + kTPB - 1) / kTPB);
  if (!r_fused && variant == 1) {  // A/B arm: plain (cached) loads
    hipLaunchKernelGGL((k_update<false, 0, 8>), dim3(grid), dim3(kTPB), 0, s, V, ldv, p0, n2, nrows, j, c, r_fused, beta);
    return;
  }
  if (variant == 2) {  // A/B arm: one position per lane, non-temporal loads (stores trickle through the whole launch)
    if (r_fused)
      hipLaunchKernelGGL((k_update<true, 1, 8>), dim3(grid), dim3(kTPB), 0, s, V, ldv, p0, n2, nrows, j, c, r_fused, beta);
    else
      hipLaunchKernelGGL((k_update<false, 1, 8>), dim3(grid), dim3(kTPB), 0, s, V, ldv, p0, n2, nrows, j, c, r_fused, beta);
    return;
  }
  // default: slice-owner kernel.  Positions per lane P in {16, 8, 4, 2} (always 16 loads in flight): like the Q.w pass this
  // one finishes with the most loaded CU, so take the P whose block count spreads most evenly over the 256 CUs
  // (M = 1.25e6: 306 blocks of P = 8 run at 60 %, 1223 blocks of P = 2 at 95 %); ties go to the larger P.
  const int64_t span = n2 - p0;
  int P = variant == 3 ? 8 : (variant == 4 ? 4 : 0);
  if (!P) {
    double best = -1.0;
    P = 2;
    for (int cand : {16, 8, 4, 2}) {
      const int64_t G = (span + (int64_t)kTPB * cand - 1) / ((int64_t)kTPB * cand);
      const double g = (double)G / kNumCU, bal = g / std::ceil(g);
      if (bal >= 0.93) {  // 16 loads in flight per lane: one block per CU already keeps the CU's share of HBM busy
        P = cand;
        break;
      }
      if (bal > best) {
        best = bal;
        P = cand;
      }
    }
  }
  if (variant == 6 || (variant == 0 && P == 16 && span > 16384)) {
    // 16 positions per lane, one row per trip (still 16 loads in flight): half as many, longer-lived blocks - fewer
    // residency rounds, so V[j] leaves in fewer, larger write bursts (1250 -> 1218 us at the headline)
    if (r_fused) launch_update_slice<true, 16, 1>(V, ldv, p0, n2, nrows, j, c, r_fused, beta, raw_c, s, 0, 0, cG, cldp, gate, usub, asub);
    else launch_update_slice<false, 16, 1>(V, ldv, p0, n2, nrows, j, c, r_fused, beta, 0, s, 0, 0, 0, 0, gate);
    return;
  }
  if (variant == 0 && span <= 4096 && r_fused && nrows > 64 && pos_hi < 0 && pos_lo <= 0 && !gate) {
    // a short vector against many rows: the element-per-lane, double-buffered kernel (see k_update_elem)
    const int grid_e = (int)((len + kTPB - 1) / kTPB);
    hipLaunchKernelGGL(k_update_elem, dim3(grid_e), dim3(kTPB), raw_c == 2 ? (size_t)nrows * sizeof(double) : 0, s, V, ldv, len, nrows, j, c, r_fused,
                       beta, raw_c, cG, cldp);
    return;
  }
  if ((variant == 0 && span <= 16384) || variant == 5) {
    // a face of the slab (overlap mode) or a tiny vector: a handful of blocks walk all rows, which is a latency chain,
    // not a bandwidth problem - one position per lane (most blocks) and 32 rows in flight per lane
    if (r_fused) launch_update_slice<true, 1, 32>(V, ldv, p0, n2, nrows, j, c, r_fused, beta, raw_c, s, pos_lo_b, pos_hi_b, cG, cldp, gate, usub, asub);
    else launch_update_slice<false, 1, 32>(V, ldv, p0, n2, nrows, j, c, r_fused, beta, 0, s, pos_lo_b, pos_hi_b, 0, 0, gate);
  } else if (P == 8) {
    if (r_fused) launch_update_slice<true, 8, 2>(V, ldv, p0, n2, nrows, j, c, r_fused, beta, raw_c, s, 0, 0, cG, cldp, gate, usub, asub);
    else launch_update_slice<false, 8, 2>(V, ldv, p0, n2, nrows, j, c, r_fused, beta, 0, s, 0, 0, 0, 0, gate);
  } else if (P == 4) {
    if (r_fused) launch_update_slice<true, 4, 4>(V, ldv, p0, n2, nrows, j, c, r_fused, beta, raw_c, s, 0, 0, cG, cldp, gate, usub, asub);
    else launch_update_slice<false, 4, 4>(V, ldv, p0, n2, nrows, j, c, r_fused, beta, 0, s, 0, 0, 0, 0, gate);
  } else {
    if (r_fused) launch_update_slice<true, 2, 8>(V, ldv, p0, n2, nrows, j, c, r_fused, beta, raw_c, s, 0, 0, cG, cldp, gate, usub, asub);
    else launch_update_slice<false, 2, 8>(V, ldv, p0, n2, nrows, j, c, r_fused, beta, 0, s, 0, 0, 0, 0, gate);
  }
}

// partial re-orthogonalisation mode, steps without a sweep: V[j] = r / sqrt(nrm2) and nothing else
__global__ __launch_bounds__(kTPB) void k_scale_store(double* __restrict__ vj, const double* __restrict__ r,
                                                     const double* __restrict__ nrm2, double* __restrict__ beta_slot, int64_t n2,
                                                     const int* __restrict__ gate) {
  if (gate && gate[0] != 0) return;  // device-resident partial re-orthogonalisation: the sweep kernels wrote V[j] and beta
  const double beta = sqrt(nrm2[0]);
  if (blockIdx.x == 0 && threadIdx.x == 0) beta_slot[0] = beta;
  const double2* r2 = reinterpret_cast<const double2*>(r);
  double2* v2 = reinterpret_cast<double2*>(vj);
  for (int64_t i = (int64_t)blockIdx.x * kTPB + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kTPB) {
    double2 x = r2[i];
    x.x = x.x / beta;
    x.y = x.y / beta;
    v2[i] = x;
  }
}
void launch_scale_store(double* vj, const double* r, const double* nrm2, double* beta_slot, int64_t len, hipStream_t s, const int* gate) {
  const int64_t n2 = len >> 1;
  int64_t g = (n2 + kTPB - 1) / kTPB;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(k_scale_store, dim3((int)g), dim3(kTPB), 0, s, vj, r, nrm2, beta_slot, n2, gate);
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
// One-reduce mode, after the single all-reduce of buf = [p_0..p_{m-1}, r''.r'', gap.., q_0..q_{m-1}, u.u, u.r'', alpha]
// (p_i = V_i.r'', q_i = V_i.u, m rows, second half at offset ldp, alpha = u.(A u) at ldp + m + 2):
//   c_i = V_i . (r'' - alpha u) = p_i - alpha q_i,   c_m = |r'' - alpha u|^2 = r''.r'' - 2 alpha u.r'' + alpha^2 u.u
// left in buf[0..m] exactly where the update kernel's raw-sums path expects [V_i.r ..., r.r]; alpha to alpha_slot.
// The three-sum form of |r|^2 cancels: its relative error is about eps (alpha^2 + beta^2) / beta^2.  Where that would
// break the 1e-10 bar (|r|^2 below 1e-4 of r''.r'', or not positive at all: shifted spectra, a nearly converged Krylov
// space) the kernel raises the sticky flag `bad`; lz_run then repeats the solve on the default two-reduce loop.
__global__ void k_onereduce_prepare(double* __restrict__ buf, int m, int ldp, double* __restrict__ alpha_slot, double* __restrict__ bad) {
  const double a = buf[ldp + m + 2];
  for (int i = threadIdx.x; i < m; i += blockDim.x) buf[i] = buf[i] - a * buf[ldp + i];
  __syncthreads();
  if (threadIdx.x == 0) {
    const double rr = buf[m], uu = buf[ldp + m], ur = buf[ldp + m + 1];
    const double v = (rr - 2.0 * a * ur) + a * a * uu;
    if (!(v > 1e-4 * rr)) bad[0] = 1.0;
    buf[m] = v;
    alpha_slot[0] = a;
  }
}
void launch_onereduce_prepare(double* buf, int m, int ldp, double* alpha_slot, double* bad, hipStream_t s) {
  hipLaunchKernelGGL(k_onereduce_prepare, dim3(1), dim3(kTPB), 0, s, buf, m, ldp, alpha_slot, bad);
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
    r2[i] = x;  // plain store: r is re-read by the very next kernel (an nt store measured no different)
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

// One-reduce partial loop, behind the SpMV of step j: r'' = y - beta v_{j-1} (k_three_term's expression) and, in the same pass,
// the block partials of the three self terms the next step's ||r||^2 is made of - r''.r'' -> part[b], u.r'' -> part[G + b],
// u.u -> part[2 G + b] with u = v_j - so that a step without a sweep needs no pass 1 at all.
__global__ __launch_bounds__(kTPB) void k_three_term_self(double* __restrict__ r, const double* __restrict__ u, const double* __restrict__ vm,
                                                         const double* __restrict__ beta, int64_t n2, double* __restrict__ part) {
  __shared__ double sm[kTPB / 64];
  const double b = vm ? beta[0] : 0.0;
  double2* r2 = reinterpret_cast<double2*>(r);
  const double2* u2 = reinterpret_cast<const double2*>(u);
  const double2* m2 = reinterpret_cast<const double2*>(vm);
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kTPB + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kTPB) {
    double2 x = r2[i];
    const double2 uu = u2[i];  // (v_j was just written and read by the SpMV: plain load)
    if (vm) {
      const double2 m = ld_stream<1>(m2 + i);
      x.x = x.x - m.x * b;
      x.y = x.y - m.y * b;
      r2[i] = x;
    }
    s0 = fma(x.x, x.x, s0);
    s0 = fma(x.y, x.y, s0);
    s1 = fma(uu.x, x.x, s1);
    s1 = fma(uu.y, x.y, s1);
    s2 = fma(uu.x, uu.x, s2);
    s2 = fma(uu.y, uu.y, s2);
  }
  s0 = block_sum(s0, sm);
  s1 = block_sum(s1, sm);
  s2 = block_sum(s2, sm);
  if (threadIdx.x == 0) {
    part[blockIdx.x] = s0;
    part[gridDim.x + blockIdx.x] = s1;
    part[2 * gridDim.x + blockIdx.x] = s2;
  }
}
int launch_three_term_self(double* r, const double* u, const double* vm, const double* beta, int64_t len, double* part, hipStream_t s) {
  const int64_t n2 = len >> 1;
  int64_t g = (n2 + kTPB - 1) / kTPB;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(k_three_term_self, dim3((int)g), dim3(kTPB), 0, s, r, u, vm, beta, n2, part);
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

}  // namespace lz
