// Two-sided (bi-orthogonal) Lanczos of the reference's Irregular copy: vector kernels.
//
// Reference: Python/Irregular/IrrLanczos.py:77-187 (driver loop) and :408-441 (bireorthogonalize, default branch).
// Four (n, M) bases live on the device: Q (right Krylov vectors, published as V), P (left), and Qb / Pb, the
// orthonormalised copies the reference projects on.  Everything per step is a chain of  dot -> scalar -> axpy  pairs
// (modified Gram-Schmidt is sequential by construction), so the design goal is: one streaming kernel per link of the
// chain, no host round trip anywhere.
//
//  * k_bi<FIRST, PEND, DOTS> handles one link for BOTH sides at once: it (optionally) forms the working pair from a
//    source pair divided by device-resident factors, (optionally) applies the axpy that the PREVIOUS link's dots
//    decided - the coefficients uv/uu are formed in the kernel from the raw sums - stores the pair, and accumulates the
//    dots the NEXT link needs.  That is 6 vector reads + 2 writes per Gram-Schmidt step instead of 8 + 2 for separate
//    dot and axpy kernels, and half the launches.
//  * k_bi_final<EPI> folds the per-block partials in a fixed order and turns the sums into the scalars the next kernel
//    reads (sqrt|q.p| and its sign, norms, alpha, beta/gamma).
//  * element-wise arithmetic is written exactly as NumPy evaluates the reference's expressions (division by the norm,
//    (uv/uu) * vector then subtraction, no FMA contraction: the library is built with -ffp-contract=off); only the
//    summation order of the dots differs from BLAS ddot.
#include "lz_device.h"

namespace lz {

constexpr int kBiMaxBlocks = 1024;

// EPI 0: S[0..K) = sums.                       1: f = {sqrt|S0|, sqrt|S0|, sign(S0)}       2: f = {sqrt S0, sqrt S1, 1}
//     3: o0[0] = (S0 + S1) / 2  (alpha_j)       4: o0[0] = beta = sqrt|S0|, o1[0] = gamma = S0 / beta, f = {beta, gamma, 1}
//     5: o0[0] = S0
template <int EPI>
__device__ __forceinline__ void bi_epilogue(const double* tot, int K, double* S, double* f, double* o0, double* o1) {
  if (EPI == 0) {
    for (int k = 0; k < K; ++k) S[k] = tot[k];
  } else if (EPI == 1) {
    const double sc = sqrt(fabs(tot[0]));
    f[0] = sc;
    f[1] = sc;
    f[2] = tot[0] > 0.0 ? 1.0 : (tot[0] < 0.0 ? -1.0 : 0.0);  // np.sign
  } else if (EPI == 2) {
    f[0] = sqrt(tot[0]);
    f[1] = sqrt(tot[1]);
    f[2] = 1.0;
  } else if (EPI == 3) {
    o0[0] = (tot[0] + tot[1]) / 2;
  } else if (EPI == 4) {
    const double be = sqrt(fabs(tot[0]));
    const double ga = tot[0] / be;
    o0[0] = be;
    o1[0] = ga;
    f[0] = be;
    f[1] = ga;
    f[2] = 1.0;
  } else {
    o0[0] = tot[0];
  }
}

// Single-launch links (ticket != nullptr): every block publishes its partials, takes a ticket, and the block that draws
// the last one folds all partials - in index order, so the result does not depend on which block that is - and runs the
// epilogue.  __threadfence() is the agent-scope release/acquire pair that makes the partials written on other XCDs
// (separate, non-coherent L2s) visible.  Saves the second dependent launch of every link of the chain, but the release
// writes back every L2 line the kernel dirtied, which measured 2-4x slower than the extra launch: A/B arm only (see
// bi_ticket() in lz_twosided_api.hip); the same result kept the main path's second-stage reductions as separate kernels.
template <int EPI>
__device__ __forceinline__ void bi_last_block_final(unsigned* ticket, const double* part, int NB, int K, double* S, double* f,
                                                    double* o0, double* o1, double* sm) {
  __shared__ int is_last;
  __shared__ double tot[4];
  __threadfence();
  if (threadIdx.x == 0) is_last = atomicAdd(ticket, 1u) == (unsigned)gridDim.x - 1u;
  __syncthreads();
  if (!is_last) return;
  __threadfence();
  for (int k = 0; k < K; ++k) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < NB; i += kTPB) acc += __builtin_nontemporal_load(part + (int64_t)k * NB + i);
    const double t = block_sum(acc, sm);
    if (threadIdx.x == 0) tot[k] = t;
  }
  if (threadIdx.x == 0) {
    bi_epilogue<EPI>(tot, K, S, f, o0, o1);
    *ticket = 0u;  // ready for the next launch (stream order)
  }
}

// The fold of k_bi_final<0> (1024 threads: one partial per thread, wave shuffle tree, 16 wave sums added in order) for the
// four sums of a link, evaluated by every 256-thread block of the CONSUMER kernel: each wave plays four of the sixteen.
// Same additions in the same order, so a link whose fold rides here produces the same bits as with the separate kernel.
__device__ __forceinline__ void bi_fold4_emulated(const double* __restrict__ pp, int NBp, double (&tot)[4]) {
  __shared__ double sm16[4][16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    for (int vw = w; vw < 16; vw += kTPB / 64) {
      double acc = 0.0;
      for (int i = vw * 64 + lane; i < NBp; i += 1024) acc += pp[(int64_t)k * NBp + i];
      acc = wave_sum(acc);
      if (lane == 0) sm16[k][vw] = acc;
    }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += sm16[k][q];
    tot[k] = t;
  }
}

// pp != nullptr (PEND links of the default path): the previous link's per-block partials (NBp blocks x 4 sums) - the
// fold that k_bi_final<0> would do in a launch of its own is done here, by every block (see bi_fold4_emulated).
template <int FIRST, int PEND, int DOTS>
__global__ __launch_bounds__(kTPB) void k_bi(double* x, double* y, const double* xs, const double* ys, const double* f,
                                            const double* ap, const double* bp, const double* Sp, const double* a,
                                            const double* b, int64_t n2, int NB, double* __restrict__ part, unsigned* ticket,
                                            double* S, double* fo, const double* __restrict__ pp, int NBp) {
  __shared__ double sm[kTPB / 64];
  double fx = 1.0, fy = 1.0, sg = 1.0, cx = 0.0, cy = 0.0;
  if (FIRST) {
    fx = f[0];
    fy = f[1];
    sg = f[2];
  }
  if (PEND) {
    if (pp) {
      double tot[4];
      bi_fold4_emulated(pp, NBp, tot);
      cx = tot[0] / tot[1];
      cy = tot[2] / tot[3];
    } else {
      cx = Sp[0] / Sp[1];
      cy = Sp[2] / Sp[3];
    }
  }
  double2* x2 = reinterpret_cast<double2*>(x);
  double2* y2 = reinterpret_cast<double2*>(y);
  const double2* xs2 = reinterpret_cast<const double2*>(xs);
  const double2* ys2 = reinterpret_cast<const double2*>(ys);
  const double2* ap2 = reinterpret_cast<const double2*>(ap);
  const double2* bp2 = reinterpret_cast<const double2*>(bp);
  const double2* a2 = reinterpret_cast<const double2*>(a);
  const double2* b2 = reinterpret_cast<const double2*>(b);
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kTPB + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kTPB) {
    double2 X, Y;
    if (FIRST) {
      X = xs2[i];
      Y = ys2[i];
      X.x = X.x / fx;
      X.y = X.y / fx;
      Y.x = Y.x / fy * sg;
      Y.y = Y.y / fy * sg;
    } else {
      X = x2[i];
      Y = y2[i];
    }
    if (PEND) {
      const double2 A = ld_stream<1>(ap2 + i), B = ld_stream<1>(bp2 + i);
      X.x = X.x - cx * A.x;
      X.y = X.y - cx * A.y;
      Y.x = Y.x - cy * B.x;
      Y.y = Y.y - cy * B.y;
    }
    if (FIRST || PEND) {
      x2[i] = X;
      y2[i] = Y;
    }
    if (DOTS == 0) {
      const double2 A = ld_stream<1>(a2 + i), B = ld_stream<1>(b2 + i);
      s0 = fma(X.x, A.x, s0);
      s0 = fma(X.y, A.y, s0);
      s1 = fma(A.x, A.x, s1);
      s1 = fma(A.y, A.y, s1);
      s2 = fma(Y.x, B.x, s2);
      s2 = fma(Y.y, B.y, s2);
      s3 = fma(B.x, B.x, s3);
      s3 = fma(B.y, B.y, s3);
    } else if (DOTS == 1) {
      s0 = fma(X.x, Y.x, s0);
      s0 = fma(X.y, Y.y, s0);
    } else if (DOTS == 2) {
      s0 = fma(X.x, X.x, s0);
      s0 = fma(X.y, X.y, s0);
      s1 = fma(Y.x, Y.x, s1);
      s1 = fma(Y.y, Y.y, s1);
    }
  }
  if (DOTS == 3) return;
  constexpr int K = DOTS == 0 ? 4 : (DOTS == 1 ? 1 : 2);
  double v[4] = {s0, s1, s2, s3};
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const double t = block_sum(v[k], sm);
    if (threadIdx.x == 0) part[(int64_t)k * NB + blockIdx.x] = t;
  }
  if (ticket) bi_last_block_final<DOTS>(ticket, part, NB, K, S, fo, nullptr, nullptr, sm);  // DOTS d pairs with epilogue d
}

// r = r - c0 * u ; s = s - c1 * v  (SUB), then DOTS 0: [da . r, db . s]   1: [r . s]   2: [da . r]
template <int SUB, int DOTS>
__global__ __launch_bounds__(kTPB) void k_bi_two_term(double* r, double* s, const double* u, const double* v, const double* c0p,
                                                     const double* c1p, const double* da, const double* db, int64_t n2, int NB,
                                                     double* __restrict__ part, unsigned* ticket, double* fo, double* o0, double* o1) {
  __shared__ double sm[kTPB / 64];
  const double c0 = SUB ? c0p[0] : 0.0, c1 = SUB ? c1p[0] : 0.0;
  double2* r2 = reinterpret_cast<double2*>(r);
  double2* s2p = reinterpret_cast<double2*>(s);
  const double2* u2 = reinterpret_cast<const double2*>(u);
  const double2* v2 = reinterpret_cast<const double2*>(v);
  const double2* da2 = reinterpret_cast<const double2*>(da);
  const double2* db2 = reinterpret_cast<const double2*>(db);
  double a0 = 0.0, a1 = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kTPB + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kTPB) {
    double2 R = r2[i], S = make_double2(0.0, 0.0);
    if (DOTS != 2) S = s2p[i];
    if (SUB) {
      const double2 U = u2[i], V = v2[i];
      R.x = R.x - c0 * U.x;
      R.y = R.y - c0 * U.y;
      S.x = S.x - c1 * V.x;
      S.y = S.y - c1 * V.y;
      r2[i] = R;
      s2p[i] = S;
    }
    if (DOTS == 0) {
      const double2 A = da2[i], B = db2[i];
      a0 = fma(A.x, R.x, a0);
      a0 = fma(A.y, R.y, a0);
      a1 = fma(B.x, S.x, a1);
      a1 = fma(B.y, S.y, a1);
    } else if (DOTS == 1) {
      a0 = fma(R.x, S.x, a0);
      a0 = fma(R.y, S.y, a0);
    } else {
      const double2 A = da2[i];
      a0 = fma(A.x, R.x, a0);
      a0 = fma(A.y, R.y, a0);
    }
  }
  double t = block_sum(a0, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
  if (DOTS == 0) {
    t = block_sum(a1, sm);
    if (threadIdx.x == 0) part[(int64_t)NB + blockIdx.x] = t;
  }
  if (ticket) bi_last_block_final<DOTS + 3>(ticket, part, NB, DOTS == 0 ? 2 : 1, nullptr, fo, o0, o1, sm);
}

// Two-launch variant of the fold (ticket == nullptr; A/B arm): one block, same epilogues.
constexpr int kBiFinalThreads = 1024;
template <int EPI>
__global__ __launch_bounds__(kBiFinalThreads) void k_bi_final(const double* __restrict__ part, int NB, int K, double* __restrict__ S,
                                                             double* __restrict__ f, double* __restrict__ o0,
                                                             double* __restrict__ o1) {
  __shared__ double sm[kBiFinalThreads / 64];
  __shared__ double tot[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int k = 0; k < K; ++k) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < NB; i += kBiFinalThreads) acc += part[(int64_t)k * NB + i];
    acc = wave_sum(acc);
    if (lane == 0) sm[w] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
#pragma unroll
      for (int q = 0; q < kBiFinalThreads / 64; ++q) t += sm[q];
      tot[k] = t;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) bi_epilogue<EPI>(tot, K, S, f, o0, o1);
}

static int bi_grid(int64_t n2) {
  int64_t g = (n2 + kTPB - 1) / kTPB;
  if (g > kBiMaxBlocks) g = kBiMaxBlocks;
  return g < 1 ? 1 : (int)g;
}

int bi_partials_needed() { return 4 * kBiMaxBlocks; }

// pend_part: where the PREVIOUS link left its block partials when its fold was deferred (else nullptr: Sp holds the sums);
// defer_fold: this link's own fold is left to its consumer (dots == 0 only; the partials stay in `part`).
void launch_bi(int first, int pend, int dots, double* x, double* y, const double* xs, const double* ys, const double* f,
               const double* ap, const double* bp, const double* Sp, const double* a, const double* b, int64_t len, double* part,
               int epi, double* S, double* fo, double* o0, double* o1, unsigned* ticket, hipStream_t s, const double* pend_part,
               bool defer_fold) {
  const int64_t n2 = len >> 1;
  const int NB = bi_grid(n2);
  const dim3 g(NB), t(kTPB);
  const double* pp = pend ? pend_part : nullptr;
  const int NBp = NB;  // every link of a chain runs over the same length
#define LZ_BI(F, P, D) hipLaunchKernelGGL((k_bi<F, P, D>), g, t, 0, s, x, y, xs, ys, f, ap, bp, Sp, a, b, n2, NB, part, ticket, S, fo, pp, NBp)
  const int key = first * 100 + pend * 10 + dots;
  switch (key) {
    case 100: LZ_BI(1, 0, 0); break;  // first link of a Gram-Schmidt chain, pair formed from a scaled source
    case 0: LZ_BI(0, 0, 0); break;    // first link, pair already stored (step API)
    case 10: LZ_BI(0, 1, 0); break;   // inner link
    case 11: LZ_BI(0, 1, 1); break;   // last axpy + q.p
    case 12: LZ_BI(0, 1, 2); break;   // last axpy + |u|^2, |v|^2
    case 102: LZ_BI(1, 0, 2); break;  // rescale + norms
    case 103: LZ_BI(1, 0, 3); break;  // normalise
    case 2: LZ_BI(0, 0, 2); break;    // norms only
    case 1: LZ_BI(0, 0, 1); break;    // q.p of a stored pair (bireorthogonalize with j = 0: nothing to project on)
    default: break;
  }
#undef LZ_BI
  if (dots == 3 || ticket || (defer_fold && dots == 0)) return;
  const int K = dots == 0 ? 4 : (dots == 1 ? 1 : 2);
  const dim3 g1(1), t1(kBiFinalThreads);
  switch (epi) {
    case 0: hipLaunchKernelGGL((k_bi_final<0>), g1, t1, 0, s, part, NB, K, S, fo, o0, o1); break;
    case 1: hipLaunchKernelGGL((k_bi_final<1>), g1, t1, 0, s, part, NB, K, S, fo, o0, o1); break;
    default: hipLaunchKernelGGL((k_bi_final<2>), g1, t1, 0, s, part, NB, K, S, fo, o0, o1); break;
  }
}

void launch_bi_two_term(int sub, int dots, double* r, double* sv, const double* u, const double* v, const double* c0, const double* c1,
                        const double* da, const double* db, int64_t len, double* part, double* fo, double* o0, double* o1,
                        unsigned* ticket, hipStream_t s) {
  const int64_t n2 = len >> 1;
  const int NB = bi_grid(n2);
  const dim3 g(NB), t(kTPB), g1(1), t1(kBiFinalThreads);
#define LZ_TT(S_, D_) hipLaunchKernelGGL((k_bi_two_term<S_, D_>), g, t, 0, s, r, sv, u, v, c0, c1, da, db, n2, NB, part, ticket, fo, o0, o1)
  if (dots == 0) {
    if (sub) LZ_TT(1, 0); else LZ_TT(0, 0);
    if (!ticket) hipLaunchKernelGGL((k_bi_final<3>), g1, t1, 0, s, part, NB, 2, nullptr, fo, o0, o1);
  } else if (dots == 1) {
    LZ_TT(1, 1);
    if (!ticket) hipLaunchKernelGGL((k_bi_final<4>), g1, t1, 0, s, part, NB, 1, nullptr, fo, o0, o1);
  } else {
    LZ_TT(0, 2);
    if (!ticket) hipLaunchKernelGGL((k_bi_final<5>), g1, t1, 0, s, part, NB, 1, nullptr, fo, o0, o1);
  }
#undef LZ_TT
}

// ---- IrrLanczos.bireorthogonalize(mem_safe=True) (IrrLanczos.py:398-407): one classical Gram-Schmidt sweep of row j of one
// base against ALL n rows of the other base, each coefficient divided by that row's own squared norm -------------------------
// coef[i] = (x . B[i]) / (B[i] . B[i]) for i < j, 0 for i == j, x . B[i] for i > j (the reference's uu[j:] = 1, uv[j] = 0;
// a zero row i < j gives 0/0 = nan exactly as NumPy does).  One block per row; padded tail entries of the rows are zero.
__global__ __launch_bounds__(kTPB) void k_ms_coef(const double* __restrict__ x, const double* __restrict__ B, int64_t ldv, int64_t n2, int j,
                                                  double* __restrict__ coef) {
  __shared__ double sm[2 * (kTPB / 64)];
  const int i = blockIdx.x;
  const double2* b2 = reinterpret_cast<const double2*>(B + (int64_t)i * ldv);
  const double2* x2 = reinterpret_cast<const double2*>(x);
  double uv = 0.0, uu = 0.0;
  for (int64_t t = threadIdx.x; t < n2; t += kTPB) {
    const double2 b = b2[t], a = x2[t];
    uv = fma(a.x, b.x, uv);
    uv = fma(a.y, b.y, uv);
    uu = fma(b.x, b.x, uu);
    uu = fma(b.y, b.y, uu);
  }
  uv = wave_sum(uv);
  uu = wave_sum(uu);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sm[2 * w] = uv;
    sm[2 * w + 1] = uu;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
    for (int q = 0; q < kTPB / 64; ++q) {
      a += sm[2 * q];
      b += sm[2 * q + 1];
    }
    coef[i] = i == j ? 0.0 : (i > j ? a : a / b);
  }
}

// x[k] -= sum_i coef[i] * B[i][k], the sum formed row by row from row 0 with every product rounded on its own - the order and
// rounding of np.sum(coef[:, None] * B, axis=0) (a reduction over the slow axis adds whole rows in turn; no pairwise tree there).
__global__ __launch_bounds__(kTPB) void k_ms_apply(double* __restrict__ x, const double* __restrict__ B, int64_t ldv, int64_t n2, int n,
                                                   const double* __restrict__ coef) {
  extern __shared__ double sc[];
  for (int i = threadIdx.x; i < n; i += kTPB) sc[i] = coef[i];
  __syncthreads();
  double2* x2 = reinterpret_cast<double2*>(x);
  for (int64_t t = (int64_t)blockIdx.x * kTPB + threadIdx.x; t < n2; t += (int64_t)gridDim.x * kTPB) {
    const double2* b2 = reinterpret_cast<const double2*>(B) + t;
    const int64_t ld2 = ldv >> 1;
    double2 b = b2[0];
    double ax = __dmul_rn(sc[0], b.x), ay = __dmul_rn(sc[0], b.y);
    for (int i = 1; i < n; ++i) {
      b = b2[(int64_t)i * ld2];
      ax = __dadd_rn(ax, __dmul_rn(sc[i], b.x));
      ay = __dadd_rn(ay, __dmul_rn(sc[i], b.y));
    }
    double2 v = x2[t];
    v.x = __dsub_rn(v.x, ax);
    v.y = __dsub_rn(v.y, ay);
    x2[t] = v;
  }
}

// one half of the mem_safe branch: row j of `X` against the n rows of `B` (both (n, ldv) bases, ldv even, rows zero-padded to len)
void launch_bi_mem_safe(double* xrow, const double* B, int64_t ldv, int n, int j, int64_t len, double* coef, hipStream_t s) {
  const int64_t n2 = len >> 1;
  hipLaunchKernelGGL(k_ms_coef, dim3(n), dim3(kTPB), 0, s, xrow, B, ldv, n2, j, coef);
  hipLaunchKernelGGL(k_ms_apply, dim3(bi_grid(n2)), dim3(kTPB), (size_t)n * sizeof(double), s, xrow, B, ldv, n2, n, coef);
}

}  // namespace lz
