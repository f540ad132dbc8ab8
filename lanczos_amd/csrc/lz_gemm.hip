// FP64-MFMA kernels: Ritz back-transform Y = V^T-layout x S, Gram matrix Y^T Y, Ritz-vector quality sums.
#include <algorithm>
#include <type_traits>
#include <utility>
#include <vector>

#include "lz_device.h"

namespace lz {

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



// The same kernel with 32-ROW tiles (the default for n <= 128): half the vector-memory and LDS instructions per MFMA.
// Measured on the 16-row version (in-kernel clocks, round 3): the oldest wave of a SIMD runs at 0.93-0.96 of the MFMA issue
// floor, but the SIMD as a whole (both waves) only at ~0.80 - every vector-memory instruction takes MFMA issue time from the
// SIMD (~50-165 cycles each), and a 16-row tile costs KS loads + 4 NT stores per NT KS MFMAs.  Here
//  * one 16-byte load per k-step brings V[k][m0 + 2 lr] and V[k][m0 + 2 lr + 1]: the A fragments of TWO 16-row tiles, the
//    even and the odd rows of a 32-row block; every B fragment read from LDS feeds two MFMAs;
//  * column tiles are PAIRED with interleaved columns (tile 2p holds the even, tile 2p+1 the odd columns of the 32-column
//    group p - a permutation applied once, when S is copied into LDS), so a lane's results for a pair are two ADJACENT
//    columns of one row: one 16-byte store per pair, rows written in 256-byte runs.  An unpaired last tile keeps 8-byte stores.
// Per 32 rows: KS loads and 8 ceil(NT / 2) stores per 2 NT KS MFMAs (n = 100: 57 instead of 106).
// Wave ids are dealt w-major (id = w gridDim + block), so that the waves with one tile more than the others sit on different
// SIMDs of a workgroup.
typedef double d2u_t __attribute__((ext_vector_type(2), aligned(8)));  // 16-byte store to an 8-byte aligned address (odd n)
// A RAGGED LAST COLUMN TILE (n not a multiple of 16) runs on v_mfma_f64_4x4x4_4b_f64: its A fragment has the SAME lane
// layout as the 16x16x4 one (A[blk][i][k] in lane 16 k + 4 blk + i = 16 k + row, measured: lz_reorth.hip), its four blocks are
// the four 4-row groups of the tile against ONE 4-column panel of S (B[blk][k][j] in lane 16 k + 4 blk + j: the same
// S[4 t + k][c + j] in every block), and it occupies the matrix pipe for 16 cycles instead of 64 (PMC: 16 busy cycles per 256
// MACs on either shape).  n = 100 = 6 x 16 + 4: 6 x 64 + 16 = 400 cycles per k-step and row tile - the ideal 100 / 16 x 64 -
// instead of 7 x 64 = 448.  R4 = ceil(r / 4) such panels replace the last tile when its r columns are <= 12.
template <int NT, int KS, bool USE4 = true>
__global__ __launch_bounds__(512) void k_gemm_tn_sl2(const double* __restrict__ A, int64_t lda, int64_t mdim, int kcount,
                                                    const double* __restrict__ B, int64_t ldb, int ncols,
                                                    double* __restrict__ C, int64_t ldc, unsigned long long* __restrict__ clk) {
  constexpr int NW = 8;
  constexpr int R4 = (4 * NT == KS || !USE4) ? 0 : KS - 4 * (NT - 1);  // KS = ceil(n / 4): 4 (NT - 1) + ceil(r / 4)
  constexpr int NF = R4 ? NT - 1 : NT;                        // full 16-column tiles
  constexpr int NP = NF / 2;                                  // paired column tiles
  constexpr int FR = NF + R4;                                 // B fragments per k-step
  constexpr int PA = NF >= 7 ? 6 : 8;  // 16-byte V fragments in flight per wave (7-8 full tiles: 224+ accumulator registers leave room for 6)
  // (Four instantiations - 7 full column tiles: n = 105-112 and 117-124 - spill 3-6 registers: loop-invariant pointers, one save /
  // reload per 32-row tile.  Round 4 tried shallower rings (PA = 6 / 4): the allocator lands on the same 256 + spills whatever the
  // ring depth.  Round 5 tried ONE set of B fragments refilled in place behind the MFMAs that consumed them (no second set, 14
  // registers less on paper): without the scheduling fences around a whole k-step the compiler hoists the LDS reads of several
  // k-steps and spills 1.4 KB per lane.  Left as is - a 16-28-byte scratch round trip per ~400 MFMAs.)
  extern __shared__ double sS[];  // [KS][FR][64]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  if (clk != nullptr && lane == 0) {
    const unsigned long long now = wall_clock64();
    atomicMin(clk + 4, now);                        // first wave in (kernel-internal span)
    if (blockIdx.x == 0) atomicMin(clk + 6, now);   // workgroup 0 in
  }
  {
    constexpr int NFR = (KS * FR + NW - 1) / NW;
    double tmp[NFR];
#pragma unroll
    for (int i = 0; i < NFR; ++i) {
      const int f = w + i * NW;
      const int t = f / FR, q = f - t * FR;
      // full tiles: paired ones hold interleaved columns; the 4-column panels: column c + (lane & 3) in every block
      const int col = q >= NF ? 16 * NF + 4 * (q - NF) + (lane & 3) : ((q | 1) < NF ? 32 * (q >> 1) + 2 * lr + (q & 1) : 16 * q + lr);
      tmp[i] = f < KS * FR ? B[(int64_t)(4 * t + lk) * ldb + col] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < NFR; ++i) {
      const int f = w + i * NW;
      if (f < KS * FR) sS[f * 64 + lane] = tmp[i];
    }
  }
  const bool rec = clk != nullptr && blockIdx.x == 0 && __builtin_amdgcn_readfirstlane(w) == 0;
  uint64_t c0 = 0, t0 = 0;
  if (rec) {
    c0 = clock64();
    t0 = wall_clock64();
  }
  __syncthreads();
  const int64_t ntiles = (mdim + 31) / 32;
  const int64_t nwaves = (int64_t)gridDim.x * NW;
  const int64_t wave = (int64_t)w * gridDim.x + blockIdx.x;
  const int64_t stride4 = 4 * lda;
  auto tile_base = [&](int64_t tile) {
    const int64_t t = tile < ntiles ? tile : ntiles - 1;
    int64_t m = t * 32 + 2 * lr;
    m = m < mdim ? m : ((mdim - 1) & ~(int64_t)1);  // a pair past the end: a valid (even) address, results not stored
    return A + (int64_t)lk * lda + m;
  };
  auto frag = [&](const double* base, int t) {  // t is a compile-time constant at every call site
    if (t >= KS - 2) {
      int kr = 4 * t + lk;
      kr = kr < kcount ? kr : kcount - 1;
      return __builtin_nontemporal_load(reinterpret_cast<const d2v_t*>(base + (int64_t)(kr - lk) * lda));
    }
    return __builtin_nontemporal_load(reinterpret_cast<const d2v_t*>(base + (int64_t)t * stride4));
  };
  const double* cur = tile_base(wave);
  d2v_t ra[PA];
#pragma unroll
  for (int p = 0; p < PA; ++p) ra[p] = frag(cur, p < KS ? p : KS - 1);
  int64_t mytiles = 0;
  // One tile.  Written for the sake of the WAIT COUNTS: vmcnt counts loads and stores alike, in issue order, and the first
  // k-steps of a tile need their V fragment - requested before the previous tile's ~26 result stores were issued.  The right
  // wait is vmcnt(8 + those stores); what the compiler emits is vmcnt(8 + the stores it can PROVE were issued on every path
  // into the loop header), and vmcnt(8) sits out the acknowledgement of most of the previous tile's stores (measured with the
  // timing-only arms: the stores cost 0.13 ms of a 0.45 ms kernel that way, the loads 0.05 ms).  So: (a) the first tile is
  // peeled (both edges into the header then carry the same history), (b) the one row-ragged tile of the whole matrix runs
  // after the loop, (c) in a whole tile every store whose columns are valid for every n of this instantiation (n >= 4 KS - 3)
  // is unconditional - no exec-mask branch around it, so it counts.
  constexpr int NVALID = 4 * KS - 3;
  auto do_tile = [&](int64_t tile, auto whole_tag) {
    constexpr bool WHOLE = decltype(whole_tag)::value;
    const double* nxt = tile_base(tile + nwaves);
    double4_t accE[NF > 0 ? NF : 1], accO[NF > 0 ? NF : 1];
    double acc4E[R4 > 0 ? R4 : 1], acc4O[R4 > 0 ? R4 : 1];
#pragma unroll
    for (int b = 0; b < NF; ++b) {
      accE[b] = (double4_t){0.0, 0.0, 0.0, 0.0};
      accO[b] = (double4_t){0.0, 0.0, 0.0, 0.0};
    }
#pragma unroll
    for (int i = 0; i < R4; ++i) acc4E[i] = acc4O[i] = 0.0;
    double bcur[FR];
#pragma unroll
    for (int q = 0; q < FR; ++q) bcur[q] = sS[q * 64 + lane];
#pragma unroll
    for (int t = 0; t < KS; ++t) {
      const d2v_t a = ra[t % PA];
      if (PA < KS) ra[t % PA] = t + PA < KS ? frag(cur, t + PA) : frag(nxt, t + PA - KS);
      double bnxt[FR];
      if (t + 1 < KS) {
#pragma unroll
        for (int q = 0; q < FR; ++q) bnxt[q] = sS[((t + 1) * FR + q) * 64 + lane];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < NF; ++b) {
        accE[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, bcur[b], accE[b], 0, 0, 0);
        accO[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, bcur[b], accO[b], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < R4; ++i) {
        acc4E[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.x, bcur[NF + i], acc4E[i], 0, 0, 0);
        acc4O[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.y, bcur[NF + i], acc4O[i], 0, 0, 0);
      }
      if (t + 1 < KS) {
#pragma unroll
        for (int q = 0; q < FR; ++q) bcur[q] = bnxt[q];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (PA >= KS) {
#pragma unroll
      for (int p = 0; p < PA; ++p) ra[p] = frag(nxt, p < KS ? p : KS - 1);
    } else if constexpr (KS % PA != 0) {
      d2v_t tmp[PA];
#pragma unroll
      for (int i = 0; i < PA; ++i) tmp[i] = ra[(i + KS) % PA];
#pragma unroll
      for (int i = 0; i < PA; ++i) ra[i] = tmp[i];
    }
    cur = nxt;
    // results: accE[b][g] = Y[m0 + 2 (lk + 4 g)][col(b, lr)], accO: the row below it
    const int64_t m0 = tile * 32;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int64_t rE = m0 + 2 * (lk + 4 * g);
      double* pE = C + rE * ldc;
      double* pO = pE + ldc;
      const bool okE = WHOLE || rE < mdim, okO = WHOLE || rE + 1 < mdim;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int col = 32 * p + 2 * lr;
        if (32 * p + 31 < NVALID || col + 1 < ncols) {
          if (okE) __builtin_nontemporal_store((d2u_t){accE[2 * p][g], accE[2 * p + 1][g]}, reinterpret_cast<d2u_t*>(pE + col));
          if (okO) __builtin_nontemporal_store((d2u_t){accO[2 * p][g], accO[2 * p + 1][g]}, reinterpret_cast<d2u_t*>(pO + col));
        } else if (col < ncols) {
          if (okE) __builtin_nontemporal_store(accE[2 * p][g], pE + col);
          if (okO) __builtin_nontemporal_store(accO[2 * p][g], pO + col);
        }
      }
      if constexpr (NF & 1) {
        const int col = 16 * (NF - 1) + lr;
        if (16 * NF - 1 < NVALID || col < ncols) {
          if (okE) __builtin_nontemporal_store(accE[NF - 1][g], pE + col);
          if (okO) __builtin_nontemporal_store(accO[NF - 1][g], pO + col);
        }
      }
    }
    if constexpr (R4 > 0) {  // the 4-column panels: D[blk][i][j] sits in lane 16 i + 4 blk + j: tile row 4 blk + i, column c + j
      const int64_t rE = m0 + 2 * (4 * ((lane >> 2) & 3) + (lane >> 4));
      double* pE = C + rE * ldc;
      double* pO = pE + ldc;
      const bool okE = WHOLE || rE < mdim, okO = WHOLE || rE + 1 < mdim;
#pragma unroll
      for (int i = 0; i < R4; ++i) {
        const int col = 16 * NF + 4 * i + (lane & 3);
        if (16 * NF + 4 * i + 3 < NVALID || col < ncols) {
          if (okE) __builtin_nontemporal_store(acc4E[i], pE + col);
          if (okO) __builtin_nontemporal_store(acc4O[i], pO + col);
        }
      }
    }
  };
  const int64_t nfull = mdim / 32;  // whole tiles; at most one more (ragged in rows) follows
  if (wave < nfull) {
    do_tile(wave, std::true_type{});
    ++mytiles;
    for (int64_t tile = wave + nwaves; tile < nfull; tile += nwaves, ++mytiles) do_tile(tile, std::true_type{});
  }
  if (nfull < ntiles && nfull % nwaves == wave) {  // (its fragments are what the ring holds: it is this wave's next tile)
    do_tile(nfull, std::false_type{});
    ++mytiles;
  }
  if (rec) {
    clk[0] = clock64() - c0;
    clk[1] = wall_clock64() - t0;
    clk[2] = (unsigned long long)(2 * mytiles);  // in 16-row tiles
    clk[3] = (unsigned long long)((4 * NF + R4) * KS * 2);  // x 16 = the SIMD's matrix-pipe cycles per 16-row tile of each of its two waves
  }
  if (clk != nullptr && lane == 0) {
    const unsigned long long now = wall_clock64();
    atomicMax(clk + 5, now);
    if (blockIdx.x == 0) atomicMax(clk + 7, now);  // workgroup 0 out: with [6], the span the whole workgroup (not only its oldest wave) needed
    if (blockIdx.x < 256) atomicMax(clk + 8 + blockIdx.x, now);
  }
}

// Ritz back-transform, S-STATIONARY kernel (variant 5; n in 193..208): S lives in registers, V streams through LDS.
// Why.  The kernels above keep a tile of Y in the accumulators and re-read S every k-step: 13 operand fetches per 26 MFMAs,
// and the MFMA stream with that fetch is what takes the time (15.5 of 16.1 ms).  Turned around, the
// stationary operand is S: one workgroup of EIGHT waves (two per SIMD, 256 registers each) holds all of S in registers.
// Waves (s, 0) and (s, 1) share SIMD s and the column panels s, s+4, s+8: wave (s, h) keeps their k-steps
// [h KS/2, (h+1) KS/2) (3 x 25 doubles per lane at n = 200); wave (s, 1) also keeps the k-steps [KS s/4, KS (s+1)/4) of
// the 13th panel (a K-split: every SIMD gets 3.25 panel-products per row tile, so the four MFMA pipes carry equal work).
// The only operand that moves is a 16-row tile of V (25.6 KB): one LDS-DMA copy per workgroup into a double-buffered LDS
// tile in MFMA-fragment order (a k-step's fragment is 64 consecutive doubles: conflict-free ds_read_b64); every fragment
// read feeds three or four MFMAs.
// Why two waves per SIMD: a vector-memory instruction costs the issuing wave 160-280 cycles among FP64 MFMAs
// (profiles/r02/ritz_sreg_ablation.json: with one wave per SIMD the 20 loads and stores of a tile cost 3 300 of 14 000
// cycles whatever their width, placement or cache policy), and the only thing that keeps the MFMA pipe fed meanwhile is
// a second wave.  So the memory work is split by kind: the (s, 1) waves issue the LDS-DMA of the next tile and hand
// their partial tiles over through LDS; the (s, 0) waves add them to their own partial tiles and store the previous
// tile's results, one store per k-step, between their MFMAs.  One raw barrier per tile with counted waits.
// Stores are unconditional: C needs (round_up(mdim, 16) + 16) rows (the last 16 take the "previous tile" of a
// workgroup's first trip).
template <int NT, int KS>
__global__ __launch_bounds__(512) void k_gemm_tn_sreg(const double* __restrict__ A, int64_t lda, int64_t mdim, int kcount,
                                                     const double* __restrict__ B, int64_t ldb, int ncols,
                                                     double* __restrict__ C, int64_t ldc, unsigned long long* __restrict__ clk) {
  constexpr int FULL = NT / 4, REM = NT % 4, KH = KS / 2, KQ = (KS + 3) / 4;
  constexpr int TILE = 64 * KS;                  // doubles per staged tile: 4 KS basis rows x 16 matrix rows
  constexpr int NPC = TILE / 2 / 64;             // LDS-DMA instructions (64 x 16 bytes) per tile
  constexpr int NLD = (NPC + 3) / 4;             // per loader wave
  constexpr int IMG = FULL * 4 * 256;            // doubles per partial image: [wave s][panel j][reg g][lane]
  constexpr int PART = (REM > 0 ? REM : 1) * 4 * 256;  // K-split partials: [panel][wave s][reg][lane]
  static_assert(KS % 2 == 0 && FULL >= 1 && 2 + 4 * FULL + REM <= KH, "shape");  // the storing waves' riders need one k-step each
  extern __shared__ double lds_sreg[];
  double* vt = lds_sreg;
  double* img = lds_sreg + 2 * TILE;
  double* part = img + 2 * IMG;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, s = w & 3, h = w >> 2;
  const int lr = lane & 15, lk = lane >> 4;
  const int kh0 = h * KH;
  double sf[FULL][KH];
#pragma unroll
  for (int j = 0; j < FULL; ++j)
#pragma unroll
    for (int t = 0; t < KH; ++t) sf[j][t] = B[(int64_t)(4 * (kh0 + t) + lk) * ldb + 16 * (s + 4 * j) + lr];
  const int64_t ntiles = (mdim + 15) / 16;
  const int64_t per = (ntiles + gridDim.x - 1) / gridDim.x;
  const int64_t first = (int64_t)blockIdx.x * per;
  const int64_t last = first + per < ntiles ? first + per : ntiles;
  if (first >= last) return;  // uniform over the block
  // In-kernel clock record (lz_ritz_info): shader-clock cycles (s_memtime) against the constant 100 MHz counter
  // (s_memrealtime) over this workgroup's whole trip, taken by one wave of workgroup 0 - two scalar reads per launch.
  const bool rec = clk != nullptr && blockIdx.x == 0 && __builtin_amdgcn_readfirstlane(w) == 0;  // wave-uniform: lives in SGPRs
  uint64_t kb_c0 = 0, kb_t0 = 0;
  if (rec) {
    kb_c0 = clock64();
    kb_t0 = wall_clock64();
  }
  constexpr int PF = 4;  // fragments are read from LDS PF k-steps ahead of their MFMAs (explicit ring, fully unrolled loops)
  if (h == 1) {
    // ---------------- loader waves: second K half, the K-split panels, the LDS-DMA
    __builtin_amdgcn_s_setprio(3);  // the longer of a SIMD's two waves (87.5 of 162.5 MFMAs plus the LDS-DMA): its MFMAs go first
    const int q0 = KS * s / 4, q1 = KS * (s + 1) / 4;
    double sr[REM > 0 ? REM : 1][KQ];
#pragma unroll
    for (int i = 0; i < REM; ++i)
#pragma unroll
      for (int t = 0; t < KQ; ++t) sr[i][t] = q0 + t < q1 ? B[(int64_t)(4 * (q0 + t) + lk) * ldb + 16 * (4 * FULL + i) + lr] : 0.0;
    // LDS-DMA map: instruction p = s + 4 i moves the 16-byte pieces [64 p, 64 p + 64) of the tile: lane -> basis row
    // 8 p + lane / 8 (clamped: S has zero rows past kcount), chunk lane % 8 of that row's 128 bytes.  Addresses are
    // formed when used (no registers held across the MFMA loop).
    // LDS-DMA map: instruction p = s + 4 i moves the 16-byte pieces [64 p, 64 p + 64) of the tile: lane -> basis row
    // 8 p + lane / 8 (clamped: S has zero rows past kcount), chunk lane % 8 of that row's 128 bytes.  Addresses are
    // formed when used (no registers held across the MFMA loop).
    auto issue = [&](int64_t tile, double* buf) {
      const double* base = A + tile * 16 + 2 * (lane & 7);
      int lrow = lane >> 3;
      asm volatile("" : "+v"(lrow));  // keeps the per-piece addresses from being hoisted out of the tile loop (registers)
#pragma unroll
      for (int i = 0; i < NLD; ++i)
        if (s + 4 * i < NPC) {
          int row = 8 * (s + 4 * i) + lrow;
          if (row >= kcount) row = kcount - 1;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (int64_t)row * lda),
                                           (__attribute__((address_space(3))) void*)(buf + 128 * (s + 4 * i)), 16, 0, 0);
        }
    };
    issue(first, vt);
    __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    int it = 0;
    for (int64_t tile = first; tile < last; ++tile, ++it) {
      const double* vb = vt + (it & 1) * TILE;
      double* ib = img + (it & 1) * IMG;
      double* pb = part + (it & 1) * PART;
      double fr[PF];
#pragma unroll
      for (int p = 0; p < PF; ++p) fr[p] = vb[64 * (kh0 + (p < KH ? p : KH - 1)) + lane];
      double ar = REM > 0 ? vb[64 * q0 + lane] : 0.0;
      __builtin_amdgcn_sched_barrier(0);
      issue(tile + 1 < last ? tile + 1 : last - 1, vt + ((it + 1) & 1) * TILE);  // all at once, first thing: measured no slower than spread over the k-steps or issued two tiles ahead
      __builtin_amdgcn_sched_barrier(0);
      double4_t acc[FULL], accr[REM > 0 ? REM : 1];
#pragma unroll
      for (int j = 0; j < FULL; ++j) acc[j] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int i = 0; i < REM; ++i) accr[i] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int t = 0; t < KH; ++t) {
        const double a = fr[t % PF];
        if (t + PF < KH) fr[t % PF] = vb[64 * (kh0 + t + PF) + lane];
        __builtin_amdgcn_sched_barrier(0);  // the scheduler would sink the LDS read down to its use, PF k-steps later
#pragma unroll
        for (int j = 0; j < FULL; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sf[j][t], acc[j], 0, 0, 0);
        if (REM > 0 && (t & 1) == 0 && (t >> 1) < KQ) {  // the K-split panel rides along: one more independent chain
          const double arc = ar;
          if ((t >> 1) + 1 < KQ) ar = vb[64 * (q0 + (t >> 1) + 1) + lane];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < REM; ++i) accr[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(arc, sr[i][t >> 1], accr[i], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int j = 0; j < FULL; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) ib[((s * FULL + j) * 4 + g) * 64 + lane] = acc[j][g];
#pragma unroll
      for (int i = 0; i < REM; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) pb[((i * 4 + s) * 4 + g) * 64 + lane] = accr[i][g];
      __builtin_amdgcn_s_waitcnt(0x0070);  // the next tile has landed (vmcnt(0)), the partial tiles are written (lgkmcnt(0))
      __builtin_amdgcn_s_barrier();
    }
  } else {
    // ---------------- storing waves: first K half, the previous tile's results
    static_assert(KQ <= (KH + 1) / 2 + 1, "K-split slots");
    __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();
    double4_t pacc[FULL];
#pragma unroll
    for (int j = 0; j < FULL; ++j) pacc[j] = (double4_t){0.0, 0.0, 0.0, 0.0};
    int64_t pm0 = ntiles * 16;  // the slack rows
    const int colr = 16 * (4 * FULL) + lr;  // first K-split panel's column of this lane
    int it = 0;
    auto ksplit = [&](const double* pprev, int i) {  // wave s adds result register s (rows lk + 4 s) of the four K-split partial tiles
      double v = pprev[((i * 4 + 0) * 4 + s) * 64 + lane];
      v += pprev[((i * 4 + 1) * 4 + s) * 64 + lane];
      v += pprev[((i * 4 + 2) * 4 + s) * 64 + lane];
      v += pprev[((i * 4 + 3) * 4 + s) * 64 + lane];
      return v;
    };
    for (int64_t tile = first; tile < last; ++tile, ++it) {
      const double* vb = vt + (it & 1) * TILE;
      const double* iprev = img + ((it + 1) & 1) * IMG + s * FULL * 256 + lane;
      const double* pprev = part + ((it + 1) & 1) * PART;
      double* cprev = C + pm0 * ldc;
      double fr[PF];
#pragma unroll
      for (int p = 0; p < PF; ++p) fr[p] = vb[64 * (p < KH ? p : KH - 1) + lane];
      double* prow[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) prow[g] = cprev + (lk + 4 * g) * ldc + 16 * s + lr;
      double other = iprev[0];  // the (s, 1) wave's partial of result 0: read one k-step before it is needed
      __builtin_amdgcn_sched_barrier(0);
      double4_t acc[FULL];
#pragma unroll
      for (int j = 0; j < FULL; ++j) acc[j] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int t = 0; t < KH; ++t) {
        const double a = fr[t % PF];
        if (t + PF < KH) fr[t % PF] = vb[64 * (t + PF) + lane];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < FULL; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sf[j][t], acc[j], 0, 0, 0);
        // ---- riders: the previous tile's results, one store instruction per k-step
        if (t >= 2 && t < 2 + 4 * FULL) {
          const int r = t - 2, j = r >> 2, g = r & 3;
          const double v = pacc[j][g] + other;
          if (r + 1 < 4 * FULL) other = iprev[(r + 1) * 64];
          // (with REM == 0 the LAST full panel group can be ragged: columns >= ncols belong to the next row of C)
          if (REM > 0 || j + 1 < FULL || 16 * (s + 4 * j) + lr < ncols) __builtin_nontemporal_store(v, prow[g] + 64 * j);
        }
        if (t >= 2 + 4 * FULL && t < 2 + 4 * FULL + REM) {
          const int i = t - 2 - 4 * FULL;
          const double v = ksplit(pprev, i);
          if (colr + 16 * i < ncols) __builtin_nontemporal_store(v, cprev + (lk + 4 * s) * ldc + colr + 16 * i);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int j = 0; j < FULL; ++j) pacc[j] = acc[j];
      pm0 = tile * 16;
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) only: the stores stay in flight
      __builtin_amdgcn_s_barrier();
    }
    // the last tile's results
    {
      const double* iprev = img + ((it + 1) & 1) * IMG + s * FULL * 256 + lane;
      const double* pprev = part + ((it + 1) & 1) * PART;
      double* cprev = C + pm0 * ldc;
#pragma unroll
      for (int j = 0; j < FULL; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          if (REM > 0 || j + 1 < FULL || 16 * (s + 4 * j) + lr < ncols)
            __builtin_nontemporal_store(pacc[j][g] + iprev[(j * 4 + g) * 64], cprev + (lk + 4 * g) * ldc + 16 * (s + 4 * j) + lr);
#pragma unroll
      for (int i = 0; i < REM; ++i) {
        const double v = ksplit(pprev, i);
        if (colr + 16 * i < ncols) __builtin_nontemporal_store(v, cprev + (lk + 4 * s) * ldc + colr + 16 * i);
      }
    }
  }
  if (rec) {
    clk[0] = clock64() - kb_c0;
    clk[1] = wall_clock64() - kb_t0;
    clk[2] = (unsigned long long)(last - first);
    clk[3] = (unsigned long long)(NT * KS);  // MFMAs per 16-row tile over the four SIMDs
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

// S-stationary launcher: one instantiation per (column tiles NT, k-steps KS).  More than 64 KiB of dynamic LDS needs the
// per-kernel limit raised (once per instantiation; the result is checked).
template <int NT, int KS>
static hipError_t launch_sreg(const double* V, int64_t ldv, int64_t rows, int n, const double* Spad, int npad, double* Y, int64_t ldy,
                              hipStream_t s, unsigned long long* clk) {
  constexpr int FULL = NT / 4, REM = NT % 4;
  constexpr size_t lds = (size_t)(2 * 64 * KS + 2 * FULL * 4 * 256 + 2 * (REM > 0 ? REM : 1) * 4 * 256) * sizeof(double);
  static hipError_t attr = hipErrorNotReady;
  if (attr == hipErrorNotReady)
    attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_tn_sreg<NT, KS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (attr != hipSuccess) return attr;
  hipLaunchKernelGGL((k_gemm_tn_sreg<NT, KS>), dim3(kNumCU), dim3(512), lds, s, V, ldv, rows, n, Spad, (int64_t)npad, n, Y, ldy, clk);
  return hipSuccess;
}


template <int NT, int KS, bool USE4 = true>
static hipError_t launch_sl2(const double* V, int64_t ldv, int64_t rows, int n, const double* Spad, int npad, double* Y, int64_t ldy,
                             hipStream_t s, unsigned long long* clk) {
  constexpr int R4 = (4 * NT == KS || !USE4) ? 0 : KS - 4 * (NT - 1);
  constexpr size_t lds = (size_t)KS * ((R4 ? NT - 1 : NT) + R4) * 64 * sizeof(double);
  static hipError_t attr = hipErrorNotReady;
  if (attr == hipErrorNotReady)
    attr = lds > 65536 ? hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_tn_sl2<NT, KS, USE4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                       : hipSuccess;
  if (attr != hipSuccess) return attr;
  const int64_t ntiles = (rows + 31) / 32;
  const int grid = (int)std::min<int64_t>(kNumCU, (ntiles + 7) / 8);
  hipLaunchKernelGGL((k_gemm_tn_sl2<NT, KS, USE4>), dim3(grid), dim3(512), lds, s, V, ldv, rows, n, Spad, (int64_t)npad, n, Y, ldy, clk);
  return hipSuccess;
}

// The ragged last column tile runs on the 4x4x4 MFMA (USE4) only for NT >= 7 (n >= 97), where the kernel is bound by the
// matrix pipe: measured in one process (profiles/r03), n = 100: 0.394 vs 0.416 ms, n = 117: 0.566 vs 0.610 ms with it; but
// n = 37 / 50 / 70, which are bound by HBM: 0.216 / 0.237 / 0.349 ms with it against 0.181 / 0.200 / 0.279 ms without - a
// 4-column panel is stored in 32-byte pieces (16 rows x 4 columns per instruction), partial sectors on the write side.
static bool sl2_dispatch(const double* V, int64_t ldv, int64_t rows, int n, const double* Spad, int npad, double* Y, int64_t ldy,
                         hipStream_t s, unsigned long long* clk, hipError_t* err) {
  const int NT = (n + 15) / 16;
  const int KS = (n + 3) / 4;
#define LZ_SL(nt, ks)                                                                     \
  if (NT == nt && KS == ks) {                                                             \
    *err = launch_sl2<nt, ks, (nt >= 7)>(V, ldv, rows, n, Spad, npad, Y, ldy, s, clk);     \
    return true;                                                                          \
  }
#define LZ_SL4(nt) LZ_SL(nt, 4 * nt - 3) LZ_SL(nt, 4 * nt - 2) LZ_SL(nt, 4 * nt - 1) LZ_SL(nt, 4 * nt)
  LZ_SL4(1) LZ_SL4(2) LZ_SL4(3) LZ_SL4(4) LZ_SL4(5) LZ_SL4(6) LZ_SL4(7) LZ_SL4(8)
#undef LZ_SL4
#undef LZ_SL
  return false;
}

// The S-stationary kernel covers 129 <= n <= 200 (below that S fits the LDS: k_gemm_tn_sl): NT = ceil(n / 16) column tiles
// (9..13) and KS = ceil(n / 4) k-steps rounded up to even (the two waves of a SIMD split them in halves); S is zero-padded
// to 16 NT rows and columns.
static bool sreg_dispatch(const double* V, int64_t ldv, int64_t rows, int n, const double* Spad, int npad, double* Y, int64_t ldy,
                          hipStream_t s, unsigned long long* clk, hipError_t* err) {
  const int NT = (n + 15) / 16;
  const int KS = (((n + 3) / 4) + 1) & ~1;
#define LZ_SR(nt, ks)                                                                  \
  if (NT == nt && KS == ks) {                                                          \
    *err = launch_sreg<nt, ks>(V, ldv, rows, n, Spad, npad, Y, ldy, s, clk);            \
    return true;                                                                       \
  }
  LZ_SR(9, 34) LZ_SR(9, 36) LZ_SR(10, 38) LZ_SR(10, 40) LZ_SR(11, 42) LZ_SR(11, 44) LZ_SR(12, 46) LZ_SR(12, 48) LZ_SR(13, 50)
#undef LZ_SR
  return false;
}

#ifdef LZ_KBENCH  // the retired Ritz GEMM kernels and their dispatch: a kernel-bench-only file
#include "lz_gemm_kbench.h"
#endif

hipError_t launch_ritz_gemm(const double* V, int64_t ldv, int64_t rows, int n, const double* Spad, int npad, double* Y,
                            int64_t ldy, hipStream_t s, int variant, unsigned long long* clk) {
  const int64_t ntiles = (rows + 31) / 32;
  // 0 (auto): 33 <= n <= 128 and at least 4096 rows: S resident in LDS (k_gemm_tn_sl2, 32-row tiles; kernel-bench build: 6 = the
  // superseded 16-row-tile form);
  // 129 <= n <= 200: the S-stationary kernel where it applies (enough row tiles for a persistent grid, 16-byte aligned
  // rows); else one workgroup per 128 rows.  1 forces the latter.
  const bool sreg_ok = n > 128 && n <= 200 && (ldv & 1) == 0 && (reinterpret_cast<uintptr_t>(V) & 15) == 0 && npad >= 16 * ((n + 15) / 16) &&
                       ntiles >= 2 * kNumCU * (kTPB / 64);
  const bool sl_ok = n > 32 && n <= 128 && rows >= 4096 && npad >= 16 * ((n + 15) / 16);  // (n <= 32 is a pure stream: the many short-lived workgroups of k_gemm_tn keep more loads in flight)
#ifdef LZ_KBENCH
  if (hipError_t e; kbench_ritz_arm(V, ldv, rows, n, Spad, npad, Y, ldy, s, variant, &e)) return e;
#endif
  if (variant != 1 && sl_ok) {
    hipError_t e = hipSuccess;
    const bool pairs_ok = (ldv & 1) == 0 && (reinterpret_cast<uintptr_t>(V) & 15) == 0;  // 16-byte loads of V
#ifdef LZ_KBENCH
    if (variant == 6 && sl_dispatch<2>(V, ldv, rows, n, Spad, npad, Y, ldy, s, clk, &e)) return e;
#endif
    if (pairs_ok && sl2_dispatch(V, ldv, rows, n, Spad, npad, Y, ldy, s, clk, &e)) return e;
  }
  if (variant != 1 && sreg_ok) {
    hipError_t e = hipSuccess;
    if (sreg_dispatch(V, ldv, rows, n, Spad, npad, Y, ldy, s, clk, &e)) return e;
  }
  launch_gemm_tn(V, ldv, rows, n, n, 1, Spad, npad, n, Y, ldy, 0, s);
  return hipSuccess;
}

// columns [0, ncols) of V^T-layout x B for all rows, B = a column block of the padded S (chunked Ritz-vector quality)
void launch_ritz_gemm_cols(const double* V, int64_t ldv, int64_t rows, int kcount, const double* B, int ldb, int ncols, double* Y,
                           int64_t ldy, hipStream_t s) {
  launch_gemm_tn(V, ldv, rows, kcount, kcount, 1, B, ldb, ncols, Y, ldy, 0, s);
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

// ------------------------------------------------------------------ Gram matrix G = Y^T Y, accumulator-stationary (round 4)
// The two checks of get_H_eigs (Lanczos.py:157-158, 288-323) need the n x n Gram matrix of the Ritz vectors.  Until round 3
// this was k_gemm_tn with A = B = Y: every wave owned 32 columns x all n columns and re-read the whole row slab of Y per
// k-step, all n^2 entries were computed, 17.7 ms at the headline.  G is symmetric and, in the MFMA's lane layout, the A
// fragment of column tile i IS the B fragment of column tile i (lane (lr, lk) holds Y[k + lk][16 i + lr] either way).  So:
// a workgroup of four waves (one per SIMD) owns a K range of rows of Y; per k-step every wave loads the CT fragments of the
// 4-row slab (one contiguous 4 n doubles: Y is row-major) through a ring that runs PA k-steps ahead, and the CT (CT + 1) / 2
// upper-triangular 16 x 16 tiles of G - dealt round-robin to the waves at compile time - stay in the accumulators for the
// whole K range: 91 MFMAs per k-step at n = 200 instead of 182, no operand other than the slab, Y streamed once.
// Split-K only across workgroups: every workgroup leaves its upper tiles in its own n x n slice, k_sum_slices_sym adds the
// slices in fixed order and mirrors the result (G[i][j] == G[j][i] bit for bit).
constexpr int gram_tri_count(int CT) { return CT * (CT + 1) / 2; }
constexpr int gram_tri_row(int CT, int t) {
  int i = 0;
  while (t >= CT - i) {
    t -= CT - i;
    ++i;
  }
  return i;
}
constexpr int gram_tri_col(int CT, int t) {
  int i = 0;
  while (t >= CT - i) {
    t -= CT - i;
    ++i;
  }
  return i + t;
}

template <int CT, int W, int Q, int NW = 4>
struct GramTile {  // the Q-th tile of wave W of NW: indices as compile-time constants (register arrays must never be indexed at run time)
  static constexpr int t = W + NW * Q;
  static constexpr int i = gram_tri_row(CT, t), j = gram_tri_col(CT, t);
};
template <int CT, int W, int NW, int... Q>
__device__ __forceinline__ void gram_mfmas(const double (&f)[CT], double4_t* acc, std::integer_sequence<int, Q...>) {
  ((acc[Q] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[GramTile<CT, W, Q, NW>::i], f[GramTile<CT, W, Q, NW>::j], acc[Q], 0, 0, 0)), ...);
}
template <int CT, int W, int Q, int NW>
__device__ __forceinline__ void gram_store_tile(const double4_t& a, int n, int lr, int lk, double* __restrict__ Cz) {
  const int col = 16 * GramTile<CT, W, Q, NW>::j + lr;
  if (col >= n) return;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int r = 16 * GramTile<CT, W, Q, NW>::i + lk + 4 * g;
    if (r < n) Cz[(int64_t)r * n + col] = a[g];
  }
}
template <int CT, int W, int NW, int... Q>
__device__ __forceinline__ void gram_store(const double4_t* acc, int n, int lr, int lk, double* __restrict__ Cz, std::integer_sequence<int, Q...>) {
  (gram_store_tile<CT, W, Q, NW>(acc[Q], n, lr, lk, Cz), ...);
}

template <int CT, int W, int NW = 4>
__device__ __forceinline__ void gram_wave(const double* __restrict__ Y, int64_t ldy, int64_t k_lo, int64_t k_hi, int n,
                                          double* __restrict__ Cz) {
  constexpr int NTRI = gram_tri_count(CT);
  constexpr int NTW = (NTRI - W + NW - 1) / NW;  // this wave's tiles: t = W, W + NW, ...
  constexpr int PA = 4;                    // k-steps the slab loads run ahead of the MFMAs
  const int lane = threadIdx.x & 63;
  const int lr = lane & 15, lk = lane >> 4;
  double4_t acc[NTW > 0 ? NTW : 1];
#pragma unroll
  for (int q = 0; q < NTW; ++q) acc[q] = (double4_t){0.0, 0.0, 0.0, 0.0};
  const int nsteps = k_hi > k_lo ? (int)((k_hi - k_lo + 3) >> 2) : 0;
  double ring[PA][CT];
  auto load = [&](int step, double (&f)[CT]) {
    int64_t kr = k_lo + 4 * (int64_t)step + lk;
    const bool ok = kr < k_hi;  // rows past the range contribute zero (A and B are the same registers)
    if (!ok) kr = k_hi - 1;
    const double* row = Y + kr * ldy + lr;
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      const double v = row[16 * t];
      f[t] = ok ? v : 0.0;
    }
  };
  if (nsteps > 0) {
#pragma unroll
    for (int p = 0; p < PA; ++p) load(p < nsteps ? p : nsteps - 1, ring[p]);
  }
  for (int s0 = 0; s0 < nsteps; s0 += PA) {
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      const int st = s0 + p;
      if (st < nsteps) {
        gram_mfmas<CT, W, NW>(ring[p], acc, std::make_integer_sequence<int, NTW>());
        load(st + PA < nsteps ? st + PA : nsteps - 1, ring[p]);  // refill this ring slot
      }
    }
  }
  gram_store<CT, W, NW>(acc, n, lr, lk, Cz, std::make_integer_sequence<int, NTW>());
}

// (An eight-wave form - two waves per SIMD, 11-12 tiles each, so that one wave's fragment loads hide behind the other's MFMAs - was
// built and measured in round 4: 12.1 ms against 7.9-8.2 ms (profiles/r04/ab_gram_eight_waves.jsonl): every fragment is then fetched
// by twice as many waves, and the operand fetch is what this kernel is bound by.  Not kept.)
// clk (may be null; 4 words): in-kernel clock record of workgroup 0's wave 0, like the Ritz kernels' (lz_gram_info): shader
// cycles (s_memtime), ticks of the constant 100 MHz counter (s_memrealtime), k-steps walked, MFMAs per k-step of that wave.
template <int CT>
__global__ __launch_bounds__(kTPB) void k_gram_sym(const double* __restrict__ Y, int64_t ldy, int64_t rows, int64_t kchunk, int n,
                                                  double* __restrict__ part, unsigned long long* __restrict__ clk) {
  const int64_t k_lo = (int64_t)blockIdx.x * kchunk;
  const int64_t k_hi = k_lo + kchunk < rows ? k_lo + kchunk : rows;
  double* Cz = part + (int64_t)blockIdx.x * n * n;
  const bool rec = clk != nullptr && blockIdx.x == 0 && threadIdx.x < 64;
  unsigned long long c0 = 0, t0 = 0;
  if (rec) {
    c0 = clock64();
    t0 = wall_clock64();
  }
  switch (threadIdx.x >> 6) {  // wave-uniform: every wave runs its own fully unrolled tile list
    case 0: gram_wave<CT, 0>(Y, ldy, k_lo, k_hi, n, Cz); break;
    case 1: gram_wave<CT, 1>(Y, ldy, k_lo, k_hi, n, Cz); break;
    case 2: gram_wave<CT, 2>(Y, ldy, k_lo, k_hi, n, Cz); break;
    default: gram_wave<CT, 3>(Y, ldy, k_lo, k_hi, n, Cz); break;
  }
  if (rec && threadIdx.x == 0) {
    clk[0] = clock64() - c0;
    clk[1] = wall_clock64() - t0;
    clk[2] = (unsigned long long)((k_hi - k_lo + 3) >> 2);
    clk[3] = (unsigned long long)((gram_tri_count(CT) + 3) / 4);
  }
}

// ---- more than 13 column tiles (n > 208; BASELINE config C5: n = 500): the same accumulator-stationary scheme over GROUPS of
// kGramGS = 11 column tiles.  A workgroup owns a K range AND one unit of the upper triangle of groups: a diagonal unit (the
// triangle of one group, 66 tiles) or an off-diagonal pair (11 x 11 = 121 tiles).  Within a unit the tiles are dealt to the four
// waves BY ROW TILE - rows w, w + 4, w + 8 of a pair; a snake over the rows of a triangle (18 / 17 / 16 / 15 tiles) - so that a wave
// loads only the fragments of its own rows plus the column fragments: 14 loads for 33 MFMAs in a pair (the first, tile-by-tile
// deal of this round needed 17 loads for 17 MFMAs and was SLOWER than the split-K GEMM it replaces: 96.7 vs 84.9 ms at C5 - the
// operand fetch, not the matrix pipe, is what this kernel has to economise).  Tile indices are compile-time within a unit; the
// unit's tile offsets (ta0, tb0) come from a table.  Column tiles at or past ceil(n / 16) (the last group's padding) are loaded
// from the last real tile - valid memory - and their results are dropped by the store's bounds.
constexpr int kGramGS = 11;
constexpr int gram_snake_wave(int row) { return (row / 4) % 2 == 0 ? row % 4 : 3 - row % 4; }  // rows 0 1 2 3 | 3 2 1 0 | 0 1 2 3 ...
// DIAG: the Q-th tile of wave W walking its rows (snake) in increasing order, each row i over columns i .. NA-1
constexpr int gram_diag_count(int NA, int W) {
  int c = 0;
  for (int i = 0; i < NA; ++i)
    if (gram_snake_wave(i) == W) c += NA - i;
  return c;
}
constexpr int gram_diag_minrow(int NA, int W) {
  for (int i = 0; i < NA; ++i)
    if (gram_snake_wave(i) == W) return i;
  return NA;
}
constexpr int gram_diag_row(int NA, int W, int q) {
  for (int i = 0; i < NA; ++i)
    if (gram_snake_wave(i) == W) {
      if (q < NA - i) return i;
      q -= NA - i;
    }
  return 0;
}
constexpr int gram_diag_col(int NA, int W, int q) {
  for (int i = 0; i < NA; ++i)
    if (gram_snake_wave(i) == W) {
      if (q < NA - i) return i + q;
      q -= NA - i;
    }
  return 0;
}
template <int NA, int NB, bool DIAG, int W>
struct GramUnit {
  static constexpr int RW = (NA - W + 3) / 4;                               // pair: row tiles W, W + 4, ... of group a
  static constexpr int F0 = DIAG ? gram_diag_minrow(NA, W) : 0;            // triangle: fragments F0 .. NA-1 of the one group
  static constexpr int NF = DIAG ? NA - F0 : RW + NB;                       // fragments this wave loads per k-step
  static constexpr int NTW = DIAG ? gram_diag_count(NA, W) : RW * NB;       // tiles it owns
};
template <int NA, int NB, bool DIAG, int W, int Q>
struct GramUnitTile {
  using U = GramUnit<NA, NB, DIAG, W>;
  static constexpr int it = DIAG ? gram_diag_row(NA, W, Q) : W + 4 * (Q / NB);  // row tile within group a
  static constexpr int jt = DIAG ? gram_diag_col(NA, W, Q) : Q % NB;            // column tile within group b (DIAG: group a)
  static constexpr int fi = DIAG ? it - U::F0 : Q / NB;                          // their fragment slots
  static constexpr int fj = DIAG ? jt - U::F0 : U::RW + jt;
};
template <int NA, int NB, bool DIAG, int W, int NF, int... Q>
__device__ __forceinline__ void gram_unit_mfmas(const double (&f)[NF], double4_t* acc, std::integer_sequence<int, Q...>) {
  ((acc[Q] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[GramUnitTile<NA, NB, DIAG, W, Q>::fi], f[GramUnitTile<NA, NB, DIAG, W, Q>::fj], acc[Q], 0, 0, 0)), ...);
}
template <int NA, int NB, bool DIAG, int W, int Q>
__device__ __forceinline__ void gram_unit_store_tile(const double4_t& a, int n, int ta0, int tb0, int lr, int lk, double* __restrict__ Cz) {
  const int col = 16 * (tb0 + GramUnitTile<NA, NB, DIAG, W, Q>::jt) + lr;
  if (col >= n) return;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int r = 16 * (ta0 + GramUnitTile<NA, NB, DIAG, W, Q>::it) + lk + 4 * g;
    if (r < n) Cz[(int64_t)r * n + col] = a[g];
  }
}
template <int NA, int NB, bool DIAG, int W, int... Q>
__device__ __forceinline__ void gram_unit_store(const double4_t* acc, int n, int ta0, int tb0, int lr, int lk, double* __restrict__ Cz,
                                                std::integer_sequence<int, Q...>) {
  (gram_unit_store_tile<NA, NB, DIAG, W, Q>(acc[Q], n, ta0, tb0, lr, lk, Cz), ...);
}

template <int NA, int NB, bool DIAG, int W>
__device__ __forceinline__ void gram_unit_wave(const double* __restrict__ Y, int64_t ldy, int64_t k_lo, int64_t k_hi, int n, int CT, int ta0,
                                               int tb0, double* __restrict__ Cz) {
  using U = GramUnit<NA, NB, DIAG, W>;
  constexpr int NTW = U::NTW, NF = U::NF;
  // k-steps the fragment loads run ahead of the MFMAs: ~4 us worth of matrix-pipe time (a triangle's k-step is 15-18 MFMAs = half a
  // pair's: with 4 k-steps its loads were 2.3 us ahead and the unit ran at half its MFMA rate - profiles/r04/gram_groups_c5.txt)
  constexpr int PA = DIAG ? 8 : 4;
  const int lane = threadIdx.x & 63;
  const int lr = lane & 15, lk = lane >> 4;
  double4_t acc[NTW > 0 ? NTW : 1];
#pragma unroll
  for (int q = 0; q < NTW; ++q) acc[q] = (double4_t){0.0, 0.0, 0.0, 0.0};
  const int nsteps = k_hi > k_lo ? (int)((k_hi - k_lo + 3) >> 2) : 0;
  int off[NF > 0 ? NF : 1];  // element offsets of this wave's fragments within a row (tiles past the last real one alias it)
#pragma unroll
  for (int t = 0; t < NF; ++t) {
    const int tile = DIAG ? ta0 + U::F0 + t : (t < U::RW ? ta0 + W + 4 * t : tb0 + (t - U::RW));
    off[t] = 16 * (tile < CT ? tile : CT - 1);
  }
  double ring[PA][NF > 0 ? NF : 1];
  auto load = [&](int step, double (&f)[NF > 0 ? NF : 1]) {
    int64_t kr = k_lo + 4 * (int64_t)step + lk;
    const bool ok = kr < k_hi;
    if (!ok) kr = k_hi - 1;
    const double* row = Y + kr * ldy + lr;
#pragma unroll
    for (int t = 0; t < NF; ++t) {
      const double v = row[off[t]];
      f[t] = ok ? v : 0.0;
    }
  };
  if (nsteps > 0) {
#pragma unroll
    for (int p = 0; p < PA; ++p) load(p < nsteps ? p : nsteps - 1, ring[p]);
  }
  for (int s0 = 0; s0 < nsteps; s0 += PA) {
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      const int st = s0 + p;
      if (st < nsteps) {
        gram_unit_mfmas<NA, NB, DIAG, W, (NF > 0 ? NF : 1)>(ring[p], acc, std::make_integer_sequence<int, NTW>());
        load(st + PA < nsteps ? st + PA : nsteps - 1, ring[p]);
      }
    }
  }
  gram_unit_store<NA, NB, DIAG, W>(acc, n, ta0, tb0, lr, lk, Cz, std::make_integer_sequence<int, NTW>());
}

// One launch covers every unit of the upper triangle of groups (round 5): blockIdx.y < npairs -> an off-diagonal pair, else a
// triangle.  The pairs come first in the dispatch order (they are the long blocks: ceil(GS / 4) GS tiles on their busiest wave
// against a triangle's ~GS (GS + 1) / 8 + ...), the triangles fill the tail - two launches had two tails.
template <int GS>
__global__ __launch_bounds__(kTPB) void k_gram_units(const double* __restrict__ Y, int64_t ldy, int64_t rows, int64_t kchunk, int n, int CT,
                                                    const int2* __restrict__ units, int npairs, double* __restrict__ part) {
  const int64_t k_lo = (int64_t)blockIdx.x * kchunk;
  const int64_t k_hi = k_lo + kchunk < rows ? k_lo + kchunk : rows;
  double* Cz = part + (int64_t)blockIdx.x * n * n;
  const int2 u = units[blockIdx.y];  // {first row tile, first column tile}
  if ((int)blockIdx.y < npairs) {
    switch (threadIdx.x >> 6) {
      case 0: gram_unit_wave<GS, GS, false, 0>(Y, ldy, k_lo, k_hi, n, CT, u.x, u.y, Cz); break;
      case 1: gram_unit_wave<GS, GS, false, 1>(Y, ldy, k_lo, k_hi, n, CT, u.x, u.y, Cz); break;
      case 2: gram_unit_wave<GS, GS, false, 2>(Y, ldy, k_lo, k_hi, n, CT, u.x, u.y, Cz); break;
      default: gram_unit_wave<GS, GS, false, 3>(Y, ldy, k_lo, k_hi, n, CT, u.x, u.y, Cz); break;
    }
  } else {
    switch (threadIdx.x >> 6) {
      case 0: gram_unit_wave<GS, GS, true, 0>(Y, ldy, k_lo, k_hi, n, CT, u.x, u.y, Cz); break;
      case 1: gram_unit_wave<GS, GS, true, 1>(Y, ldy, k_lo, k_hi, n, CT, u.x, u.y, Cz); break;
      case 2: gram_unit_wave<GS, GS, true, 2>(Y, ldy, k_lo, k_hi, n, CT, u.x, u.y, Cz); break;
      default: gram_unit_wave<GS, GS, true, 3>(Y, ldy, k_lo, k_hi, n, CT, u.x, u.y, Cz); break;
    }
  }
}

// ---- round 5: the same two kernels with the row slab staged ONCE PER WORKGROUP through LDS -----------------------------------
// What bounded the register-ring forms above is the operand fetch: every wave fetched its fragments itself, through L1, four
// 128-byte lines per 8-byte-per-lane load (tools/probes/mfma_f64_operands: the same MFMA stream runs at 65 TF with operands loaded
// per k-step from L1/L2 and at 73 TF with operands read from LDS).  Here the workgroup copies a block of KB k-steps (4 KB rows of
// the unit's one or two column groups: contiguous runs of Y's rows) global -> registers -> LDS with coalesced 16-byte loads, double
// buffered - the registers of block b + 1 are in flight while block b is multiplied, one barrier per block - and every wave reads
// its fragments with ds_read_b64 (row stride = 128 bytes mod 256: the four rows of a fragment fall on disjoint banks, two passes
// of 32 lanes, conflict free), two k-steps of fragments in registers so that the reads hide behind the previous step's MFMAs.
// Tiles, their dealing to the waves, accumulators, stores and the slice sum are the kernels' above: bit-identical results
// (same MFMAs in the same order per tile).  Needs an even n (16-byte aligned rows); odd n keeps the register-ring kernels.
template <int GS, bool DIAG, int W>
struct GramLdsUnitDeal {
  using U = GramUnit<GS, GS, DIAG, W>;
  // (pairs of 10 or 11 column tiles keep 30 / 33 accumulator tiles per wave - 240 / 264 registers: two k-steps per block, so that the
  //  staging registers are few and nothing spills)
  static constexpr int NF = U::NF > 0 ? U::NF : 1, NTW = U::NTW, NSEG = DIAG ? 1 : 2, SEGD = 16 * GS, KB = DIAG ? 8 : (GS >= 10 ? 2 : 4);
  static constexpr int col(int t) { return DIAG ? 16 * (U::F0 + t) : (t < U::RW ? 16 * (W + 4 * t) : SEGD + 16 * (t - U::RW)); }
  template <int Q>
  struct Tile {
    using T = GramUnitTile<GS, GS, DIAG, W, Q>;
    static constexpr int fi = T::fi, fj = T::fj, it = T::it, jt = T::jt;
  };
};
template <int CT, int W>
struct GramLdsSymDeal {
  static constexpr int NTRI = gram_tri_count(CT);
  static constexpr int NF = CT, NTW = (NTRI - W + 3) / 4, NSEG = 1, SEGD = 16 * CT, KB = 8;
  static constexpr int col(int t) { return 16 * t; }
  template <int Q>
  struct Tile {
    using T = GramTile<CT, W, Q, 4>;
    static constexpr int fi = T::i, fj = T::j, it = T::i, jt = T::j;
  };
};
constexpr int gram_lds_stride(int rowd) { return (rowd * 8) % 256 == 128 ? rowd : rowd + 16; }  // doubles; 128 bytes mod 256
constexpr size_t gram_lds_bytes(int nseg, int segd, int kb) { return (size_t)2 * 4 * kb * gram_lds_stride(nseg * segd) * sizeof(double); }

template <class D, int... Q>
__device__ __forceinline__ void gram_lds_mfmas(const double (&f)[D::NF], double4_t* acc, std::integer_sequence<int, Q...>) {
  ((acc[Q] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[D::template Tile<Q>::fi], f[D::template Tile<Q>::fj], acc[Q], 0, 0, 0)), ...);
}
template <class D, int Q>
__device__ __forceinline__ void gram_lds_store_tile(const double4_t& a, int n, int ta0, int tb0, int lr, int lk, double* __restrict__ Cz) {
  const int col = 16 * (tb0 + D::template Tile<Q>::jt) + lr;
  if (col >= n) return;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int r = 16 * (ta0 + D::template Tile<Q>::it) + lk + 4 * g;
    if (r < n) Cz[(int64_t)r * n + col] = a[g];
  }
}
template <class D, int... Q>
__device__ __forceinline__ void gram_lds_store(const double4_t* acc, int n, int ta0, int tb0, int lr, int lk, double* __restrict__ Cz,
                                               std::integer_sequence<int, Q...>) {
  (gram_lds_store_tile<D, Q>(acc[Q], n, ta0, tb0, lr, lk, Cz), ...);
}

// One wave's whole life (staging included: all four waves of the workgroup run their own instantiation of this function and meet at
// the same barriers).  ca / cb: first column (doubles) of the one or two column groups in a row of Y.
template <class D>
__device__ __forceinline__ void gram_lds_wave(const double* __restrict__ Y, int64_t ldy, int64_t k_lo, int64_t k_hi, int n, int ca, int cb, int ta0,
                                              int tb0, double* __restrict__ Cz, double* __restrict__ lds) {
  constexpr int NF = D::NF, NTW = D::NTW, KB = D::KB, NROWB = 4 * KB;
  constexpr int ROWD = D::NSEG * D::SEGD, S = gram_lds_stride(ROWD);
  constexpr int ROW2 = ROWD / 2, SEG2 = D::SEGD / 2;   // double2 per staged row / per segment
  constexpr int ITEMS = NROWB * ROW2;                  // double2 per block
  constexpr int NR = (ITEMS + kTPB - 1) / kTPB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int lr = lane & 15, lk = lane >> 4;
  double4_t acc[NTW > 0 ? NTW : 1];
#pragma unroll
  for (int q = 0; q < NTW; ++q) acc[q] = (double4_t){0.0, 0.0, 0.0, 0.0};
  const int nblk = k_hi > k_lo ? (int)((k_hi - k_lo + NROWB - 1) / NROWB) : 0;
  double2 rg[NR];
  auto gload = [&](int64_t k0) {
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      const int i = tid + q * kTPB;
      rg[q] = make_double2(0.0, 0.0);
      if (ITEMS % kTPB == 0 || i < ITEMS) {
        const int row = i / ROW2, w = i - row * ROW2;
        const int seg = D::NSEG == 2 ? w / SEG2 : 0, c2 = w - seg * SEG2;
        const int64_t kr = k0 + row;
        if (kr < k_hi) rg[q] = *reinterpret_cast<const double2*>(Y + kr * ldy + (seg ? cb : ca) + 2 * c2);  // rows past the range: zero
      }
    }
  };
  auto sstore = [&](double* buf) {
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      const int i = tid + q * kTPB;
      if (ITEMS % kTPB == 0 || i < ITEMS) {
        const int row = i / ROW2, w = i - row * ROW2;
        *reinterpret_cast<double2*>(buf + row * S + 2 * w) = rg[q];
      }
    }
  };
  auto fload = [&](const double* buf, int s, double (&f)[NF]) {
    const double* base = buf + (4 * s + lk) * S + lr;
#pragma unroll
    for (int t = 0; t < NF; ++t) f[t] = base[D::col(t)];
  };
  double* buf0 = lds;
  double* buf1 = lds + NROWB * S;
  if (nblk > 0) {
    gload(k_lo);
    sstore(buf0);
  }
  __syncthreads();
  for (int b = 0; b < nblk; ++b) {
    if (b + 1 < nblk) gload(k_lo + (int64_t)(b + 1) * NROWB);  // in flight while this block is multiplied
    const double* cur = (b & 1) ? buf1 : buf0;
    double f0[NF], f1[NF];
    fload(cur, 0, f0);
#pragma unroll
    for (int s = 0; s < KB; s += 2) {
      fload(cur, s + 1, f1);
      gram_lds_mfmas<D>(f0, acc, std::make_integer_sequence<int, NTW>());
      if (s + 2 < KB) fload(cur, s + 2, f0);
      gram_lds_mfmas<D>(f1, acc, std::make_integer_sequence<int, NTW>());
    }
    if (b + 1 < nblk) sstore((b & 1) ? buf0 : buf1);
    __syncthreads();
  }
  gram_lds_store<D>(acc, n, ta0, tb0, lr, lk, Cz, std::make_integer_sequence<int, NTW>());
}

template <int GS>
__global__ __launch_bounds__(kTPB) void k_gram_units_lds(const double* __restrict__ Y, int64_t ldy, int64_t rows, int64_t kchunk, int n, int CT,
                                                        const int2* __restrict__ units, int npairs, double* __restrict__ part) {
  extern __shared__ double gram_lds[];
  const int64_t k_lo = (int64_t)blockIdx.x * kchunk;
  const int64_t k_hi = k_lo + kchunk < rows ? k_lo + kchunk : rows;
  double* Cz = part + (int64_t)blockIdx.x * n * n;
  const int2 u = units[blockIdx.y];  // {first row tile, first column tile}
  const int ca = 16 * u.x, cb = 16 * u.y;
  if ((int)blockIdx.y < npairs) {
    switch (threadIdx.x >> 6) {
      case 0: gram_lds_wave<GramLdsUnitDeal<GS, false, 0>>(Y, ldy, k_lo, k_hi, n, ca, cb, u.x, u.y, Cz, gram_lds); break;
      case 1: gram_lds_wave<GramLdsUnitDeal<GS, false, 1>>(Y, ldy, k_lo, k_hi, n, ca, cb, u.x, u.y, Cz, gram_lds); break;
      case 2: gram_lds_wave<GramLdsUnitDeal<GS, false, 2>>(Y, ldy, k_lo, k_hi, n, ca, cb, u.x, u.y, Cz, gram_lds); break;
      default: gram_lds_wave<GramLdsUnitDeal<GS, false, 3>>(Y, ldy, k_lo, k_hi, n, ca, cb, u.x, u.y, Cz, gram_lds); break;
    }
  } else {
    switch (threadIdx.x >> 6) {
      case 0: gram_lds_wave<GramLdsUnitDeal<GS, true, 0>>(Y, ldy, k_lo, k_hi, n, ca, ca, u.x, u.y, Cz, gram_lds); break;
      case 1: gram_lds_wave<GramLdsUnitDeal<GS, true, 1>>(Y, ldy, k_lo, k_hi, n, ca, ca, u.x, u.y, Cz, gram_lds); break;
      case 2: gram_lds_wave<GramLdsUnitDeal<GS, true, 2>>(Y, ldy, k_lo, k_hi, n, ca, ca, u.x, u.y, Cz, gram_lds); break;
      default: gram_lds_wave<GramLdsUnitDeal<GS, true, 3>>(Y, ldy, k_lo, k_hi, n, ca, ca, u.x, u.y, Cz, gram_lds); break;
    }
  }
}

template <int CT>
__global__ __launch_bounds__(kTPB) void k_gram_sym_lds(const double* __restrict__ Y, int64_t ldy, int64_t rows, int64_t kchunk, int n,
                                                      double* __restrict__ part, unsigned long long* __restrict__ clk) {
  extern __shared__ double gram_lds[];
  const int64_t k_lo = (int64_t)blockIdx.x * kchunk;
  const int64_t k_hi = k_lo + kchunk < rows ? k_lo + kchunk : rows;
  double* Cz = part + (int64_t)blockIdx.x * n * n;
  const bool rec = clk != nullptr && blockIdx.x == 0 && threadIdx.x < 64;
  unsigned long long c0 = 0, t0 = 0;
  if (rec) {
    c0 = clock64();
    t0 = wall_clock64();
  }
  switch (threadIdx.x >> 6) {
    case 0: gram_lds_wave<GramLdsSymDeal<CT, 0>>(Y, ldy, k_lo, k_hi, n, 0, 0, 0, 0, Cz, gram_lds); break;
    case 1: gram_lds_wave<GramLdsSymDeal<CT, 1>>(Y, ldy, k_lo, k_hi, n, 0, 0, 0, 0, Cz, gram_lds); break;
    case 2: gram_lds_wave<GramLdsSymDeal<CT, 2>>(Y, ldy, k_lo, k_hi, n, 0, 0, 0, 0, Cz, gram_lds); break;
    default: gram_lds_wave<GramLdsSymDeal<CT, 3>>(Y, ldy, k_lo, k_hi, n, 0, 0, 0, 0, Cz, gram_lds); break;
  }
  if (rec && threadIdx.x == 0) {
    clk[0] = clock64() - c0;
    clk[1] = wall_clock64() - t0;
    clk[2] = (unsigned long long)((k_hi - k_lo + 3) >> 2);
    clk[3] = (unsigned long long)((gram_tri_count(CT) + 3) / 4);
  }
}

// out[r][c] = sum_z slice_z[r][c] for the upper 16 x 16 tiles, mirrored into the lower ones (fixed order of z).
// A workgroup owns 16 consecutive entries of G (one 128-byte line of every slice) and deals the slices to 16 lanes per entry, four
// independent accumulators each; the 16 lane sums are added in order.  (Until round 5 one thread walked all slices of its entry with
// two accumulators - 40 workgroups and a chain of 384 dependent loads at n = 100: 0.26 ms of C2's 0.51 ms Gram matrix.)
__global__ __launch_bounds__(kTPB) void k_sum_slices_sym(const double* __restrict__ part, int nz, int n, double* __restrict__ out) {
  __shared__ double sm[16][17];
  const int o = threadIdx.x & 15, zl = threadIdx.x >> 4;
  const int64_t count = (int64_t)n * n;
  const int64_t i = (int64_t)blockIdx.x * 16 + o;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  if (i < count) {
    const int r = (int)(i / n), c = (int)(i - (int64_t)r * n);
    const int64_t src = (r >> 4) <= (c >> 4) ? i : (int64_t)c * n + r;
    const double* p = part + src;
    int z = zl;
    for (; z + 48 < nz; z += 64) {
      a0 += p[(int64_t)z * count];
      a1 += p[(int64_t)(z + 16) * count];
      a2 += p[(int64_t)(z + 32) * count];
      a3 += p[(int64_t)(z + 48) * count];
    }
    for (; z < nz; z += 16) a0 += p[(int64_t)z * count];
  }
  sm[zl][o] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (threadIdx.x < 16 && i < count) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sm[k][threadIdx.x];
    out[i] = t;
  }
}

// G = Y^T Y into out (n x n, symmetric, complete); `part` = scratch of gram_scratch_doubles(n) doubles.
// Returns false when the shape is not covered (n > 3520 or a handful of rows): the caller takes launch_gram + launch_sum_slices.
// K slices: at most kGramMaxSlices, and no more than ~1 GB of n x n partial slices (n = 1000: 125)
int gram_max_slices(int n) {
  const int64_t cap = ((int64_t)1 << 27) / ((int64_t)n * n);
  return (int)std::max<int64_t>(16, std::min<int64_t>(kGramMaxSlices, cap));
}
size_t gram_scratch_doubles(int n) { return (size_t)gram_max_slices(n) * n * n + 4096; }  // (+ the unit table of the grouped form)
// Number of K slices (= workgroups per unit).  Every workgroup fills a CU (four waves, one per SIMD, ~330 registers each), so a
// launch runs in residency rounds of 256 workgroups and a ragged last round idles most of the chip: C5 ran 3 x 536 = 1608 pair
// blocks = 6.3 rounds (7 to wait for) until round 5; 3 x 512 = 6.0 now.  `weight` = the launch's workgroups per slice in units
// of its longest block.
// `slots` = workgroups the chip holds at once: 256 CUs x the kernel's occupancy (registers and LDS; the small forms fit two or more per CU).
static int gram_slices(int n, int64_t rows, double weight, int override, int slots) {
  const int cap = (int)std::min<int64_t>(gram_max_slices(n), std::max<int64_t>(1, (rows + 1023) / 1024));
  if (override > 0) return std::min(override, cap);
  if (cap * weight <= slots) return cap;  // less than one round either way: as many slices as the rows allow
  // equal workgroups (the single-unit kernels, weight 1) with plenty of rows each: exactly one round - perfectly balanced, and a third
  // of the partial slices for the slice sum to read (headline: 768 slices = 245 MB and 0.26 ms of a 8.5 ms Gram matrix; 256: 82 MB)
  if (weight == 1.0 && cap >= slots && rows / slots >= 4096) return slots;
  int best = cap;
  double best_eff = -1.0;
  for (int nz = cap; nz >= std::max(1, (cap * 3) / 5); --nz) {
    const double rounds = nz * weight / slots;
    const double eff = rounds / std::ceil(rounds - 1e-9);
    if (eff > best_eff + 1e-9) {
      best_eff = eff;
      best = nz;
    }
  }
  return best;
}
template <class K>
static bool gram_lds_attr(K kern, size_t bytes) {  // more than 64 KiB of dynamic LDS has to be allowed per kernel
  return bytes <= 65536 || hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
}
bool launch_gram_sym(const double* Y, int64_t ldy, int64_t rows, int n, double* part, double* out, hipStream_t s, unsigned long long* clk,
                     int nz_override, int variant) {
  const int CT = (n + 15) / 16;
  if (CT < 3 || rows < 4096 || ldy != n) return false;
  // variant 0: operands staged through LDS once per workgroup (needs 16-byte aligned rows: even n); 2: the register-ring kernels of round 4
  const bool use_lds = variant != 2 && (n % 2 == 0) && (reinterpret_cast<uintptr_t>(Y) % 16 == 0);
  int nz = 0;
  int64_t kchunk = 0;
  auto slices = [&](double weight, int occupancy) {
    nz = gram_slices(n, rows, weight, nz_override, kNumCU * occupancy);
    kchunk = (rows + nz - 1) / nz;
    kchunk = (kchunk + 3) & ~(int64_t)3;
    nz = (int)((rows + kchunk - 1) / kchunk);
  };
#define LZ_GS(ct)                                                                                                                       \
  case ct:                                                                                                                              \
    if (use_lds && gram_lds_attr(k_gram_sym_lds<ct>, gram_lds_bytes(1, 16 * ct, 8)))                                                     \
      hipLaunchKernelGGL((k_gram_sym_lds<ct>), dim3(nz), dim3(kTPB), gram_lds_bytes(1, 16 * ct, 8), s, Y, ldy, rows, kchunk, n, part, clk); \
    else                                                                                                                                \
      hipLaunchKernelGGL((k_gram_sym<ct>), dim3(nz), dim3(kTPB), 0, s, Y, ldy, rows, kchunk, n, part, clk);                              \
    break;
  if (CT <= 13) {
    // workgroups per CU of k_gram_sym_lds<CT> (registers: 6 / 4 / 4 / 3 / 2 / 2 / 2 / 1 ... for CT = 3 ..; LDS: 160 KiB / its two buffers)
    static const int reg_occ[14] = {0, 0, 0, 6, 4, 4, 3, 2, 2, 2, 1, 1, 1, 1};
    const int occ = use_lds ? std::max(1, std::min<int>(reg_occ[CT], (int)((160 * 1024) / gram_lds_bytes(1, 16 * CT, 8)))) : 1;
    slices(1.0, occ);
    switch (CT) {
      LZ_GS(3) LZ_GS(4) LZ_GS(5) LZ_GS(6) LZ_GS(7) LZ_GS(8) LZ_GS(9) LZ_GS(10) LZ_GS(11) LZ_GS(12) LZ_GS(13)
      default: return false;
    }
  } else {
    // groups of GS column tiles, GS the smallest size that covers the CT tiles with ceil(CT / 11) groups (round 4 had 11 always:
    // n = 400 - 25 tiles - ran as 3 x 11 with 8 tiles of padding, 1.7 x the tile products it needs; 3 x 9 now): the diagonal units
    // and the off-diagonal pairs of the upper triangle of groups
    if ((CT + kGramGS - 1) / kGramGS > 20) return false;  // (n > 3520: the split-K form)
    // group size: the one whose units cost the fewest MFMA issue slots per k-step on their busiest waves (a pair: ceil(GS / 4) GS
    // tiles, a triangle: the snake's heaviest share) - CT = 32 (C5): 4 groups of 8 tile the triangle exactly (6 x 16 + 4 x 9 = 132
    // slots, every wave equally loaded) where 3 groups of 11 need 3 x 33 + 3 x 18 = 153; CT = 25 (n = 400): 4 x 7
    int GS = kGramGS, ng = (CT + kGramGS - 1) / kGramGS;
    {
      double best = 1e300;
      for (int g = 7; g <= kGramGS; ++g) {
        const int k = (CT + g - 1) / g;
        double wt = 0.0;
        for (int w = 0; w < 4; ++w) wt = std::max(wt, (double)gram_diag_count(g, w));
        // (11: its pairs hold 33 accumulator tiles per wave = 264 registers, the LDS-staged form spills there: it runs the register-ring
        //  kernel, which measured ~12 % further from its issue floor)
        const double cost = (0.5 * k * (k - 1) * ((g + 3) / 4) * g + k * wt) * (g == kGramGS ? 1.12 : 1.0);
        if (cost < best - 1e-9) {
          best = cost;
          GS = g;
          ng = k;
        }
      }
    }
    std::vector<int2> ud, up;
    for (int a = 0; a < ng; ++a) {
      ud.push_back(make_int2(GS * a, GS * a));
      for (int b = a + 1; b < ng; ++b) up.push_back(make_int2(GS * a, GS * b));
    }
    // busiest wave of a pair: ceil(GS / 4) row tiles x GS columns; of a triangle: the snake's heaviest share (~ (GS + 1) GS / 8 + ...)
    const double wp = (double)((GS + 3) / 4) * GS;
    double wt = 0.0;
    for (int w = 0; w < 4; ++w) wt = std::max(wt, (double)gram_diag_count(GS, w));
    slices((double)up.size() + (double)ud.size() * wt / wp, (use_lds && GS <= 8) ? 2 : 1);  // (k_gram_units_lds<7 | 8>: two workgroups per CU)
    // the unit table rides behind the partial slices (gram_scratch_doubles reserves 4096 doubles = 4096 int2 for it)
    int2* tab = reinterpret_cast<int2*>(part + (size_t)gram_max_slices(n) * n * n);
    std::vector<int2> all(up);  // pairs first
    all.insert(all.end(), ud.begin(), ud.end());
    if (hipMemcpyAsync(tab, all.data(), all.size() * sizeof(int2), hipMemcpyHostToDevice, s) != hipSuccess) return false;
    if (hipStreamSynchronize(s) != hipSuccess) return false;  // (`all` is a local; once per Gram matrix)
#define LZ_GU(gs)                                                                                                                              \
  case gs: {                                                                                                                                   \
    const size_t lb = std::max(gram_lds_bytes(2, 16 * gs, gs >= 10 ? 2 : 4), gram_lds_bytes(1, 16 * gs, 8));                                    \
    if (use_lds && gs < kGramGS && gram_lds_attr(k_gram_units_lds<(gs < kGramGS ? gs : 7)>, lb))                                                \
      hipLaunchKernelGGL((k_gram_units_lds<(gs < kGramGS ? gs : 7)>), dim3(nz, (unsigned)all.size()), dim3(kTPB), lb, s, Y, ldy, rows, kchunk, n, CT, tab, (int)up.size(), part); \
    else                                                                                                                                       \
      hipLaunchKernelGGL((k_gram_units<gs>), dim3(nz, (unsigned)all.size()), dim3(kTPB), 0, s, Y, ldy, rows, kchunk, n, CT, tab, (int)up.size(), part); \
    break;                                                                                                                                     \
  }
    switch (GS) {
      LZ_GU(7) LZ_GU(8) LZ_GU(9) LZ_GU(10) LZ_GU(11)
      default: return false;
    }
#undef LZ_GU
  }
#undef LZ_GS
  hipLaunchKernelGGL(k_sum_slices_sym, dim3((unsigned)(((int64_t)n * n + 15) / 16)), dim3(kTPB), 0, s, part, nz, n, out);
  return true;
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

__global__ __launch_bounds__(kTPB) void k_extract_column(const double* __restrict__ Y, int64_t ldy, int col, int64_t rows, int64_t rows_pad,
                                                        double* __restrict__ x) {
  const int64_t r = (int64_t)blockIdx.x * kTPB + threadIdx.x;
  if (r < rows_pad) x[r] = r < rows ? Y[r * ldy + col] : 0.0;
}
void launch_extract_column(const double* Y, int64_t ldy, int col, int64_t rows, int64_t rows_pad, double* x, hipStream_t s) {
  hipLaunchKernelGGL(k_extract_column, dim3((unsigned)((rows_pad + kTPB - 1) / kTPB)), dim3(kTPB), 0, s, Y, ldy, col, rows, rows_pad, x);
}

}  // namespace lz
