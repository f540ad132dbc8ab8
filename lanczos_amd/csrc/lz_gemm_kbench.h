// Kernel-bench build ONLY (make KBENCH=1 -> liblanczos_kbench.so; included by lz_gemm.hip under LZ_KBENCH, inside namespace lz,
// never by the product build): the RETIRED Ritz back-transform kernels - built, measured slower than what replaced them, kept
// bit-identity-tested (tests/test_gpu_lanczos.py::test_retired_ritz_gemm_arms_in_the_kernel_bench_build):
//   k_gemm_tn_persist (knob 9 = 2)  persistent waves, operands through L1
//   k_gemm_tn_lds     (knob 9 = 3, 4)  S staged through LDS in 16-row panels, two / one wave per SIMD
//   k_gemm_tn_sl      (knob 9 = 6)  S resident in LDS, 16-ROW tiles (superseded by k_gemm_tn_sl2's 32-row tiles: 0.49 vs 0.43 ms at n = 100)
// Measurements: DESIGN.md (lab notebook, rounds 2-3), profiles/r02/ablate_pb_rows_and_ritz.json, profiles/r03/.
// The timing-only ablation arms these kernels (and the live ones) used to carry as template parameters - wrong results on purpose,
// for one-off measurements whose results are in profiles/r01 .. r04 - were deleted in round 5 (last present in commit 66d3330).
#pragma once

// Persistent variant for the Ritz back-transform (one K chunk, all columns in one group): the same 32 x NT*16 wave tile
// and MFMA schedule, but a wave does not end with its tile - a grid of one workgroup per CU (one wave per SIMD: the tile
// needs ~330 registers) walks the row tiles with a grid stride.  Why: with one short-lived workgroup per tile the CU
// sits idle from the moment the first of its four waves finishes until the next workgroup has been dispatched and its
// first operands have arrived - 50 k-steps (35 us) of work per 10-15 us of turnover, MfmaUtil 65 % (profiles/r01).  Here the
// operand pipeline simply runs on into the next tile: the A ring (PA k-steps ahead) is fed by a cursor that crosses tile
// boundaries, the B row of step 0 is prefetched during the last step, and the only per-tile overhead left is storing
// the 32 x n results.  Each tile consumes a whole number of ring turns (the padding steps load nothing).
template <int NT>
__global__ __launch_bounds__(kTPB) void k_gemm_tn_persist(const double* __restrict__ A, int64_t lda, int64_t mdim, int kcount,
                                                         const double* __restrict__ B, int64_t ldb, int ncols,
                                                         double* __restrict__ C, int64_t ldc) {
  // Branch-free inner loop (one basic block per ring turn, so the scheduler can count outstanding loads exactly):
  //  * every step of a tile runs, the padding steps included: their A rows are clamped to the last real row (finite
  //    values) and their B rows are the zero rows of the padded S (4 * nsteps_pad == ldb rows exist), product 0;
  //  * the A cursor runs past the last tile with clamped (valid, unused) addresses.
  constexpr int PA = 4;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  const int64_t ntiles = (mdim + 31) / 32;
  const int64_t wave = (int64_t)blockIdx.x * (kTPB / 64) + w, nwaves = (int64_t)gridDim.x * (kTPB / 64);
  const int nsteps_pad = ((kcount + 3) / 4 + PA - 1) / PA * PA;
  const int CT = (ncols + 15) / 16;
  int colb[NT];
#pragma unroll
  for (int b = 0; b < NT; ++b) colb[b] = 16 * (b < CT ? b : CT - 1) + lr;
  // A cursor: (tile, step) of the next load to issue
  int64_t pt = wave;
  int ps = 0;
  auto issue_a = [&](double& x0, double& x1) {
    const int64_t t = pt < ntiles ? pt : ntiles - 1;
    int64_t ma = t * 32 + lr, mb = ma + 16;
    ma = ma < mdim ? ma : mdim - 1;  // rows past the end: valid address, result never stored
    mb = mb < mdim ? mb : mdim - 1;
    int kr = 4 * ps + lk;
    kr = kr < kcount ? kr : kcount - 1;
    const double* ar = A + (int64_t)kr * lda;
    x0 = __builtin_nontemporal_load(ar + ma);
    x1 = __builtin_nontemporal_load(ar + mb);
    ++ps;
    if (ps == nsteps_pad) {
      ps = 0;
      pt += nwaves;
    }
  };
  double ra[PA][2];
#pragma unroll
  for (int p = 0; p < PA; ++p) issue_a(ra[p][0], ra[p][1]);
  double bcur[NT];
  {
    const double* sr = B + (int64_t)lk * ldb;
#pragma unroll
    for (int b = 0; b < NT; ++b) bcur[b] = sr[colb[b]];
  }
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    double4_t acc[2][NT];
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      acc[0][b] = (double4_t){0.0, 0.0, 0.0, 0.0};
      acc[1][b] = (double4_t){0.0, 0.0, 0.0, 0.0};
    }
    for (int s0 = 0; s0 < nsteps_pad; s0 += PA) {
#pragma unroll
      for (int p = 0; p < PA; ++p) {
        const int st = s0 + p;
        const double a0 = ra[p][0], a1 = ra[p][1];
        issue_a(ra[p][0], ra[p][1]);  // refill this ring slot (PA steps ahead, possibly in the next tile)
        double bnxt[NT];
        {
          const int nx = st + 1 < nsteps_pad ? st + 1 : 0;  // last step: row 0 for the next tile
          const double* sr = B + (int64_t)(4 * nx + lk) * ldb;
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
    const int64_t m0 = tile * 32;
    if (m0 + 32 <= mdim) {  // wave-uniform: every tile but possibly the last stores without row checks
      double* cbase = C + (m0 + lk) * ldc + lr;
#pragma unroll
      for (int b = 0; b < NT; ++b) {
        const bool colok = b + 1 < CT || 16 * b + lr < ncols;  // only the last column tile can be ragged
        if (b < CT && colok) {
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) cbase[(int64_t)(16 * a + 4 * g) * ldc + 16 * b] = acc[a][b][g];
        }
      }
    } else {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) {
          if (b >= CT) continue;
          const int col = 16 * b + lr;
          if (col >= ncols) continue;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int64_t m = m0 + 16 * a + lk + 4 * g;
            if (m < mdim) C[m * ldc + col] = acc[a][b][g];
          }
        }
    }
  }
}

// Ritz back-transform, A/B arm (variant 3): persistent waves, TWO per SIMD, S staged through LDS.
// What the two earlier kernels taught (profiles/r02/ritz_gemm_ab.json): k_gemm_tn and its persistent-wave variant both sit at
// MfmaUtil 65 % - so wave turnover was not the gap; what they share is ONE wave per SIMD (a 32-row x n tile needs ~330
// registers), and a lone wave cannot keep the f64 matrix pipe issuing back to back.  Halving the tile to 16 rows (104
// accumulator registers) lets two waves share a SIMD, but doubles the S traffic per MFMA - one 512-byte L1/L2 read per
// MFMA is the vector-memory pipe's whole budget - so S is staged through LDS instead: the 512 threads of a block copy S
// in panels of 16 k-rows (4 MFMA k-steps) into a double buffer, one barrier per panel, and every wave reads its B
// operands with ds_read_b64 (a quarter of the LDS bandwidth).  The only per-wave global traffic left in the loop is the
// V stream itself: one 8-byte non-temporal load per lane per k-step, four steps ahead.
// Every wave of a block runs the same number of tiles (surplus tiles are clamped and not stored): the barriers match.
template <int NT, int NA = 1>
__global__ __launch_bounds__(NA == 2 ? 256 : 512) void k_gemm_tn_lds(const double* __restrict__ A, int64_t lda, int64_t mdim, int kcount,
                                                                   const double* __restrict__ B, int ldb, int ncols,
                                                                   double* __restrict__ C, int64_t ldc) {
  // NA = 1: two waves per SIMD (512 threads), a wave owns 16 rows; NA = 2: one wave per SIMD (256 threads), 32 rows -
  // each staged S fragment then feeds two MFMAs (the operand probe's best case: 72.9 TF, tools/probes/mfma_f64_operands)
  constexpr int KP = 4;       // k-steps per panel = turns of the A ring
  constexpr int NTHR = NA == 2 ? 256 : 512;
  constexpr int TR = 16 * NA;  // rows per wave tile
  extern __shared__ double sB[];  // 2 panels of 16 rows x ldb doubles
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  const int64_t ntiles = (mdim + TR - 1) / TR;
  const int64_t nwaves = (int64_t)gridDim.x * (NTHR / 64);
  const int64_t wave = (int64_t)blockIdx.x * (NTHR / 64) + w;
  const int64_t rounds = (ntiles + nwaves - 1) / nwaves;
  const int npanels = ((kcount + 3) / 4 + KP - 1) / KP;  // 16 * npanels == ldb rows of the zero-padded S
  const int panel_d2 = 16 * ldb / 2;                     // double2 elements per panel
  constexpr int PF = NA == 2 ? 8 : 4;                    // double2 loads per thread per panel (ldb <= 256: 16 * 256 / 2 / NTHR)
  const int CT = (ncols + 15) / 16;
  int colb[NT];
#pragma unroll
  for (int b = 0; b < NT; ++b) colb[b] = 16 * (b < CT ? b : CT - 1) + lr;
  const double2* B2 = reinterpret_cast<const double2*>(B);
  double2* s2 = reinterpret_cast<double2*>(sB);
  // panel 0 -> buffer 0
  for (int i = threadIdx.x; i < panel_d2; i += NTHR) s2[i] = B2[i];
  // A cursor: (round, step) of the next load
  int64_t pr = 0;
  int ps = 0;
  auto issue_a = [&](double (&x)[NA]) {
    int64_t t = wave + pr * nwaves;
    t = t < ntiles ? t : ntiles - 1;
    int kr = 4 * ps + lk;
    kr = kr < kcount ? kr : kcount - 1;  // padding steps: finite values, multiplied by the zero rows of S
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      int64_t m = t * TR + 16 * a + lr;
      m = m < mdim ? m : mdim - 1;
      x[a] = __builtin_nontemporal_load(A + (int64_t)kr * lda + m);
    }
    if (++ps == KP * npanels) {
      ps = 0;
      ++pr;
    }
  };
  double ra[KP][NA];
#pragma unroll
  for (int p = 0; p < KP; ++p) issue_a(ra[p]);
  int buf = 0;
  for (int64_t rd = 0; rd < rounds; ++rd) {
    double4_t acc[NA][NT];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};
    for (int pn = 0; pn < npanels; ++pn) {
      __syncthreads();  // panel `pn` is complete in buffer `buf`; nobody reads buffer buf^1 any more
      // prefetch the next panel of S (wrapping to panel 0 for the next tile) into registers
      const int nxt = pn + 1 < npanels ? pn + 1 : 0;
      double2 pf[PF];
#pragma unroll
      for (int q = 0; q < PF; ++q) {
        const int i = threadIdx.x + q * NTHR;
        pf[q] = i < panel_d2 ? B2[(int64_t)nxt * panel_d2 + i] : make_double2(0.0, 0.0);
      }
      const double* sb = sB + (size_t)buf * 16 * ldb;
#pragma unroll
      for (int p = 0; p < KP; ++p) {
        double a0[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) a0[a] = ra[p][a];
        issue_a(ra[p]);
        const double* srow = sb + (4 * p + lk) * ldb;
        double bv[NT];
#pragma unroll
        for (int b = 0; b < NT; ++b) bv[b] = srow[colb[b]];
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
          for (int a = 0; a < NA; ++a) {
            acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[a], bv[b], acc[a][b], 0, 0, 0);
          }
      }
      double2* dst = s2 + (size_t)(buf ^ 1) * panel_d2;
#pragma unroll
      for (int q = 0; q < PF; ++q) {
        const int i = threadIdx.x + q * NTHR;
        if (i < panel_d2) dst[i] = pf[q];
      }
      buf ^= 1;
    }
    const int64_t tile = wave + rd * nwaves;
    if (tile < ntiles) {  // wave-uniform
      const int64_t m0 = tile * TR;
      if (m0 + TR <= mdim) {
        double* cbase = C + (m0 + lk) * ldc + lr;
#pragma unroll
        for (int b = 0; b < NT; ++b) {
          const bool colok = b + 1 < CT || 16 * b + lr < ncols;
          if (b < CT && colok) {
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
              for (int g = 0; g < 4; ++g) cbase[(int64_t)(16 * a + 4 * g) * ldc + 16 * b] = acc[a][b][g];
          }
        }
      } else {
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
          for (int b = 0; b < NT; ++b) {
            if (b >= CT) continue;
            const int col = 16 * b + lr;
            if (col >= ncols) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int64_t m = m0 + 16 * a + lk + 4 * g;
              if (m < mdim) C[m * ldc + col] = acc[a][b][g];
            }
          }
      }
    }
  }
}


// Ritz back-transform for n <= 128: S RESIDENT IN LDS, Y-stationary waves, no barrier after the prologue.
// Why a third kernel.  The S-stationary kernel pays ~1400 cycles per 16-row tile for its barrier, partial-tile hand-over and
// ring refill whatever n is (measured in round 3: 11 874 cycles per tile against an MFMA floor of 10 400 at n = 200, but 4 286
// against 2 912 at n = 100 - 0.68 in cycles, 0.49 of peak at BASELINE C2).  For n <= 128 the whole of S (<= 128 KB) fits the
// CU's LDS, so nothing has to be staged, handed over or synchronised: every wave owns whole 16-row tiles of Y (NT
// accumulators), streams its own V fragments from HBM through a register ring that runs across tile boundaries, and reads
// its B operands out of LDS - stored in MFMA-fragment order (the fragment of k-step t and column tile b is 64 consecutive
// doubles: conflict-free ds_read_b64), one LDS read per MFMA, fetched one k-step ahead.  WPS waves per SIMD hide each
// other's memory instructions (with one wave per SIMD every vector-memory instruction costs ~165 cycles of MFMA issue,
// DESIGN.md section 4).  S is zero-padded to 4 KS rows, so the padding k-steps and the ragged last column tile add zeros.
template <int NT, int KS, int WPS>
__global__ __launch_bounds__(256 * WPS) void k_gemm_tn_sl(const double* __restrict__ A, int64_t lda, int64_t mdim, int kcount,
                                                         const double* __restrict__ B, int64_t ldb, int ncols,
                                                         double* __restrict__ C, int64_t ldc, unsigned long long* __restrict__ clk) {
  constexpr int NW = 4 * WPS;  // waves per workgroup
  constexpr int PA = NT <= 7 ? 16 : 8;  // V fragments in flight per wave (k-steps ahead; 8 where the accumulators leave no room)
  extern __shared__ double sS[];  // [KS][NT][64]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  if (clk != nullptr && lane == 0) {
    const unsigned long long now = wall_clock64();
    atomicMin(clk + 4, now);                        // first wave in (kernel-internal span)
    if (blockIdx.x == 0) atomicMin(clk + 6, now);   // workgroup 0 in
  }
  {
    // S -> LDS, fragment f = (t, b): S[4 t + lk][16 b + lr] (rows / columns inside the zero-padded S).  ALL of a wave's loads
    // are issued before the first LDS store: as a load -> store loop the ~22 dependent round trips per wave took ~70 us of a
    // 450 us kernel (measured with the in-kernel clock, round 3).
    constexpr int NF = (KS * NT + NW - 1) / NW;
    double tmp[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const int f = w + i * NW;
      const int t = f / NT, b = f - t * NT;
      tmp[i] = f < KS * NT ? B[(int64_t)(4 * t + lk) * ldb + 16 * b + lr] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const int f = w + i * NW;
      if (f < KS * NT) sS[f * 64 + lane] = tmp[i];
    }
  }
  const bool rec = clk != nullptr && blockIdx.x == 0 && __builtin_amdgcn_readfirstlane(w) == 0;
  uint64_t c0 = 0, t0 = 0;
  if (rec) {
    c0 = clock64();
    t0 = wall_clock64();
  }
  __syncthreads();
  const int64_t ntiles = (mdim + 15) / 16;
  const int64_t nwaves = (int64_t)gridDim.x * NW;
  const int64_t wave = (int64_t)blockIdx.x * NW + w;
  // V fragments: k-step t of a tile is the 4 basis rows 4 t + lk at the 16 matrix rows m0 + lr.  The ring slot of step t is
  // t % PA; the fragment of step t + PA is requested when step t's has been consumed - from the NEXT tile once t + PA >= KS -
  // so all indices are compile-time and the per-step address work is one 64-bit add (the run-time cursor of the first
  // version cost ~15 integer VALU instructions per k-step: as many issue cycles as one of its 7 MFMAs).
  const int64_t stride4 = 4 * lda;  // doubles between consecutive k-steps of one lane
  auto tile_base = [&](int64_t tile) {  // lane's pointer to (basis row lk, matrix row m0 + lr), both clamped into the array
    const int64_t t = tile < ntiles ? tile : ntiles - 1;  // past the last tile: a valid address, never used
    int64_t m = t * 16 + lr;
    m = m < mdim ? m : mdim - 1;                           // rows past the end: finite values, results not stored
    return A + (int64_t)lk * lda + m;
  };
  auto frag = [&](const double* base, int t) {  // t is a compile-time constant at every call site
    if (t >= KS - 2) {  // only the last two k-steps can reach past basis row kcount - 1 (S has zero rows there: any finite value will do)
      int kr = 4 * t + lk;
      kr = kr < kcount ? kr : kcount - 1;
      return __builtin_nontemporal_load(base + (int64_t)(kr - lk) * lda);
    }
    return __builtin_nontemporal_load(base + (int64_t)t * stride4);
  };
  const double* cur = tile_base(wave);
  double ra[PA];
#pragma unroll
  for (int p = 0; p < PA; ++p) ra[p] = frag(cur, p < KS ? p : KS - 1);
  int64_t mytiles = 0;
  for (int64_t tile = wave; tile < ntiles; tile += nwaves, ++mytiles) {
    const double* nxt = tile_base(tile + nwaves);
    double4_t acc[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[b] = (double4_t){0.0, 0.0, 0.0, 0.0};
    double bcur[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b) bcur[b] = sS[b * 64 + lane];
#pragma unroll
    for (int t = 0; t < KS; ++t) {
      const double a = ra[t % PA];
      if (PA < KS) ra[t % PA] = t + PA < KS ? frag(cur, t + PA) : frag(nxt, t + PA - KS);
      double bnxt[NT];
      if (t + 1 < KS) {
#pragma unroll
        for (int b = 0; b < NT; ++b) bnxt[b] = sS[((t + 1) * NT + b) * 64 + lane];
      }
      __builtin_amdgcn_sched_barrier(0);  // one k-step of operands in flight, not all KS of them (the loop is fully unrolled)
#pragma unroll
      for (int b = 0; b < NT; ++b) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bcur[b], acc[b], 0, 0, 0);
      if (t + 1 < KS) {
#pragma unroll
        for (int b = 0; b < NT; ++b) bcur[b] = bnxt[b];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (PA >= KS) {  // tiny n: the whole next tile's fragments are requested at once
#pragma unroll
      for (int p = 0; p < PA; ++p) ra[p] = frag(nxt, p < KS ? p : KS - 1);
    } else if constexpr (KS % PA != 0) {
      // the fragments already requested for the next tile (its steps 0 .. PA-1) sit KS % PA slots further on: rotate the
      // ring by that compile-time amount so that slot t % PA holds step t again - a few register moves per tile
      double tmp[PA];
#pragma unroll
      for (int i = 0; i < PA; ++i) tmp[i] = ra[(i + KS) % PA];
#pragma unroll
      for (int i = 0; i < PA; ++i) ra[i] = tmp[i];
    }
    cur = nxt;
    // results: D[row = lk + 4 g][col = lr] of column tile b
    const int64_t m0 = tile * 16;
    double* cb = C + (m0 + lk) * ldc + lr;
    if (m0 + 16 <= mdim) {  // wave-uniform: only the very last tile can be ragged in rows
#pragma unroll
      for (int b = 0; b < NT; ++b) {
        if (16 * b + lr < ncols) {
#pragma unroll
          for (int g = 0; g < 4; ++g) __builtin_nontemporal_store(acc[b][g], cb + (int64_t)(4 * g) * ldc + 16 * b);
        }
      }
    } else {
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          if (16 * b + lr < ncols && m0 + lk + 4 * g < mdim) cb[(int64_t)(4 * g) * ldc + 16 * b] = acc[b][g];
    }
  }
  if (rec) {
    clk[0] = clock64() - c0;
    clk[1] = wall_clock64() - t0;
    clk[2] = (unsigned long long)mytiles;
    clk[3] = (unsigned long long)(NT * KS * NW);  // MFMAs per "tile slot" over the four SIMDs: NW waves each do NT KS per tile they own
  }
  if (clk != nullptr && lane == 0) {
    const unsigned long long now = wall_clock64();
    atomicMax(clk + 5, now);  // last wave out
    if (blockIdx.x == 0) atomicMax(clk + 7, now);  // workgroup 0 out
    if (blockIdx.x < 256) atomicMax(clk + 8 + blockIdx.x, now);  // per-workgroup exit tick (diagnostic: LZ_DEBUG_TIMING)
  }
}


// S-in-LDS launcher (n <= 128)
template <int NT, int KS, int WPS>
static hipError_t launch_sl(const double* V, int64_t ldv, int64_t rows, int n, const double* Spad, int npad, double* Y, int64_t ldy,
                            hipStream_t s, unsigned long long* clk) {
  constexpr size_t lds = (size_t)KS * NT * 64 * sizeof(double);
  static hipError_t attr = hipErrorNotReady;
  if (attr == hipErrorNotReady)
    attr = lds > 65536 ? hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_tn_sl<NT, KS, WPS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                       : hipSuccess;
  if (attr != hipSuccess) return attr;
  const int64_t ntiles = (rows + 15) / 16;
  const int grid = (int)std::min<int64_t>(kNumCU, (ntiles + 4 * WPS - 1) / (4 * WPS));
  hipLaunchKernelGGL((k_gemm_tn_sl<NT, KS, WPS>), dim3(grid), dim3(256 * WPS), lds, s, V, ldv, rows, n, Spad, (int64_t)npad, n, Y, ldy, clk);
  return hipSuccess;
}

template <int WPS>
static bool sl_dispatch(const double* V, int64_t ldv, int64_t rows, int n, const double* Spad, int npad, double* Y, int64_t ldy,
                        hipStream_t s, unsigned long long* clk, hipError_t* err) {
  const int NT = (n + 15) / 16;
  const int KS = (n + 3) / 4;  // exact: 4 NT - 3 .. 4 NT (no even rounding here: every wave owns all of K)
#define LZ_SL(nt, ks)                                                                    \
  if (NT == nt && KS == ks) {                                                            \
    *err = launch_sl<nt, ks, WPS>(V, ldv, rows, n, Spad, npad, Y, ldy, s, clk);           \
    return true;                                                                         \
  }
#define LZ_SL4(nt) LZ_SL(nt, 4 * nt - 3) LZ_SL(nt, 4 * nt - 2) LZ_SL(nt, 4 * nt - 1) LZ_SL(nt, 4 * nt)
  LZ_SL4(1) LZ_SL4(2) LZ_SL4(3) LZ_SL4(4) LZ_SL4(5) LZ_SL4(6) LZ_SL4(7) LZ_SL4(8)
#undef LZ_SL4
#undef LZ_SL
  return false;
}


// The retired Ritz GEMM kernels by knob-9 value: 2 persistent waves, 3 / 4 S staged through LDS (two / one wave per SIMD).
static bool kbench_ritz_arm(const double* V, int64_t ldv, int64_t rows, int n, const double* Spad, int npad, double* Y, int64_t ldy,
                            hipStream_t s, int variant, hipError_t* err) {
  *err = hipSuccess;
  const int CT = (n + 15) / 16;
  const int64_t ntiles = (rows + 31) / 32;
  if (variant < 2 || variant == 5 || CT > 16 || ntiles < 2 * kNumCU * (kTPB / 64)) return false;
  if (variant == 4) {  // one wave per SIMD with a 32-row tile, S through LDS: every staged fragment feeds two MFMAs
    const size_t lds = (size_t)2 * 16 * npad * sizeof(double);
#define LZ_TNL2(nt)                                                                                                           \
  case nt:                                                                                                                    \
    hipLaunchKernelGGL((k_gemm_tn_lds<nt, 2>), dim3(kNumCU), dim3(256), lds, s, V, ldv, rows, n, Spad, npad, n, Y, ldy);    \
    break;
    switch (CT) {
      LZ_TNL2(1) LZ_TNL2(2) LZ_TNL2(3) LZ_TNL2(4) LZ_TNL2(5) LZ_TNL2(6) LZ_TNL2(7) LZ_TNL2(8)
      LZ_TNL2(9) LZ_TNL2(10) LZ_TNL2(11) LZ_TNL2(12) LZ_TNL2(13) LZ_TNL2(14) LZ_TNL2(15) LZ_TNL2(16)
      default: break;
    }
#undef LZ_TNL2
    return true;
  }
  if (variant == 3) {  // two waves per SIMD, S through LDS
    const size_t lds = (size_t)2 * 16 * npad * sizeof(double);
#define LZ_TNL(nt)                                                                                                            \
  case nt:                                                                                                                    \
    hipLaunchKernelGGL((k_gemm_tn_lds<nt>), dim3(kNumCU), dim3(512), lds, s, V, ldv, rows, n, Spad, npad, n, Y, ldy);         \
    break;
    switch (CT) {
      LZ_TNL(1) LZ_TNL(2) LZ_TNL(3) LZ_TNL(4) LZ_TNL(5) LZ_TNL(6) LZ_TNL(7) LZ_TNL(8)
      LZ_TNL(9) LZ_TNL(10) LZ_TNL(11) LZ_TNL(12) LZ_TNL(13) LZ_TNL(14) LZ_TNL(15) LZ_TNL(16)
      default: break;
    }
#undef LZ_TNL
    return true;
  }
  if (variant == 2) {
    const dim3 grid(kNumCU), block(kTPB);
#define LZ_TNP(nt)                                                                                                  \
  case nt:                                                                                                          \
    hipLaunchKernelGGL((k_gemm_tn_persist<nt>), grid, block, 0, s, V, ldv, rows, n, Spad, (int64_t)npad, n, Y, ldy); \
    break;
    switch (CT) {
      LZ_TNP(1) LZ_TNP(2) LZ_TNP(3) LZ_TNP(4) LZ_TNP(5) LZ_TNP(6) LZ_TNP(7) LZ_TNP(8)
      LZ_TNP(9) LZ_TNP(10) LZ_TNP(11) LZ_TNP(12) LZ_TNP(13) LZ_TNP(14) LZ_TNP(15) LZ_TNP(16)
      default: break;
    }
#undef LZ_TNP
    return true;
  }
  return false;
}
