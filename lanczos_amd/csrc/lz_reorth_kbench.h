// Kernel-bench build ONLY (make KBENCH=1 -> liblanczos_kbench.so; included by lz_reorth.hip under LZ_KBENCH, never by the product
// build): the retired 16x16x4 MFMA form of re-orthogonalisation pass 1 (LZ_FLAG_QTW_MFMA).  Built in round 1, measured 20 % slower
// than the 4x4x4 kernel (its A fragment forces 16 rows x 64 bytes per wave-load - half lines; DESIGN.md section 4,
// profiles/r01/ab_qtw_mfma4.json), kept bit-identity-tested (tests/test_gpu_lanczos.py::test_kernel_variants_agree).
// (Included INSIDE namespace lz, behind qtw_stage_w.)
#pragma once

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


template <int SCALE>
static hipError_t kb_launch_qtw_mfma(double* V, int64_t ldv, int64_t len, int nrows, int j, const double* r, const double* nrm2, double* beta_slot,
                                     const QtwPlan& plan, double* part, hipStream_t s) {
  const size_t lds = (size_t)plan.L * sizeof(double);
  const dim3 grid(plan.G), block(kTPB);
#define LZ_QTW_KB_ARGS V, ldv, len, nrows, j, r, nrm2, beta_slot, plan.L, plan.P, part
  switch (plan.variant) {
    case 1: hipLaunchKernelGGL((k_qtw_mfma<SCALE, 4, 1>), grid, block, lds, s, LZ_QTW_KB_ARGS); break;
    case 2: hipLaunchKernelGGL((k_qtw_mfma<SCALE, 16, 1>), grid, block, lds, s, LZ_QTW_KB_ARGS); break;
    case 4: hipLaunchKernelGGL((k_qtw_mfma<SCALE, 8, 2>), grid, block, lds, s, LZ_QTW_KB_ARGS); break;
    default: hipLaunchKernelGGL((k_qtw_mfma<SCALE, 8, 1>), grid, block, lds, s, LZ_QTW_KB_ARGS); break;
  }
#undef LZ_QTW_KB_ARGS
  return hipSuccess;
}
