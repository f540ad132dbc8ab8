"""SpMV of a 2-D 5-point and of 3-D 7-point stencils, a few launches each, for rocprofv3 --pmc passes (GPU box):
    rocprofv3 --kernel-trace --pmc <counters> ... -- python3 tools/spmv_pmc_probe.py
Prints the device time per launch (hipEvents of the library)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanczos_amd import _capi  # noqa: E402

CASES = [("lap2d_4000x2500", (4000, 2500, 1), 5), ("lap2d_1000x1000", (1000, 1000, 1), 5), ("lap3d_300^3", (300, 300, 300), 7), ("lap3d_500x500x100", (500, 500, 100), 7)]
for name, dims, pts in CASES:
    M = int(np.prod(dims))
    h = _capi.Handle(0)
    h.set_options(_capi.FLAG_FUSED_NORM | _capi.FLAG_PROFILE)
    if os.environ.get("LZ_RB"):  # A/B arms of the fixed-K kernel (knob 5)
        h.set_tuning(_capi.TUNE_FIXED_ROWS, int(os.environ["LZ_RB"]))
    if os.environ.get("LZ_LAYOUT"):  # knob 17: 1 CSR-order kernel, 2 ELL one row per lane (default), 3 ELL two rows per lane
        h.set_tuning(_capi.TUNE_FIXED_LAYOUT, int(os.environ["LZ_LAYOUT"]))
    if pts == 7:
        h.build_stencil3d_block(dims, 7, 1.0, (-6.0, 1.0, 0.0, 0.0), 0, M, ())
    else:
        from lanczos_amd import synthetic
        A = synthetic.laplacian_2d_5pt(dims[0], dims[1])
        h.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
    h.basis_alloc(2)
    xs = np.random.default_rng(0).uniform(-1, 1, M)
    if os.environ.get("LZ_CHECK"):  # against SciPy, bit for bit
        import scipy.sparse
        rp, ci, va = h.get_csr()
        ref = scipy.sparse.csr_matrix((va, ci, rp), shape=(M, M)) * xs
        got = h.spmv_host(xs)
        print(json.dumps({"case": name, "bit_exact_vs_scipy": bool(np.array_equal(got, ref)), "max_abs_diff": float(np.abs(got - ref).max())}), flush=True)
    h.basis_set_row(0, xs)
    h.step_spmv(0)
    h.timings()
    for _ in range(4):
        h.step_spmv(0)
    t = h.timings()["spmv"]
    us = 1e3 * t["ms"] / max(t["timed_launches"], 1)
    print(json.dumps({"case": name, "layout_knob": os.environ.get("LZ_LAYOUT", "default"), "rows": M, "K": pts, "spmv_us": round(us, 1), "GBps_on_12K+16_bytes_per_row": round((12 * pts + 16) * M / us / 1e3, 1)}), flush=True)
    h.close()
