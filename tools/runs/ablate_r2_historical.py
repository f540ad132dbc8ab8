#!/usr/bin/env python3
"""HISTORICAL (rounds 2-3; the ablation arms it drives were deleted from the sources in round 5, results: profiles/r02/ablate_*.json).
Timing-only ablations (kernel-bench build of the library) of the two kernels that sit furthest below their estimate:
phase 2 of the two-phase irregular SpMV (k_pb_rows) and the LDS-staged Ritz GEMM.  Prints one JSON object."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanczos_amd import _capi, synthetic  # noqa: E402

_capi.LIB_PATH = os.path.join(os.path.dirname(_capi.LIB_PATH), "liblanczos_kbench.so")
out = {}
only = sys.argv[1] if len(sys.argv) > 1 else "all"

# --- k_pb_rows arms on the C3 matrix: knob 3 = ablation (1 no product loads, 2 no perm loads, 4 no LDS gathers)
A = synthetic.random_graph_laplacian(10_000_000 if only not in ("ritz", "sreg") else 1000, 35_000_000 if only not in ("ritz", "sreg") else 3000, seed=1234)
M = A.shape[0]
x = np.random.default_rng(0).standard_normal(M)
for arm in ((0, 1, 2, 7) if only not in ("ritz", "sreg") else ()):
    h = _capi.Handle(0)
    h.set_options(_capi.FLAG_PROFILE)
    h.set_tuning(_capi.TUNE_SPMV_ABLATION, arm)
    h.set_tuning(_capi.TUNE_SPMV_PLAN, 2)
    h.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
    h.basis_alloc(2)
    h.basis_set_row(1, x)
    h.step_spmv(1)
    h.timings()
    for _ in range(5):
        h.step_spmv(1)
    t = h.timings()["spmv"]
    out[f"pb_spmv_both_phases_arm{arm}_us"] = round(1e3 * t["ms"] / t["launches"], 1)
    h.close()
del A
if only == "pb":
    print(json.dumps(out, indent=1))
    sys.exit(0)

# --- Ritz GEMM arms at the headline shape: knob 9 = 10 + ablation (1 no stores, 2 no V loads, 4 no MFMA)
A = synthetic.laplacian_2d_5pt(4000, 2500)
M = A.shape[0]
v0 = synthetic.reference_start_vector(M)
v0 /= np.linalg.norm(v0)
h = _capi.Handle(0)
h.set_options(_capi.FLAG_PROFILE | _capi.FLAG_FUSED_NORM | _capi.FLAG_REORTH_PARTIAL)  # cheap run: the basis content is irrelevant here
h.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
a, b = h.run(200, v0)
S = np.linalg.eigh(np.diag(a) + np.diag(b, 1) + np.diag(b, -1))[1]
for arm in ((0, 5, 20, 21, 22, 23, 5) if only == "sreg" else (0, 2, 3, 4, 5, 0, 5)):
    h.set_tuning(_capi.TUNE_RITZ_KERNEL, arm)
    h.ritz_vectors(S, fetch=False)
    h.timings()
    for _ in range(3):
        h.ritz_vectors(S, fetch=False)
    t = h.timings()["ritz"]
    out[f"ritz_gemm_variant{arm}_ms"] = round(t["ms"] / t["launches"], 3)
h.close()
print(json.dumps(out, indent=1))
