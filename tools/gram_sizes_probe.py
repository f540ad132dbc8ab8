#!/usr/bin/env python3
"""Gram matrix G = Y^T Y (lz_ritz_gram, the two checks of get_H_eigs) at the sizes VERDICT r4 item 3 names - C2 (M = 1e6, n = 100),
the headline (1e7, 200), the reference's largest run (160^3, n = 400), C5 (1e7, 500) - event-timed, second call, against the split-K
TN GEMM (knob 19 = 1) for the values; optional K-slice overrides (knob 21) as extra arms.
usage: gram_sizes_probe.py [case,case,...] [slices,slices,...]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanczos_amd import _capi, synthetic  # noqa: E402

CASES = {"c2": ((1000, 1000), 100), "headline": ((4000, 2500), 200), "n400": ((2560, 1600), 400), "c5": ((4000, 2500), 500),
         "n224": ((2000, 1000), 224), "n300": ((2000, 1000), 300)}
which = (sys.argv[1] if len(sys.argv) > 1 else "c2,headline,n400,c5").split(",")
overrides = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else []
for name in which:
    dims, n = CASES[name]
    A = synthetic.laplacian_2d_5pt(*dims)
    M = A.shape[0]
    v0 = synthetic.reference_start_vector(M)
    v0 /= np.linalg.norm(v0)
    h = _capi.Handle(0)
    h.set_options(_capi.FLAG_REORTH_PARTIAL | _capi.FLAG_PROFILE)  # (a quick basis: the Gram kernel's time does not depend on the values)
    h.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
    a, b = h.run(n, v0)
    S = np.linalg.eigh(np.diag(a) + np.diag(b, 1) + np.diag(b, -1))[1]
    h.ritz_vectors(S, fetch=False)
    h.set_tuning(_capi.TUNE_GRAM_KERNEL, 1)
    ref = h.ritz_gram()
    h.timings()
    ref = h.ritz_gram()
    t_ref = h.timings()["ritz"]["ms"]
    h.set_tuning(_capi.TUNE_GRAM_KERNEL, 0)
    flops = float(M) * n * (n + 1)
    for ov in [0] + overrides:
        h.set_tuning(_capi.TUNE_GRAM_SLICES, ov)
        h.ritz_gram()
        h.timings()
        best = 1e9
        for _ in range(3):
            G = h.ritz_gram()
            best = min(best, h.timings()["ritz"]["ms"])
        print(json.dumps({"case": name, "M": M, "n": n, "slices_override": ov, "ms": round(best, 3), "tflops_symmetric_half": round(flops / best / 1e9, 2),
                          "frac_of_78.6": round(flops / best / 1e9 / 78.6, 4), "split_k_gemm_ms": round(t_ref, 3),
                          "max_diff_vs_split_k": float(np.abs(G - ref).max()), "symmetric": bool(np.array_equal(G, G.T)), **h.gram_info()}), flush=True)
    h.close()
