#!/usr/bin/env python3
"""Max |alpha - alpha_ref|, |beta - beta_ref| of the headline run (M = 1e7, k = 200) on the GPU against the full-size run of
the reference itself (tests/golden/headline_lap2d_4000x2500_n200.npz, oracle/gen_golden_headline.py); both reorth modes."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanczos_amd import Lanczos, synthetic  # noqa: E402

g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "headline_lap2d_4000x2500_n200.npz"))
H = synthetic.laplacian_2d_5pt(4000, 2500).to_scipy()
out = {"reference_run_to_run": [float(g["alpha_moved"].max()), float(g["beta_moved"].max())]}
Lanczos.verbose = False
for fused in (True, False):
    Lanczos.fused_norm = fused
    s = Lanczos(H)
    s.execute_Lanczos(200)
    a, b = np.diag(s.H_eff), np.diag(s.H_eff, 1)
    out["fused_norm" if fused else "reference_order"] = {"max_abs_dalpha": float(np.abs(a - g["alpha"]).max()), "max_abs_dbeta": float(np.abs(b - g["beta"]).max()),
                                                          "max_rel_ritz": float(np.abs(np.linalg.eigvalsh(s.H_eff) - np.linalg.eigvalsh(np.diag(g["alpha"]) + np.diag(g["beta"], 1) + np.diag(g["beta"], -1))).max() / 8)}
print(json.dumps(out, indent=1))
