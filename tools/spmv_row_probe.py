"""Does the SpMV's time depend on WHICH basis row is its input when the basis is 160 GB (BASELINE C4 on one GPU)?  (GPU box.)
    python tools/spmv_row_probe.py [n ...]      default: 24 200  (basis rows allocated; 500 x 500 x 400 7-point stencil)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanczos_amd import _capi  # noqa: E402

dims = (500, 500, 400)
M = int(np.prod(dims))
x = np.random.default_rng(0).uniform(-1, 1, M)
for n in [int(a) for a in sys.argv[1:]] or [24, 200]:
    h = _capi.Handle(0)
    h.set_options(_capi.FLAG_FUSED_NORM | _capi.FLAG_PROFILE)
    h.build_stencil3d_block(dims, 7, 1.0, (-6.0, 1.0, 0.0, 0.0), 0, M, ())
    h.basis_alloc(n)
    rows = sorted({0, n // 8, n // 4, n // 2, 3 * n // 4, n - 1})
    for j in rows:
        h.basis_set_row(j, x)
        h.step_spmv(j)
        h.timings()
        for _ in range(6):
            h.step_spmv(j)
        t = h.timings()["spmv"]
        us = 1e3 * t["ms"] / max(t["timed_launches"], 1)
        print(json.dumps({"basis_rows": n, "basis_GB": round(8e-9 * n * M, 1), "x_is_row": j, "spmv_us": round(us, 1),
                          "frac_of_8TBs_on_100B_per_row": round(100.0 * M / us / 8e6, 4)}), flush=True)
    h.close()
    time.sleep(1)
