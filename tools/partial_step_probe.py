"""Where a step of the selective (partial re-orthogonalisation) loop spends its time, one MI355X: the headline matrix (and a 7-point
grid) on the device-decided loop, per kernel class from the library's hipEvents (every launch bracketed: stride 1), for the SpMV
arms - knob 17 = 0 row-class coded (knob 23: units per workgroup), 2 uncoded ELL.   python tools/partial_step_probe.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanczos_amd import _capi, synthetic  # noqa: E402

CASES = [("lap2d_5pt_4000x2500", lambda: synthetic.laplacian_2d_5pt(4000, 2500), 200), ("lap3d_7pt_300^3", None, 100)]
for name, build, n in CASES:
    A = build() if build else None
    M = A.shape[0] if A is not None else 300 ** 3
    v0 = np.random.RandomState(99).uniform(-1, 1, M)
    v0 /= np.linalg.norm(v0)
    base = None
    for layout, group in ((2, 0), (0, 1), (0, 0)):  # knob 23: 1 = one row per lane (A/B), 0 = two adjacent rows per lane
        h = _capi.Handle(0)
        h.set_options(_capi.FLAG_REORTH_PARTIAL | _capi.FLAG_PROFILE)
        h.set_tuning(_capi.TUNE_FIXED_LAYOUT, layout)
        h.set_tuning(_capi.TUNE_CLS_GROUP, group)
        h.set_tuning(_capi.TUNE_PROFILE_STRIDE, 1)
        if A is not None:
            h.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
        else:
            h.build_stencil3d_block((300, 300, 300), 7, 1.0, (-6.0, 1.0, 0.0, 0.0), 0, M, ())
        h.run(n, v0)
        h.timings()
        t0 = time.perf_counter()
        a, b = h.run(n, v0)
        wall = time.perf_counter() - t0
        t = h.timings()
        if base is None:
            base = (a, b)
        out = {"case": name, "n": n, "knob17": layout, "knob23": group, "coding": h.spmv_coding()[0], "engine": h.last_engine(), "sweeps": h.last_sweeps(),
               "wall_ms_with_every_launch_timed": round(1e3 * wall, 2), "device_ms": round(t["total_ms"], 2),
               "same_bits_as_uncoded": bool(np.array_equal(a, base[0]) and np.array_equal(b, base[1]))}
        for k in ("spmv", "three_term", "qtw", "update", "final"):
            c = t[k]
            if c["timed_launches"]:
                out[k] = {"us": round(1e3 * c["ms"] / c["timed_launches"], 1), "launches": c["launches"],
                          "GBps": round(c["timed_bytes"] / max(c["ms"], 1e-9) / 1e6, 0) if c["timed_bytes"] else None}
        print(json.dumps(out), flush=True)
        h.close()
