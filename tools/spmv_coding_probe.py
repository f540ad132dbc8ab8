"""Row-class coded SpMV (round 5) against the uncoded kernels, one MI355X: device time per SpMV (the library's hipEvents) for knob 17 =
0 (auto: coded where the rows fall into classes; knob 23 = 0 two adjacent rows per lane, 1 / 3 one row per lane with one / two units per workgroup), 1 (CSR-order fixed-K kernel / CSR-stream), 2 (uncoded
ELL), 4 (offsets-only coding);
y checked bit for bit against knob 1.   python tools/spmv_coding_probe.py > gpurun_out/spmv_coding_probe.jsonl"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanczos_amd import _capi, synthetic  # noqa: E402

CASES = [("lap2d_5pt_4000x2500", (4000, 2500, 1), 5, None), ("lap2d_5pt_1000x1000", (1000, 1000, 1), 5, None), ("lap3d_7pt_300^3", (300, 300, 300), 7, None),
         ("stencil3d_27pt_160^3_potential", (160, 160, 160), 27, "potential"), ("stencil3d_7pt_464^3", (464, 464, 464), 7, None)]
only = os.environ.get("LZ_CASES")
for name, dims, pts, pot in CASES:
    if only and name not in only.split(","):
        continue
    M = int(np.prod(dims))
    xs = np.random.default_rng(0).uniform(-1, 1, M)
    ref = None
    for knob, group in ((1, 0), (2, 0), (4, 0), (0, 1), (0, 3), (0, 0)):
        h = _capi.Handle(0)
        h.set_options(_capi.FLAG_FUSED_NORM | _capi.FLAG_PROFILE)
        h.set_tuning(_capi.TUNE_FIXED_LAYOUT, knob)
        h.set_tuning(_capi.TUNE_CLS_GROUP, group)
        if pts == 5:
            A = synthetic.laplacian_2d_5pt(dims[0], dims[1])
            h.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
            del A
        elif pot:
            V = np.random.default_rng(3).uniform(-1, 0, M)
            h.build_stencil3d_block(dims, pts, 1.0, (-6.0, 0.5, 0.25, 0.125), 0, M, (), potential=V)
        else:
            h.build_stencil3d_block(dims, pts, 1.0, (-6.0, 1.0, 0.0, 0.0), 0, M, ())
        coding = h.spmv_coding()
        h.basis_alloc(2)
        y = h.spmv_host(xs)
        if ref is None:
            ref = y
        same = bool(np.array_equal(y, ref))
        h.basis_set_row(0, xs)
        a0 = h.step_spmv(0)
        h.timings()
        for _ in range(10):
            h.step_spmv(0)
        t = h.timings()["spmv"]
        us = 1e3 * t["ms"] / max(t["timed_launches"], 1)
        fmt_bytes = t["timed_bytes"] / max(t["timed_launches"], 1)
        print(json.dumps({"case": name, "knob17": knob, "knob23": group, "coding": coding[0], "classes": coding[1], "rows": M, "K": pts, "spmv_us": round(us, 1),
                          "format_bytes_MB": round(fmt_bytes / 1e6, 1), "format_GBps": round(fmt_bytes / us / 1e3, 1),
                          "csr_bytes_MB": round((12 * pts + 16) * M / 1e6, 1), "y_equals_knob1_bits": same, "alpha": a0}), flush=True)
        h.close()
