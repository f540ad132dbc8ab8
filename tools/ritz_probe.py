"""Timing probe of the Ritz back-transform kernels (GPU box only): Y = V S for a list of (n, lz_set_tuning(9, variant)) on one
basis, device time from the library's hipEvents + the kernels' own clock record (lz_ritz_info).
    python tools/ritz_probe.py [M_x M_y] -- n:variant ...      e.g.  python tools/ritz_probe.py 1000 1000 -- 100:0 100:7 50:0 50:6"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanczos_amd import _capi, synthetic  # noqa: E402

if os.environ.get("LZ_KBENCH") == "1":  # timing-only arms (variants >= 10) live in the kernel-bench build
    _capi.LIB_PATH = _capi.KBENCH_LIB_PATH

args = sys.argv[1:]
split = args.index("--") if "--" in args else 0
dims = tuple(int(x) for x in args[:split]) or (1000, 1000)
cases = [tuple(int(v) for v in a.split(":")) for a in args[split + 1:]] or [(100, 0)]
A = synthetic.laplacian_2d_5pt(*dims)
M = A.shape[0]
v0 = synthetic.reference_start_vector(M)
v0 /= np.linalg.norm(v0)
out = []
for n in sorted({c[0] for c in cases}):
    h = _capi.Handle(0)
    h.set_options(_capi.FLAG_FUSED_NORM | _capi.FLAG_PROFILE)
    h.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
    a, b = h.run(n, v0)
    S = np.linalg.eigh(np.diag(a) + np.diag(b, 1) + np.diag(b, -1))[1]
    for nn, variant in cases:
        if nn != n:
            continue
        h.set_tuning(_capi.TUNE_RITZ_KERNEL, variant)
        h.ritz_vectors(S, fetch=False)  # first call of a kernel: code-object load
        h.timings()
        ms = []
        for _ in range(5):
            h.ritz_vectors(S, fetch=False)
            ms.append(h.timings()["ritz"]["ms"])
        info = h.ritz_info()
        flops = 2.0 * M * n * n
        best = min(ms)
        rec = {"M": M, "n": n, "variant": variant, "ms_min": round(best, 4), "ms_median": round(float(np.median(ms)), 4),
               "tflops": round(flops / best / 1e9, 2), "frac_of_78.6": round(flops / best / 1e9 / 78.6, 4),
               "gbps_in_plus_out": round(16.0 * M * n / best / 1e6, 1), **{k: round(v, 1) for k, v in info.items()}}
        if info["cycles_per_tile"]:
            rec["util_in_cycles"] = round(info["mfma_floor_cycles_per_tile"] / info["cycles_per_tile"], 4)
        print(json.dumps(rec), flush=True)
        out.append(rec)
    h.close()
