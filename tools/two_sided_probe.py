"""Timing probe of the two-sided (bi-orthogonal) Lanczos path (GPU box only; parity lives in tests/test_gpu_two_sided.py)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanczos_amd import IrrLanczos, _capi, synthetic  # noqa: E402

_capi.LIB_PATH = _capi.KBENCH_LIB_PATH  # the single-launch / deferred-fold link arms live in the kernel-bench build (round 3)

out = {}
IrrLanczos.verbose = False
for (nx, ny, n, arm) in [(1000, 1000, 50, 2), (1000, 1000, 50, 1), (4000, 2500, 24, 2), (4000, 2500, 24, 1), (300, 300, 60, 2), (300, 300, 60, 1), (512, 512, 60, 2), (512, 512, 60, 1)]:
    A = synthetic.laplacian_2d_5pt(nx, ny).to_scipy()
    s = IrrLanczos(A)
    s.options = _capi.FLAG_PROFILE
    s.execute_Lanczos(4, seed=1)  # warm-up (code objects)
    s._handle.set_tuning(_capi.TUNE_BI_LINKS, arm)  # 2 = single-launch links (last block folds the partials), 1 = separate fold kernel
    s.execute_Lanczos(4, seed=1)
    t0 = time.perf_counter()
    s.execute_Lanczos(n, seed=1)
    wall = time.perf_counter() - t0
    tm = s._timings
    M = nx * ny
    q = tm["qtw"]
    out[f"M{M}_n{n}_{'two_launch' if arm == 1 else 'one_launch'}"] = {"alpha_tail": s._alpha[-3:].tolist(),"wall_s": wall, "device_ms": tm["total_ms"], "biorth_ms": q["ms"], "biorth_GBps": q["timed_bytes"] / q["ms"] / 1e6 if q["ms"] else None,
                         "spmv_ms": tm["spmv"]["ms"], "two_term_ms": tm["three_term"]["ms"], "finite": bool(np.isfinite(s.H_eff).all())}
print(json.dumps(out))
