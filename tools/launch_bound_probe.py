#!/usr/bin/env python3
"""Is the small-problem loop bound by the host's enqueue rate or by the GPU's dependent-dispatch rate?
Runs config C1 (dense 512 x 512, k = 20) and a 1-D box (N = 500, n = 50) with LZ_DEBUG_TIMING=1: lz_run prints how long the
host needed to enqueue the loop and how long it then waited for the GPU."""
import os
import sys
import time

import numpy as np

os.environ["LZ_DEBUG_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanczos_amd import _capi, synthetic  # noqa: E402

rng = np.random.default_rng(0)
for name, M, n in (("c1_dense512_k20", 512, 20), ("box1d_N500_n50", 500, 50)):
    B = rng.standard_normal((M, M))
    H = (B + B.T) / 2
    v0 = rng.standard_normal(M)
    v0 /= np.linalg.norm(v0)
    for knob in (0, 1):
        h = _capi.Handle(0)
        h.set_options(_capi.FLAG_FUSED_NORM)
        h.set_tuning(_capi.TUNE_LOOP, knob)
        h.set_dense(H)
        for rep in range(4):
            t = time.perf_counter()
            h.run(n, v0)
            dt = time.perf_counter() - t
            print(f"{name} knob15={knob} rep {rep}: {dt * 1e3:.3f} ms wall, {dt / n * 1e6:.1f} us/step", file=sys.stderr)
        h.close()
