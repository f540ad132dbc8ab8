#!/usr/bin/env python3
"""Golden coefficients of BASELINE config C3 (random-graph Laplacian, M = 1e7, average degree 7, n = 200) by RUNNING THE
REFERENCE's CPU path at full size, here in the build container (needs /root/reference, ~40 GB of memory, ~90 min):

    cd /tmp && MPLBACKEND=Agg python /root/repo/oracle/gen_golden_c3.py

One run (8 BLAS threads).  Unlike the headline Laplacian this spectrum converges at its top within 200 steps, so the late
coefficients are rounding noise in the reference itself; tests/test_gpu_fullsize.py determines the stable prefix with a second
device run from a start vector perturbed in its last bit.  Data only; no reference source travels.
"""
import contextlib
import io
import os
import sys
import time
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/Python"
for _m in ("cupy", "cupyx", "cupyx.scipy", "cupyx.scipy.sparse"):
    sys.modules.setdefault(_m, types.ModuleType(_m))
sys.path.insert(0, os.path.join(REF, "Regular"))
sys.path.insert(0, REPO)

import Lanczos as ref_regular  # noqa: E402  (reference, read-only)

from lanczos_amd import synthetic  # noqa: E402

M, E, N = 10_000_000, 35_000_000, 200
H = synthetic.random_graph_laplacian(M, E, seed=1234).to_scipy()
obj = ref_regular.Lanczos(H)
t = time.time()
with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
    obj.execute_Lanczos(N, use_cuda=False)
print(f"reference run: {time.time() - t:.0f} s", flush=True)
a, b = np.diag(obj.H_eff).copy(), np.diag(obj.H_eff, 1).copy()
out = os.path.join(REPO, "tests", "golden", "c3_graph_M1e7_n200.npz")
np.savez_compressed(out, name="c3_graph_M1e7_n200", M=M, n=N, seed=99, alpha=a, beta=b,
                    generator=f"random_graph_laplacian({M}, {E}, seed=1234); reference Lanczos.execute_Lanczos({N}, use_cuda=False)",
                    numpy_version=np.__version__)
print("saved", out, flush=True)
