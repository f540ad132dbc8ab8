#!/usr/bin/env python3
"""Golden coefficients of the BASELINE headline (2-D periodic 5-point Laplacian 4000 x 2500, M = 1e7, n = 200) by RUNNING THE
REFERENCE's CPU path at full size, here in the build container (needs /root/reference and ~20 GB of memory, ~15 min on 8 cores):

    cd /tmp && MPLBACKEND=Agg python /root/repo/oracle/gen_golden_headline.py

Two runs of the reference with different BLAS thread counts (the dot products are then summed in a different order, i.e. the
input of every step is perturbed at rounding level): the fixture stores the first run's alpha / beta and, per coefficient,
how far the second run moved - the reference's own noise floor at this size.  Data only; no reference source travels.
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
from threadpoolctl import threadpool_limits  # noqa: E402

from lanczos_amd import synthetic  # noqa: E402

NX, NY, N = 4000, 2500, 200


def run(threads):
    H = synthetic.laplacian_2d_5pt(NX, NY).to_scipy()
    with threadpool_limits(limits=threads):
        obj = ref_regular.Lanczos(H)
        t = time.time()
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            obj.execute_Lanczos(N, use_cuda=False)
        print(f"reference run with {threads} BLAS threads: {time.time() - t:.0f} s", flush=True)
    a, b = np.diag(obj.H_eff).copy(), np.diag(obj.H_eff, 1).copy()
    del obj
    return a, b


a8, b8 = run(8)
a3, b3 = run(3)
out = os.path.join(REPO, "tests", "golden", "headline_lap2d_4000x2500_n200.npz")
np.savez_compressed(out, name="headline_lap2d_4000x2500_n200", M=NX * NY, n=N, seed=99, alpha=a8, beta=b8,
                    alpha_moved=np.abs(a8 - a3), beta_moved=np.abs(b8 - b3),
                    generator=f"laplacian_2d_5pt({NX}, {NY}); reference Lanczos.execute_Lanczos({N}, use_cuda=False), 8 vs 3 BLAS threads",
                    numpy_version=np.__version__)
print("max moved:", np.abs(a8 - a3).max(), np.abs(b8 - b3).max(), "->", out)
