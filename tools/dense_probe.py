import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanczos_amd import _capi
for M in (4096, 16384, 32768, 32767):
    A = np.random.default_rng(0).standard_normal((M, M)); A = (A + A.T) / 2
    h = _capi.Handle(0); h.set_options(_capi.FLAG_PROFILE); h.set_dense(A)
    h.basis_alloc(2); x = np.random.default_rng(1).standard_normal(M); h.basis_set_row(1, x)
    for _ in range(2): h.step_spmv(1)
    h.timings()
    for _ in range(5): a = h.step_spmv(1)
    t = h.timings()["spmv"]; us = 1e3 * t["ms"] / t["launches"]
    y = h.r_get(); ref = A @ x
    print(M, "us", round(us, 1), "GB/s", round(t["bytes"] / t["launches"] / us / 1e3), "maxerr", float(np.abs(y - ref).max() / np.abs(ref).max()), flush=True)
    h.close()
