"""How fast do the results reach the caller?  (GPU box.)  Times ``.V`` and ``.H_eigvecs`` (the two M x n arrays the reference
publishes, Lanczos.py:132-141,153-156) after a run: D2H through the C ABI into fresh NumPy memory.
    python tools/publish_probe.py [M_x M_y n]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanczos_amd import Lanczos, _capi, synthetic  # noqa: E402

mx, my, n = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (4000, 2500, 100)
H = synthetic.laplacian_2d_5pt(mx, my)
Lanczos.verbose = False
s = Lanczos(H.to_scipy())
t = time.perf_counter()
s.execute_Lanczos(n)
t_run = time.perf_counter() - t
t = time.perf_counter()
s.get_H_eigs()
t_ritz = time.perf_counter() - t
gb = 8e-9 * s.M * n
out = {"M": s.M, "n": n, "array_GB": round(gb, 2), "execute_s": round(t_run, 3), "get_H_eigs_s": round(t_ritz, 3)}
for name in ("V", "H_eigvecs"):
    t = time.perf_counter()
    a = getattr(s, name)
    dt = time.perf_counter() - t
    out[name + "_s"] = round(dt, 3)
    out[name + "_GBps"] = round(gb / dt, 2)
    assert a.shape == (s.M, n)
    del a
h = s._handle
for rep in range(2):  # the raw C-ABI copy into an array that is already faulted in
    buf = np.zeros((n, s.M))
    t = time.perf_counter()
    h.check(h.lib.lz_get_basis(h._h, _capi.dptr(buf), s.M))
    dt = time.perf_counter() - t
    out["lz_get_basis_into_touched_memory_GBps_%d" % rep] = round(gb / dt, 2)
print(json.dumps(out))
