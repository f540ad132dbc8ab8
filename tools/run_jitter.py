import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanczos_amd import _capi, synthetic
nx, ny, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
A = synthetic.laplacian_2d_5pt(nx, ny); M = A.shape[0]
h = _capi.Handle(0); h.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
v0 = np.random.RandomState(99).uniform(-1, 1, M); v0 /= np.linalg.norm(v0)
for mode in ["plain", "devsync", "timings", "sleep10ms", "plain"]:
    ts = []
    for i in range(6):
        if mode == "devsync": h.synchronize()
        if mode == "timings": h.timings()
        if mode == "sleep10ms": time.sleep(0.01)
        t = time.perf_counter(); h.run(k, v0); ts.append(1e3 * (time.perf_counter() - t))
    print(mode, [round(x, 1) for x in ts])
