"""Where does a first call through the class surface wait?  While a headline-size solver is alive (32 GB of big buffers), fresh
`Lanczos(H)` objects run their first `execute_Lanczos(200)`; LZ_DEBUG_TIMING=1 makes the library print the phases of the matrix upload
(validation sweep, device allocation, H2D, row-class detection) and of lz_run (allocation, v0 upload, enqueue, drain).
usage: LZ_DEBUG_TIMING=1 python tools/first_call_probe.py [repeats]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lanczos_amd  # noqa: E402
from lanczos_amd import synthetic  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
H = synthetic.laplacian_2d_5pt(4000, 2500).to_scipy()
lanczos_amd.Lanczos.verbose = False
keeper = lanczos_amd.Lanczos(H)
keeper.execute_Lanczos(200)
_ = keeper.H_eigvals  # (the Ritz vectors are formed on the device: another 16 GB big buffer stays alive)
for r in range(reps):
    s = lanczos_amd.Lanczos(H)
    t0 = time.perf_counter()
    s.execute_Lanczos(200)
    t = time.perf_counter() - t0
    dev = s._timings["total_ms"] / 1e3 if getattr(s, "_timings", None) else float("nan")
    print("first call %d: wall %.3f s, device %.3f s, overhead %.3f s" % (r, t, dev, t - dev), file=sys.stderr, flush=True)
    s.close()
keeper.close()
