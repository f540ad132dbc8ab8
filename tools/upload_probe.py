#!/usr/bin/env python3
"""Host -> device half of the boundary (round 4): wall time and rate of lz_set_csr (validation sweep + upload + SpMV layout),
lz_set_dense and the start-vector upload inside lz_run, with the staged pipeline of lz_xfer.hip (default) or the plain
hipMemcpy path (LZ_XFER_THREADS=0, read once per process: run the tool twice).  One JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanczos_amd import _capi, synthetic  # noqa: E402


def main():
    A = synthetic.laplacian_2d_5pt(4000, 2500)
    M = A.shape[0]
    h = _capi.Handle(0)
    rec = {"LZ_XFER_THREADS": os.environ.get("LZ_XFER_THREADS", "default"), "M": M}
    csr_bytes = A.rowptr.nbytes + A.colidx.nbytes + A.vals.nbytes
    ts = []
    for _ in range(3):
        t = time.perf_counter()
        h.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
        ts.append(time.perf_counter() - t)
    rec["set_csr"] = {"bytes": csr_bytes, "wall_s": [round(x, 4) for x in ts], "gbs_incl_validation_and_layout": round(csr_bytes / min(ts) / 1e9, 1)}
    v0 = synthetic.reference_start_vector(M)
    v0 /= np.linalg.norm(v0)
    h.set_options(_capi.FLAG_REORTH_PARTIAL)
    h.run(4, v0)
    ts = []
    for _ in range(5):
        t = time.perf_counter()
        h.run(2, v0)
        ts.append(time.perf_counter() - t)
    rec["lz_run_n2_incl_v0_upload_ms"] = [round(1e3 * x, 3) for x in ts]
    n = 24000  # 4.6 GB dense
    D = np.random.default_rng(0).standard_normal((n, n))
    ts = []
    for _ in range(3):
        t = time.perf_counter()
        h.set_dense(D)
        ts.append(time.perf_counter() - t)
    rec["set_dense"] = {"bytes": D.nbytes, "wall_s": [round(x, 4) for x in ts], "gbs": round(D.nbytes / min(ts) / 1e9, 1)}
    print(json.dumps(rec))
    h.close()


if __name__ == "__main__":
    main()
