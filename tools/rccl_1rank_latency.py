"""Lower bound on the per-call cost of the RCCL collectives the multi-GPU path issues: a 1-rank communicator on the
compute stream (forced through tuning knob 6), timed by the library's own hipEvents.  No inter-GPU latency in this
number - only launch + kernel overhead of ncclAllReduce / grouped ncclSend+ncclRecv."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanczos_amd import _capi, synthetic  # noqa: E402

nx, ny = 4000, 313
A = synthetic.laplacian_2d_5pt(nx, ny)
M = A.shape[0]
v0 = np.random.RandomState(99).uniform(-1, 1, M)
v0 /= np.linalg.norm(v0)
out = {}
for name, flags in (("unfused", 0), ("fused_norm", _capi.FLAG_FUSED_NORM)):
    h = _capi.Handle(0)
    h.comm_init_rccl(1, 0, h.unique_id())
    h.set_tuning(_capi.TUNE_FORCE_COLLECTIVES, 1)
    h.set_options(_capi.FLAG_PROFILE | flags)
    rows_pad = h.padded_rows(M)
    h.set_csr(M, 0, A.rowptr, A.colidx, A.vals, ncols_ext=rows_pad)
    h.set_allgather(rows_pad)
    h.run(20, v0)
    h.timings()
    h.run(100, v0)
    t = h.timings()
    c = t["comm"]
    out[name] = {"comm_calls": c["launches"], "comm_avg_us": 1e3 * c["ms"] / max(c["timed_launches"], 1), "total_ms": t["total_ms"],
                 "per_iteration_us": 10 * t["total_ms"]}
    h.close()
print(json.dumps(out, indent=1))
