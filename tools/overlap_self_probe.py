"""Does the halo exchange really run BEHIND the interior update?  One GPU, 1-rank RCCL communicator exchanging with
itself (the periodic seam of a 2-D stencil is routed through the ghost tail), one rank's share of the headline at N=8.
Compares device time per iteration: no collectives at all / collectives in line / LZ_FLAG_OVERLAP_HALO."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanczos_amd import _capi, synthetic  # noqa: E402

nx, ny, n = 4000, int(sys.argv[1]) if len(sys.argv) > 1 else 313, 100
A = synthetic.laplacian_2d_5pt(nx, ny)
M = A.shape[0]
v0 = np.random.RandomState(99).uniform(-1, 1, M)
v0 /= np.linalg.norm(v0)
rows_pad = (M + 31) // 32 * 32
row_of = np.repeat(np.arange(M), np.diff(A.rowptr))
wrap = np.abs(A.colidx.astype(np.int64) - row_of) > nx
ghost_cols = np.unique(A.colidx[wrap])
col = A.colidx.astype(np.int64).copy()
col[wrap] = rows_pad + np.searchsorted(ghost_cols, A.colidx[wrap])
out = {"M": M, "n": n}
ref = None
for name, flags, comm in (("no_comm", 0, False), ("inline", _capi.FLAG_FUSED_NORM, True), ("overlap", _capi.FLAG_FUSED_NORM | _capi.FLAG_OVERLAP_HALO, True),
                          ("inline2", _capi.FLAG_FUSED_NORM, True), ("overlap2", _capi.FLAG_FUSED_NORM | _capi.FLAG_OVERLAP_HALO, True)):
    h = _capi.Handle(0)
    if comm:
        h.comm_init_rccl(1, 0, h.unique_id())
        h.set_tuning(_capi.TUNE_FORCE_COLLECTIVES, 1)
        h.set_options(flags)
        h.set_csr(M, 0, A.rowptr, col.astype(np.int32), A.vals, ncols_ext=rows_pad + len(ghost_cols))
        h.set_halo([0, 0], [nx, nx], ghost_cols.astype(np.int32), [nx, nx])
    else:
        h.set_options(flags)
        h.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
    h.run(20, v0)
    h.timings()
    best = 1e9
    for _ in range(5):
        a, b = h.run(n, v0)
        best = min(best, h.timings()["total_ms"])
    out[name] = {"us_per_iteration": 1e3 * best / n}
    if ref is None:
        ref = a
    out[name]["max_alpha_diff"] = float(np.abs(a - ref).max())
    h.close()
print(json.dumps(out, indent=1))
