"""What the collectives of each Krylov loop cost per iteration at the N = 8 rank share of the headline, as far as ONE GPU can
tell.  One rank's share (`lap2d_5pt_M1.25e6_k200`: 4000 x 313 grid rows) runs
  (a) with no communicator at all                     -> the loop's compute floor,
  (b) with a 1-rank RCCL communicator, knob 6 forcing every collective the loop would issue at N = 8: ncclAllReduce of the loop's
      reduce buffer and the grouped ncclSend/ncclRecv of the two 4000-double faces (the periodic seam of the slab is routed
      through the ghost tail, so the exchange moves exactly the bytes a middle rank exchanges with its two neighbours) - to itself.
(b) - (a) is what the calls cost in launch + kernel + stream-order overhead with ZERO link latency: a lower bound of the real
per-iteration communication cost.  The projection to 8 GPUs adds an assumed per-collective xGMI latency on top and is labelled as
such; nothing here measures a second device.
usage: python tools/collective_budget.py [out.json] [c4]      (c4: one rank's slab of BASELINE config C4 instead - 500 x 500 x 50 of the
500 x 500 x 400 7-point grid, M = 1.25e7 per rank, two faces of 250 000 doubles = 2 MB each, k = 200)"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanczos_amd import _capi, synthetic  # noqa: E402

C4 = len(sys.argv) > 2 and sys.argv[2] == "c4"
if C4:
    gx, gy, gz, n = 500, 500, 50, 200
    A = synthetic.laplacian_3d_7pt(gx, gy, gz)
    nx = gx * gy  # a face = one xy plane
else:
    nx, ny, n = 4000, 313, 200
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

F, P, O, V = _capi.FLAG_FUSED_NORM, _capi.FLAG_REORTH_PARTIAL, _capi.FLAG_ONE_REDUCE, _capi.FLAG_OVERLAP_HALO
LOOPS = [("full_default", F), ("full_one_reduce", F | O), ("full_overlap_halo", F | V), ("partial_device", P), ("partial_one_reduce", P | O)]
out = {"workload": ("lap3d_7pt 500x500x50 slab, k200 (one rank's share of BASELINE C4 at N = 8)" if C4 else "lap2d_5pt_M1.25e6_k200 (one rank's share of the headline at N = 8)"),
       "M": M, "k": n, "face_doubles": nx, "loops": {}}
for name, flags in LOOPS:
    rec = {}
    for comm in (False, True):
        if not comm and (flags & V):
            continue
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
        runs = []
        for _ in range(3 if C4 else 7):
            a, b = h.run(n, v0)
            t = h.timings()
            runs.append(t["total_ms"])
        runs.sort()
        key = "rccl_1rank" if comm else "no_comm"
        rec[key] = {"ms_per_solve_min": round(runs[0], 3), "ms_per_solve_median": round(runs[len(runs) // 2], 3),
                    "us_per_iteration": round(1e3 * runs[0] / n, 2), "engine": h.last_engine(), "sweeps": h.last_sweeps(),
                    "host_syncs": h.last_host_syncs()}
        if comm:
            rec[key].update({"allreduces_per_iteration": round(t["allreduces"] / n, 3), "exchanges_per_iteration": round(t["exchanges"] / n, 3)})
        h.close()
    if "no_comm" in rec:
        d = rec["rccl_1rank"]["us_per_iteration"] - rec["no_comm"]["us_per_iteration"]
        calls = rec["rccl_1rank"]["allreduces_per_iteration"] + rec["rccl_1rank"]["exchanges_per_iteration"]
        rec["collectives_us_per_iteration_zero_latency"] = round(d, 2)
        rec["us_per_collective_zero_latency"] = round(d / calls, 2)
    out["loops"][name] = rec
    print(name, json.dumps(rec), flush=True)

# ---- projection (NOT a measurement): N = 8 time per iteration = this rank share's loop with its collectives issued (1-rank RCCL,
# zero link latency) + an assumed extra latency per collective for the 8-rank ring / neighbour exchange over xGMI
n1 = {"full": None, "partial": None}
proj = {}
for lat in (5.0, 10.0, 20.0, 40.0):
    row = {}
    for name, rec in out["loops"].items():
        r1 = rec["rccl_1rank"]
        calls = r1["allreduces_per_iteration"] + r1["exchanges_per_iteration"]
        row[name] = round(r1["us_per_iteration"] + lat * calls, 1)
    proj[f"{lat:.0f}us_extra_per_collective"] = row
out["projection_us_per_iteration_at_N8"] = proj
out["projection_note"] = ("PROJECTION, not a measurement: 1-rank-RCCL time of the rank share + (assumed extra link latency) x (collectives per "
                          "iteration); divide the N = 1 time per iteration of the same loop (bench.py on the full headline) by it for a speed-up")
path = sys.argv[1] if len(sys.argv) > 1 else None
if path:
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out["projection_us_per_iteration_at_N8"], indent=1))
