#!/usr/bin/env python3
"""The partial re-orthogonalisation loop alone (headline matrix unless --workload says otherwise): wall and device time per
solve, per-class event timings, engine, host synchronisations, for a list of knob settings in ONE process.
    python tools/partial_probe.py --arms ";18=1;18=2;17=1" --reps 3
Under rocprofv3 (`rocprofv3 --kernel-trace --stats -d DIR -- python3 tools/partial_probe.py --reps 2 --no-profile`) the per-kernel
table shows the gated launches of the steps without a sweep next to the kernels that do the work."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanczos_amd import _capi, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=4000)
    ap.add_argument("--ny", type=int, default=2500)
    ap.add_argument("--nz", type=int, default=0, help="> 0: 3-D 7-point Laplacian nx x ny x nz")
    ap.add_argument("--k", type=int, default=200)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--arms", default="", help="';'-separated knob lists, e.g. ';18=1;18=2,17=1' (empty = defaults)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--stride", type=int, default=8)
    args = ap.parse_args()
    A = synthetic.laplacian_3d_7pt(args.nx, args.ny, args.nz) if args.nz else synthetic.laplacian_2d_5pt(args.nx, args.ny)
    M = A.shape[0]
    v0 = synthetic.reference_start_vector(M)
    v0 /= np.linalg.norm(v0)
    for arm in args.arms.split(";"):
        h = _capi.Handle(0)
        for kv in filter(None, arm.split(",")):
            i, v = kv.split("=")
            h.set_tuning(int(i), int(v))
        h.set_tuning(_capi.TUNE_PROFILE_STRIDE, args.stride)
        h.set_options(_capi.FLAG_REORTH_PARTIAL | (0 if args.no_profile else _capi.FLAG_PROFILE))
        h.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
        a, b = h.run(args.k, v0)
        h.timings()
        h.synchronize()
        t = time.perf_counter()
        for _ in range(args.reps):
            a, b = h.run(args.k, v0)
        h.synchronize()
        t = (time.perf_counter() - t) / args.reps
        tm = h.timings()
        rec = {"arm": arm or "default", "M": M, "k": args.k, "wall_ms_per_solve": round(1e3 * t, 3), "device_ms_per_solve": round(tm["total_ms"] / args.reps, 3),
               "engine": h.last_engine(), "host_syncs": h.last_host_syncs(), "sweeps": h.last_sweeps(), "spmv_plan": h.spmv_plan(),
               "alpha_sum": float(a.sum()), "beta_sum": float(b.sum())}
        for c in ("spmv", "qtw", "update", "three_term", "final"):
            n = max(tm[c]["timed_launches"], 1)
            rec[c] = {"avg_us": round(1e3 * tm[c]["ms"] / n, 2), "launches_per_solve": tm[c]["launches"] / args.reps,
                      "timed_gbps": round(tm[c]["timed_bytes"] / max(tm[c]["ms"], 1e-9) / 1e6, 1)}
        print(json.dumps(rec), flush=True)
        h.close()


if __name__ == "__main__":
    main()
