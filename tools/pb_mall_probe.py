"""Would keeping the product stream T2 of the two-phase SpMV inside the Infinity Cache pay?  
The proposed arm interleaves the two phases over row-block groups so that a group's T2 segment (+ x) stays resident in the 256 MB
memory-side cache between k_pb_products and k_pb_rows.  The most favourable case for that idea needs no new kernel: a random graph
SMALL enough that the WHOLE of T2 and x fit the cache - no grouping, no extra launches, no re-reads of x.  This probe times the
default two-phase SpMV on Erdos-Renyi graphs of average degree 7 (the C3 family) from 10^6 to 10^7 rows: if cache residency of T2
bought time, the small graphs would run at fewer picoseconds per entry than C3.  Run under `rocprofv3 --pmc FETCH_SIZE` /
`--pmc WRITE_SIZE` (separate passes) the same launches give the HBM bytes that back the residency claim.
usage: python tools/pb_mall_probe.py [rows ...]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanczos_amd import _capi, synthetic  # noqa: E402

sizes = [int(float(a)) for a in sys.argv[1:]] or [1_000_000, 2_000_000, 4_000_000, 10_000_000]
for M in sizes:
    A = synthetic.random_graph_laplacian(M, int(3.5 * M), seed=1234)
    h = _capi.Handle(0)
    h.set_options(_capi.FLAG_PROFILE)
    h.set_tuning(_capi.TUNE_SPMV_PLAN, 2)
    h.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
    assert h.spmv_plan() == "two-phase"
    h.basis_alloc(2)
    h.basis_set_row(1, np.random.default_rng(0).standard_normal(M))
    for _ in range(3):
        h.step_spmv(1)
    h.timings()
    reps = 20
    for _ in range(reps):
        h.step_spmv(1)
    t = h.timings()["spmv"]
    us = 1e3 * t["ms"] / max(t["timed_launches"], 1)
    nnz = int(A.rowptr[-1])
    alg = 12.0 * nnz + 4.0 * (M + 1) + 16.0 * M
    print(json.dumps({"rows": M, "nnz": nnz, "spmv_us": round(us, 1), "ps_per_entry": round(1e6 * us / nnz, 3), "algorithmic_GB": round(alg / 1e9, 4),
                      "frac_of_8TBps": round(alg / us / 1e3 / 8000.0, 4), "x_MB": round(8 * M / 1e6, 1),
                      "T2_MB_estimate": round(8 * 1.16 * (nnz - M) / 1e6, 1)}), flush=True)
    h.close()
