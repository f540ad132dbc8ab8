#!/usr/bin/env python3
"""How fast are the SpMV's random x gathers as a function of where x lives?  Random-graph Laplacians of
growing M (avg degree 7): x = 8 M bytes moves from L2 to Infinity Cache to HBM."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanczos_amd import _capi, synthetic

out = {}
for M in [int(a) for a in sys.argv[1:]] or [250_000, 1_000_000, 4_000_000, 10_000_000, 30_000_000]:
    A = synthetic.random_graph_laplacian(M, int(3.5 * M), seed=1234)
    for name, flags in (("stream", 0), ("scalar", 8)):
        h = _capi.Handle(0)
        h.set_options(flags | _capi.FLAG_PROFILE)
        h.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
        h.basis_alloc(2)
        h.basis_set_row(1, np.random.default_rng(0).standard_normal(M))
        for _ in range(3):
            h.step_spmv(1)
        h.timings()
        for _ in range(10):
            h.step_spmv(1)
        t = h.timings()["spmv"]
        us = 1e3 * t["ms"] / t["launches"]
        out[f"{M}:{name}"] = {"x_MB": 8 * M / 1e6, "us": round(us, 1), "G_gathers_per_s": round((A.nnz - M) / us / 1e3, 1),
                              "alg_GBps": round(t["bytes"] / t["launches"] / us / 1e3, 1)}
        print(M, name, out[f"{M}:{name}"], flush=True)
        h.close()
json.dump(out, open("gpurun_out/gather_probe.json", "w"), indent=1)
