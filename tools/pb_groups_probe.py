#!/usr/bin/env python3
"""A/B of the two-phase SpMV's row-block-group arm (knob 22) on the C3 matrix: products(g), rows(g) interleaved over
G groups of row blocks so that a group's segment of the product stream T2 (+ x) could stay in the 256 MB Infinity Cache between the
phases.  Per G: event time per SpMV (20 launches), y against the default's bit for bit.  The arm measured slower for every G and lives in the
kernel-bench build only (make KBENCH=1): this tool loads liblanczos_kbench.so.
usage: pb_groups_probe.py [rows] [G,G,...]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanczos_amd import _capi, synthetic  # noqa: E402

M = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
groups = [int(g) for g in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 2, 3, 4, 5, 6, 8, 12, 0]
KB = _capi.load_library(_capi.KBENCH_LIB_PATH)
A = synthetic.random_graph_laplacian(M, int(3.5 * M), seed=1234)
xs = np.random.default_rng(0).standard_normal(M)
ref = None
for G in groups:
    h = _capi.Handle(0, lib=KB)
    h.set_options(_capi.FLAG_PROFILE)
    h.set_tuning(_capi.TUNE_SPMV_PLAN, 2)
    h.set_tuning(_capi.TUNE_PB_GROUPS, G)
    h.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
    assert h.spmv_plan() == "two-phase"
    y = h.spmv_host(xs)
    if ref is None:
        ref = y
    h.basis_alloc(2)
    h.basis_set_row(1, xs)
    for _ in range(3):
        h.step_spmv(1)
    h.timings()
    for _ in range(20):
        h.step_spmv(1)
    t = h.timings()["spmv"]
    us = 1e3 * t["ms"] / max(t["timed_launches"], 1)
    nnz = int(A.rowptr[-1])
    alg = 12.0 * nnz + 4.0 * (M + 1) + 16.0 * M
    print(json.dumps({"rows": M, "groups": G, "spmv_us": round(us, 1), "frac_of_8TBps": round(alg / us / 1e3 / 8000.0, 4),
                      "y_bit_identical_to_default": bool(np.array_equal(y, ref))}), flush=True)
    h.close()
