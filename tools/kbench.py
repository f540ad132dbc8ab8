#!/usr/bin/env python3
"""Kernel A/B harness (GPU box): times the hot kernels one class at a time at the headline
size through the lz_step_* entry points with LZ_FLAG_PROFILE events, for a list of
(flags, tuning) arms, interleaved in one process.

    python tools/kbench.py --nx 4000 --ny 2500 --rows 101 --reps 5 --arms "valu:0:;mfma:2:;..."
arm syntax: name:flags:idx=val,idx=val
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanczos_amd import _capi, synthetic  # noqa: E402

# the kernel-bench build of the library (make -C lanczos_amd/csrc KBENCH=1): the retired A/B kernels; the timing-only ablation arms
# (knob 1 >= 20, knob 3, knob 9 >= 10) of rounds 1-4 no longer exist in either build (deleted in round 5; results in profiles/)
_KB = os.path.join(os.path.dirname(_capi.LIB_PATH), "liblanczos_kbench.so")
if os.path.isfile(_KB):
    _capi.LIB_PATH = _KB


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=4000)
    ap.add_argument("--ny", type=int, default=2500)
    ap.add_argument("--rows", type=int, default=101, help="basis rows present (j = rows-1)")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--arms", default="valu:0:")
    ap.add_argument("--what", default="reorth,spmv,three")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--one-handle", action="store_true", help="all arms share ONE handle/allocation (only per-launch knobs: tune 1, 8); "
                    "separate allocations differ by +-3 % on their own")
    ap.add_argument("--matrix", default="lap2d", help="lap2d | lap3d:N | deuteron27:N | graph:M")
    args = ap.parse_args()
    if args.matrix.startswith("lap3d:"):
        N = int(args.matrix.split(":")[1])
        A = synthetic.laplacian_3d_7pt(N, N, N)
    elif args.matrix.startswith("deuteron27:"):
        from lanczos_amd import Hamiltonian

        N = int(args.matrix.split(":")[1])
        Hamiltonian.verbose = False
        Hamiltonian.vectorize_potential = True
        os.chdir("/tmp")
        Hs = Hamiltonian(N, 25, synthetic.deuteron_potential, 197.327**2 / (2 * 469.4592) / (25.0 / N) ** 2).build_H("27")
        A = synthetic.CSR(Hs.indptr, Hs.indices, Hs.data, Hs.shape)
    elif args.matrix.startswith("graph:"):
        Mg = int(args.matrix.split(":")[1])
        A = synthetic.random_graph_laplacian(Mg, int(3.5 * Mg), seed=1234)
    else:
        A = synthetic.laplacian_2d_5pt(args.nx, args.ny)
    M = A.shape[0]
    rng = np.random.default_rng(0)
    arms = []
    for spec in args.arms.split(";"):
        if not spec:
            continue
        name, flags, tune = (spec.split(":") + ["", ""])[:3]
        tune = [tuple(int(x) for x in kv.split("=")) for kv in tune.split(",") if kv]
        arms.append((name, int(flags or 0), tune))
    handles = []
    n = args.rows
    rows = [rng.standard_normal(M) / np.sqrt(M) for _ in range(3)]
    for name, flags, tune in arms:
        if args.one_handle and handles:
            handles.append(handles[0])
            continue
        h = _capi.Handle(0)
        h.set_options(flags | _capi.FLAG_PROFILE)
        for idx, val in tune:
            h.set_tuning(idx, val)
        h.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
        h.basis_alloc(n)
        for i in range(n):
            h.basis_set_row(i, rows[i % 3] * (1 + 0.01 * i))
        h.r_set(rows[0])
        h.timings()
        handles.append(h)
    res = {name: {} for name, _, _ in arms}
    cref = None
    for rep in range(args.reps + 1):
        for (name, flags, tune), h in zip(arms, handles):
            if args.one_handle:
                for idx in (1, 8):
                    h.set_tuning(idx, dict(tune).get(idx, 0))
                h.timings()
            if "reorth" in args.what:
                _, c = h.step_reorth(n - 1, n, scale=False)
                if args.check:
                    if cref is None:
                        cref = c.copy()
                    else:
                        res[name]["c_maxdiff_vs_first"] = float(np.abs(c - cref).max() / np.abs(cref).max())
                h.basis_set_row(n - 1, rows[(n - 1) % 3] * (1 + 0.01 * (n - 1)))
            if "spmv" in args.what:
                h.step_spmv(n // 2)
            if "three" in args.what:
                h.step_three_term(1, 0, 0.5, 0.25)
                h.r_set(rows[0])
            t = h.timings()
            if rep == 0:
                continue  # warm-up
            for k in ("spmv", "qtw", "update", "three_term", "final"):
                if t[k]["launches"]:
                    d = res[name].setdefault(k, {"us": [], "bytes": t[k]["bytes"] / t[k]["launches"]})
                    assert t[k]["timed_launches"] == t[k]["launches"]
                    d["us"].append(1e3 * t[k]["ms"] / t[k]["launches"])
    for name in res:
        for k, d in res[name].items():
            if not isinstance(d, dict):
                continue
            us = np.array(d["us"])
            d["us_med"], d["us_min"] = float(np.median(us)), float(us.min())
            d["GBps_med"] = d["bytes"] / d["us_med"] / 1e3 if d["bytes"] else None
            d["frac_8TBps"] = d["GBps_med"] / 8000 if d["bytes"] else None
            del d["us"]
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
