#!/usr/bin/env python3
"""Gram matrix G = Y^T Y of the Ritz vectors at the headline size for a list of knob-19 arms (0 symmetric kernel, 1 split-K TN GEMM;
round 4 also measured an eight-wave arm with it: profiles/r04/ab_gram_eight_waves.jsonl), event-timed, second call of each arm.
usage: gram_probe.py [n] [arms, e.g. 0,1]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanczos_amd import _capi, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
arms = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,1").split(",")]
A = synthetic.laplacian_2d_5pt(4000, 2500)
M = A.shape[0]
v0 = synthetic.reference_start_vector(M)
v0 /= np.linalg.norm(v0)
h = _capi.Handle(0)
h.set_options(_capi.FLAG_FUSED_NORM | _capi.FLAG_PROFILE)
h.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
a, b = h.run(n, v0)
S = np.linalg.eigh(np.diag(a) + np.diag(b, 1) + np.diag(b, -1))[1]
h.ritz_vectors(S, fetch=False)
ref = None
for arm in arms * 2:
    h.set_tuning(_capi.TUNE_GRAM_KERNEL, arm)
    h.ritz_gram()
    h.timings()
    G = h.ritz_gram()
    t = h.timings()["ritz"]
    ref = G if ref is None else ref
    print(json.dumps({"knob19": arm, "n": n, "ms": round(t["ms"], 3), "max_dev_from_identity": float(np.abs(G - np.eye(n)).max()),
                      "max_diff_vs_first_arm": float(np.abs(G - ref).max()), **h.gram_info()}), flush=True)
h.close()
