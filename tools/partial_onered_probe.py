"""One-GPU probe of the one-reduce partial loop (LZ_FLAG_REORTH_PARTIAL | LZ_FLAG_ONE_REDUCE, lz_last_engine 8) against the
three-collective device loop and the full sweep: coefficients, sweeps, look-ahead misses, basis orthogonality.
usage: python tools/partial_onered_probe.py [kappa]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanczos_amd import _capi, synthetic  # noqa: E402

kappa = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = [("lap3d_20x18x16", synthetic.laplacian_3d_7pt(20, 18, 16), 120),
         ("lap2d_96x80", synthetic.laplacian_2d_5pt(96, 80), 60),
         ("graph_50k", synthetic.random_graph_laplacian(50000, 175000, seed=3), 150),
         ("lap2d_1000x1000", synthetic.laplacian_2d_5pt(1000, 1000), 100),
         ("graph_1e6", synthetic.random_graph_laplacian(1000000, 3500000, seed=5), 200)]
for name, A, n in cases:
    M = A.shape[0]
    v0 = np.random.RandomState(99).uniform(-1, 1, M)
    v0 /= np.linalg.norm(v0)
    out = {"case": name, "n": n}
    ref = None
    for tag, flags in (("full", _capi.FLAG_FUSED_NORM), ("partial", _capi.FLAG_REORTH_PARTIAL),
                       ("partial_onered", _capi.FLAG_REORTH_PARTIAL | _capi.FLAG_ONE_REDUCE)):
        h = _capi.Handle(0)
        h.set_options(flags)
        if kappa:
            h.set_tuning(_capi.TUNE_PARTIAL_LOOKAHEAD, kappa)
        h.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
        a, b = h.run(n, v0)
        V = h.get_basis()
        th = np.linalg.eigvalsh(np.diag(a) + np.diag(b, 1) + np.diag(b, -1))
        if ref is None:
            ref = (a, b, th)
        scale = max(np.abs(ref[0]).max(), np.abs(ref[1]).max())
        out[tag] = dict(engine=h.last_engine(), sweeps=h.last_sweeps(), misses=h.last_sweep_misses(), syncs=h.last_host_syncs(),
                        da=float(np.abs(a - ref[0]).max() / scale), db=float(np.abs(b - ref[1]).max() / scale),
                        dth=float(np.abs(th - ref[2]).max() / np.abs(ref[2]).max()),
                        orth=float(np.abs(V @ V.T - np.eye(n)).max()) if M <= 200000 else None, breakdown=bool(h.breakdown))
        h.close()
    print(json.dumps(out))
