"""Randomised run of the one-collective partial loop (engine 8) over awkward shapes - stencils whose row count is no multiple of any
block size, ragged CSR, dense, n from 2 to 80 - against the full sweep: engine, look-ahead misses, sweep log vs the host replay
(oracle/partial_gates.py), coefficient and Ritz-value differences.   python tests/stress_partial_onered.py SEED TRIALS"""
import os
import sys

import numpy as np
import scipy.sparse

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanczos_amd import _capi, synthetic  # noqa: E402
from oracle import partial_gates as pg  # noqa: E402  (test infrastructure: the oracle is the checker here)

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rng = np.random.default_rng(seed)
bad = 0
for trial in range(trials):
    kind = ("lap2d", "lap3d", "ragged", "dense")[trial % 4]
    if kind == "lap2d":
        A = synthetic.laplacian_2d_5pt(int(rng.integers(5, 90)), int(rng.integers(5, 90)))
        H, M, setm = A.to_scipy(), A.shape[0], lambda h, A=A: h.set_csr(A.shape[0], 0, A.rowptr, A.colidx, A.vals)
    elif kind == "lap3d":
        A = synthetic.laplacian_3d_7pt(int(rng.integers(3, 20)), int(rng.integers(3, 20)), int(rng.integers(3, 20)))
        H, M, setm = A.to_scipy(), A.shape[0], lambda h, A=A: h.set_csr(A.shape[0], 0, A.rowptr, A.colidx, A.vals)
    elif kind == "ragged":
        M = int(rng.integers(40, 3000))
        R = scipy.sparse.random(M, M, density=min(0.3, 8.0 / M), random_state=rng, format="csr")
        H = (R + R.T + scipy.sparse.diags(rng.standard_normal(M))).tocsr()
        H.sort_indices()
        setm = lambda h, S=H, M=M: h.set_csr(M, 0, S.indptr, S.indices, S.data)
    else:
        M = int(rng.integers(20, 400))
        D = synthetic.dense_symmetric(M, seed=trial)
        H = D
        setm = lambda h, D=D: h.set_dense(D)
    n = int(min(M, rng.integers(2, 81)))
    v0 = rng.uniform(-1, 1, M)
    v0 /= np.linalg.norm(v0)
    res = {}
    for tag, flags in (("full", _capi.FLAG_FUSED_NORM), ("onered", _capi.FLAG_REORTH_PARTIAL | _capi.FLAG_ONE_REDUCE)):
        h = _capi.Handle(0)
        h.set_options(flags)
        setm(h)
        a, b = h.run(n, v0)
        res[tag] = (a, b, h.last_engine(), h.last_sweeps(), h.last_sweep_misses(), h.breakdown, h.last_sweep_log() if tag == "onered" and h.last_engine() == "partial-one-reduce" else None, h.get_basis())
        h.close()
    (af, bf, ef, sf, _, bdf, _, Vf), (a, b, e, s, mi, bd, log, V) = res["full"], res["onered"]
    scale = max(np.abs(af).max(), np.abs(bf).max())
    thf, th = np.linalg.eigvalsh(np.diag(af) + np.diag(bf, 1) + np.diag(bf, -1)), np.linalg.eigvalsh(np.diag(a) + np.diag(b, 1) + np.diag(b, -1))
    replay = None
    if log is not None:
        gates, m2 = pg.lookahead_gates([float(x) for x in a], [pg.warmup_norm(H, v0)] + [float(x) for x in b])
        replay = list(log) == [bool(x) for x in gates] and m2 == mi
    orth = float(np.abs(V @ V.T - np.eye(n)).max())
    ok = np.isfinite(a).all() and np.isfinite(b).all() and e in ("partial-one-reduce", "one-reduce-repeated") and (replay is not False)
    bad += not ok
    print(f"{trial:3d} {kind:6s} M={M:5d} n={n:3d} engine={e:19s} sweeps={s:3d} misses={mi} breakdown={bd}/{bdf} replay={replay} "
          f"d_alpha={np.abs(a - af).max() / scale:.1e} d_ritz={np.abs(th - thf).max() / np.abs(thf).max():.1e} orth={orth:.1e} {'ok' if ok else 'FAIL'}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
