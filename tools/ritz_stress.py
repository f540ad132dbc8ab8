"""Randomised check of the Ritz back-transform kernels (GPU box): random row counts (ragged tiles, waves without a whole tile,
the row-ragged tile owned by any wave), every n in 2..210, resident and chunked with random chunk sizes - against NumPy.
    python tools/ritz_stress.py SEED TRIALS"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanczos_amd import _capi  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)
bad = 0
for t in range(trials):
    M = int(rng.choice([4096, 4097, 4127, 5000, 8191, 8193, 16383, 20001, 65536, 65537, 70001, 99999, 131071]) if rng.random() < 0.6
            else rng.integers(4096, 140000))
    n = int(rng.integers(2, 211))
    # a diagonal operator with distinct entries: any start vector gives a full Krylov space, and the run is cheap
    d = np.linspace(1.0, 2.0, M) + 1e-3 * rng.standard_normal(M)
    ptr = np.arange(M + 1, dtype=np.int32)
    h = _capi.Handle(0)
    h.set_options(_capi.FLAG_FUSED_NORM)
    h.set_csr(M, 0, ptr, ptr[:-1], d)
    v0 = rng.standard_normal(M)
    v0 /= np.linalg.norm(v0)
    a, b = h.run(n, v0)
    V = h.get_basis()
    S = np.linalg.eigh(np.diag(a) + np.diag(b, 1) + np.diag(b, -1))[1]
    ref = V.T @ S
    Y = h.ritz_vectors(S)
    e1 = float(np.abs(Y - ref).max())
    G = h.ritz_gram()
    chunk = int(rng.integers(1, 40)) * 16 * int(rng.choice([1, 7, 64]))
    h.set_tuning(_capi.TUNE_RITZ_CHUNK_ROWS, chunk)
    h.ritz_vectors(S, fetch=False)
    lo = int(rng.integers(0, M - 1))
    hi = int(min(M, lo + rng.integers(1, 3000)))
    e2 = float(np.abs(h.ritz_fetch_rows(lo, hi) - ref[lo:hi]).max())
    e3 = float(np.abs(h.ritz_fetch() - ref).max())
    e4 = float(np.abs(h.ritz_gram() - G).max())
    ok = max(e1, e2, e3) < 1e-13 and e4 < 1e-12
    bad += not ok
    print(f"{t:3d} M={M:6d} n={n:3d} chunk={chunk:6d} resident {e1:.1e} rows {e2:.1e} chunked {e3:.1e} gram {e4:.1e} {'ok' if ok else 'FAIL'}", flush=True)
    h.close()
print("failures:", bad)
sys.exit(1 if bad else 0)
