"""Randomised check of the Gram-matrix kernels (GPU box): random row counts, every column count in 33 .. 720 (single-unit kernels,
grouped kernels with every group size 7 .. 11, even n through LDS, odd n through the register rings), resident and chunked Y -
the symmetric kernel (knob 19 = 0) against the split-K TN GEMM (knob 19 = 1) and against its register-ring form (knob 19 = 2): bit
for bit when both run the same number of K slices (knob 21; the automatic count depends on the form's occupancy), else to rounding.
    python tools/gram_stress.py SEED TRIALS"""
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
    n = int(rng.integers(33, 721))
    M = int(rng.choice([4096, 4097, 5000, 8191, 8193, 12289, 20001, 32768, 40001]) if rng.random() < 0.6 else rng.integers(max(4096, n + 1), 50000))
    d = np.linspace(1.0, 2.0, M) + 1e-3 * rng.standard_normal(M)  # a diagonal operator with distinct entries: a full Krylov space, cheaply
    ptr = np.arange(M + 1, dtype=np.int32)
    v0 = rng.standard_normal(M)
    v0 /= np.linalg.norm(v0)
    S = rng.standard_normal((n, n))  # Y = V S NOT orthonormal: every entry of G is exercised
    chunk = int(rng.integers(1, 6)) * 4096 if rng.random() < 0.3 else 0
    slices = int(rng.choice([0, 3, 8, 21]))  # 0: automatic
    G = {}
    for knob in (0, 1, 2):
        h = _capi.Handle(0)
        h.set_options(_capi.FLAG_REORTH_PARTIAL)
        h.set_tuning(_capi.TUNE_GRAM_KERNEL, knob)
        h.set_tuning(_capi.TUNE_GRAM_SLICES, slices)
        if chunk:
            h.set_tuning(_capi.TUNE_RITZ_CHUNK_ROWS, chunk)
        h.set_csr(M, 0, ptr, ptr[:-1], d)
        h.run(n, v0)
        h.ritz_vectors(S, fetch=False)
        G[knob] = h.ritz_gram()
        h.close()
    scale = np.abs(G[1]).max()
    e01 = float(np.abs(G[0] - G[1]).max() / scale)
    same = bool(np.array_equal(G[0], G[2])) if slices else bool(np.abs(G[0] - G[2]).max() <= 1e-13 * scale)
    sym = bool(np.array_equal(G[0], G[0].T))
    ok = e01 < 1e-12 and same and sym
    bad += not ok
    print(f"{t:3d} M={M:6d} n={n:3d} chunk={chunk:6d} slices={slices:2d}  vs split-K {e01:.1e}  == register-ring form {same}  symmetric {sym}  {'ok' if ok else 'FAIL'}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
