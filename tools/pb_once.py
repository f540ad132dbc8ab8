#!/usr/bin/env python3
"""Three two-phase SpMVs on the C3 matrix (for rocprofv3 counter passes)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanczos_amd import _capi, synthetic  # noqa: E402

A = synthetic.random_graph_laplacian(10_000_000, 35_000_000, seed=1234)
M = A.shape[0]
h = _capi.Handle(0)
h.set_tuning(_capi.TUNE_SPMV_PLAN, 2)
h.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
assert h.spmv_plan() == "two-phase"
h.basis_alloc(2)
h.basis_set_row(1, np.random.default_rng(0).standard_normal(M))
for _ in range(3):
    h.step_spmv(1)
h.close()
