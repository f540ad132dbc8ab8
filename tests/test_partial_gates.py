"""The sweep DECISIONS of the opt-in partial re-orthogonalisation loops against `oracle/partial_gates.py`, a plain-Python restatement
of Simon's omega-recurrence as `k_omega` advances it (engine 7) and of the one-step look-ahead of `k_partial_onered_post` (engine 8: one
all-reduce per step).  The reference has only the full sweep (Lanczos.py:233-251), so this pins the build's own device logic: the device's
per-step record (`lz_last_sweep_log`) must equal the replay of the very coefficients the run delivered."""
import numpy as np
import pytest

from lanczos_amd import synthetic
from oracle import lanczos_ref as oracle
from oracle import partial_gates as pg


def test_replay_properties_on_the_cpu():
    """No device: the two restatements on the oracle's own coefficients - step 0 always sweeps (the reference's one-row sweep), sweeps
    come in Simon's pairs, the look-ahead triggers no later than the exact test and never misses, a huge safety factor sweeps (nearly)
    everything, kappa -> 0 leaves only what the exact row forces (misses, swept one step late)."""
    H = synthetic.laplacian_3d_7pt(10, 9, 8).to_scipy()
    n = 60
    a, b, _ = oracle.execute_lanczos(H, n, economy=True)
    v0 = oracle.start_vector(H.shape[0], 99, None)
    hb = [pg.warmup_norm(H, v0)] + list(b)
    g = pg.simon_gates(list(a), hb)
    gl, misses = pg.lookahead_gates(list(a), hb)
    assert g[0] and gl[0] and len(g) == len(gl) == n and misses == 0
    first = lambda x: next(i for i in range(1, n) if x[i])  # noqa: E731
    assert 1 < first(gl) <= first(g) < n - 1
    for gates in (g, gl):  # a swept vector (other than v_0) has a swept neighbour
        for j in range(1, n):
            if gates[j]:
                assert gates[j - 1] or (j + 1 < n and gates[j + 1]), j
    g_big, m_big = pg.lookahead_gates(list(a), hb, kappa=1e12)
    assert m_big == 0 and sum(g_big) > n // 2
    g_zero, m_zero = pg.lookahead_gates(list(a), hb, kappa=0.0)
    assert m_zero >= 1 and sum(g_zero) >= 1 + 2 * m_zero - 1


@pytest.mark.gpu
@pytest.mark.parametrize("build,n", [(lambda: synthetic.laplacian_3d_7pt(20, 18, 16), 150), (lambda: synthetic.laplacian_2d_5pt(96, 80), 60),
                                     (lambda: synthetic.random_graph_laplacian(50000, 175000, seed=3), 150),
                                     (lambda: synthetic.laplacian_3d_7pt(14, 12, 10), 120)])
def test_device_decisions_equal_the_replay(hip, build, n):
    A = build()
    H = A.to_scipy()
    M = A.shape[0]
    v0 = synthetic.reference_start_vector(M)
    v0 /= np.linalg.norm(v0)
    hb0 = pg.warmup_norm(H, v0)
    for flags, replay in ((hip.FLAG_REORTH_PARTIAL, lambda a, hb: (pg.simon_gates(a, hb), 0)),
                          (hip.FLAG_REORTH_PARTIAL | hip.FLAG_ONE_REDUCE, pg.lookahead_gates)):
        h = hip.Handle(0)
        h.set_options(flags)
        h.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
        a, b = h.run(n, v0)
        log, sweeps, misses = h.last_sweep_log(), h.last_sweeps(), h.last_sweep_misses()
        h.close()
        gates, m = replay([float(x) for x in a], [hb0] + [float(x) for x in b])
        assert log.sum() == sweeps and 1 <= sweeps < n
        assert list(log) == [bool(x) for x in gates], (flags, np.nonzero(log)[0], [i for i, x in enumerate(gates) if x])
        assert misses == m


@pytest.mark.gpu
def test_sweep_log_of_the_other_loops(hip):
    """all ones after a full-sweep loop; no record (LZ_ERR_ARG) after the host-decided loop of knob 18 = 1"""
    A = synthetic.laplacian_2d_5pt(40, 30)
    M = A.shape[0]
    v0 = synthetic.reference_start_vector(M)
    v0 /= np.linalg.norm(v0)
    h = hip.Handle(0)
    h.set_options(hip.FLAG_FUSED_NORM)
    h.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
    h.run(20, v0)
    assert h.last_sweep_log().all() and h.last_sweeps() == 20
    h.set_options(hip.FLAG_REORTH_PARTIAL)
    h.set_tuning(hip.TUNE_PARTIAL_LOOP, 1)
    h.run(20, v0)
    assert h.last_engine() == "kernels"
    with pytest.raises(hip.LanczosHipError):
        h.last_sweep_log()
    h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("build,n", [(lambda: synthetic.laplacian_3d_7pt(20, 18, 16), 120), (lambda: synthetic.laplacian_2d_5pt(96, 80), 40),
                                     (lambda: synthetic.random_graph_laplacian(6000, 20000, seed=4), 30)])
@pytest.mark.parametrize("partial", [False, True])
def test_one_collective_loops_against_their_restatement(hip, build, n, partial):
    """`oracle.execute_lanczos_one_reduce` restates what the one-collective loops compute (the two-column pass, the three-sum |r|^2, the
    subtraction order, the look-ahead gate): the device loops (engines 6 and 8) must deliver its coefficients to 1e-11 of the scale on
    these well-conditioned runs - and, for the partial loop, the same sweep schedule."""
    A = build()
    H = A.to_scipy()
    M = A.shape[0]
    v0 = synthetic.reference_start_vector(M)
    v0 /= np.linalg.norm(v0)
    ao, bo, Vo, gates, calls = oracle.execute_lanczos_one_reduce(H, n, [0, M], v0=v0, partial=partial)
    assert calls == n + 1
    h = hip.Handle(0)
    h.set_options(hip.FLAG_ONE_REDUCE | (hip.FLAG_REORTH_PARTIAL if partial else hip.FLAG_FUSED_NORM))
    h.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
    a, b = h.run(n, v0)
    assert h.last_engine() == ("partial-one-reduce" if partial else "one-reduce")
    scale = max(np.abs(ao).max(), np.abs(bo).max())
    assert np.abs(a - ao).max() <= 1e-11 * scale and np.abs(b - bo).max() <= 1e-11 * scale
    V = h.get_basis()
    assert np.abs(V[:8] - Vo[:8]).max() <= 1e-10 * np.abs(Vo[:8]).max()
    if partial:
        assert list(h.last_sweep_log()) == [bool(x) for x in gates] and h.last_sweep_misses() == 0
    h.close()


@pytest.mark.gpu
def test_one_collective_partial_loop_on_awkward_shapes():
    """`tests/stress_partial_onered.py`: engine 8 on 24 random shapes (stencils whose row count is no multiple of any block size, ragged
    CSR, dense, n = 2 .. 80, some with a nearly exhausted Krylov space) - finite coefficients, the loop itself or its guarded repeat, no
    look-ahead miss, the device's sweep log equal to the host replay, a semi-orthogonal basis."""
    import os
    import re
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "stress_partial_onered.py"), "7", "24"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    rows = [l for l in p.stdout.splitlines() if " engine=" in l]
    assert len(rows) == 24 and "failures: 0" in p.stdout
    for l in rows:
        assert "misses=0" in l and "replay=False" not in l, l
        assert float(re.search(r"orth=([0-9.e+-]+)", l).group(1)) < 1e-6, l
