"""CPU: the two-sided oracle (oracle/two_sided_ref.py) against the golden vectors produced by running the reference's
IrrLanczos.execute_Lanczos (two-sided, IrrLanczos.py:77-187) in the build container (oracle/gen_golden.py)."""
import numpy as np
import pytest

from conftest import load_golden, two_sided_names
from oracle import two_sided_ref as ts


def test_fixtures_exist():
    assert len(two_sided_names()) >= 4


@pytest.mark.parametrize("name", two_sided_names())
def test_oracle_reproduces_reference_bit_for_bit(name):
    d, H = load_golden(name)
    n, seed = int(d["n"]), int(d["seed"])
    assert float(d["ref_vs_oracle_maxabs"]) == 0.0  # recorded when the fixture was generated next to the reference
    a, b, g, Q = ts.execute_two_sided(H, n, seed=seed)
    assert np.array_equal(a, d["alpha"]) and np.array_equal(b, d["beta"]) and np.array_equal(g, d["gamma"])
    assert np.array_equal(ts.build_h_eff(a, b, g), d["H_eff"])
    if "V" in d:
        assert np.array_equal(Q, d["V"])
    # get_H_eigs of the Irregular copy: eigh reads the lower triangle (beta) only
    assert np.array_equal(np.linalg.eigh(d["H_eff"])[0], d["H_eigvals"])
    q0, p0 = ts.start_pair(int(d["M"]), seed)
    assert np.array_equal(q0, d["q0"]) and np.array_equal(p0, d["p0"])
    assert abs(abs(q0 @ p0) - 1) < 1e-14


def test_biorthogonality_of_the_early_pairs():
    d, H = load_golden("two_sided_graph_M2000_n20")
    a, b, g, Q, P, Qb, Pb = ts.execute_two_sided(H, 8, seed=int(d["seed"]), return_all=True)
    G = Q @ P.T
    assert np.abs(np.abs(np.diag(G)) - 1).max() < 1e-12
    assert np.abs(G - np.diag(np.diag(G))).max() < 1e-8
    assert np.abs(Qb @ Qb.T - np.eye(8)).max() < 1e-12 and np.abs(Pb @ Pb.T - np.eye(8)).max() < 1e-12


def test_h_eff_layout_quirk():
    T = ts.build_h_eff(np.arange(1.0, 6.0), np.arange(10.0, 14.0), np.arange(20.0, 24.0))
    assert T[0, 1] == 20.0 and T[1, 2] == 20.0 and T[2, 3] == 21.0 and T[3, 4] == 22.0  # row i >= 1 carries gamma[i-1]
    assert T[1, 0] == 10.0 and T[4, 3] == 13.0 and T[4, 4] == 5.0


def test_mem_safe_branch_matches_the_reference_static_method():
    """IrrLanczos.py:398-407 as run by the reference itself (oracle/gen_golden.py --mem-safe-only): bit for bit."""
    import os

    from conftest import GOLDEN_DIR

    d = np.load(os.path.join(GOLDEN_DIR, "bireorth_mem_safe.npz"))
    assert float(d["ref_vs_oracle_maxabs"]) == 0.0
    for tag in "abc":
        V1, V2, j = d[tag + "_V1"].copy(), d[tag + "_V2"].copy(), int(d[tag + "_j"])
        ts.bireorthogonalize_mem_safe(V1, V2, j)
        assert np.array_equal(V1[j], d[tag + "_out1"]) and np.array_equal(V2[j], d[tag + "_out2"])
        assert np.array_equal(np.delete(V1, j, 0), np.delete(d[tag + "_V1"], j, 0))
