"""The CPU oracle against the golden vectors produced by the reference itself
(oracle/gen_golden.py ran /root/reference's use_cuda=False path).  Runs anywhere."""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from oracle import lanczos_ref as oracle


@pytest.mark.parametrize("name", golden_names())
def test_oracle_reproduces_reference(name):
    d, H = load_golden(name)
    n, seed = int(d["n"]), int(d["seed"])
    v0 = d["v0"] if "v0" in d else None
    assert float(d["ref_vs_oracle_maxabs"]) == 0.0  # recorded when the fixture was generated next to the reference
    a, b, V = oracle.execute_lanczos(H, n, seed=seed, v0=v0, economy=(n > 200))
    same_stack = str(d["numpy_version"]) == np.__version__
    prefix, _ = oracle.stable_masks(H, n, d["alpha"], d["beta"], seed=seed, v0=v0) if not same_stack else (n, None)
    if same_stack:
        assert np.array_equal(a, d["alpha"]) and np.array_equal(b, d["beta"])
    else:  # another BLAS may round the dots differently
        np.testing.assert_allclose(a[:prefix], d["alpha"][:prefix], rtol=0, atol=1e-10 * np.abs(d["alpha"]).max())
    theta, S, Y = oracle.ritz_pairs(oracle.build_h_eff(a, b), V)
    if same_stack:
        assert np.array_equal(theta, d["H_eigvals"])
        if "V" in d:
            assert np.array_equal(V, d["V"])
            assert np.array_equal(Y[:, :3], d["H_eigvecs_first3"])
        assert oracle.is_normalized(Y) == float(d["norm_closest_to_1"])
        assert oracle.is_orthogonal(Y) == pytest.approx(float(d["max_offdiag_gram"]), rel=1e-6, abs=1e-12)


@pytest.mark.parametrize("name", ["lap2d_32x32_n30", "ragged_M700_n25", "graph_M2000_E7000_n40"])
def test_economy_sweep_is_bit_identical(name):
    d, H = load_golden(name)
    n, seed = int(d["n"]), int(d["seed"])
    full = oracle.execute_lanczos(H, n, seed=seed)
    eco = oracle.execute_lanczos(H, n, seed=seed, economy=True)
    for x, y in zip(full, eco):
        assert np.array_equal(x, y)


def test_dense_input_is_wrapped_as_csr():
    d, H = load_golden("box1d_N500_n50")
    a, b, _ = oracle.execute_lanczos(H.toarray(), 50)
    assert np.array_equal(a, d["alpha"]) or np.allclose(a, d["alpha"], rtol=0, atol=1e-12)


def test_reference_quirks():
    d, H = load_golden("lap2d_32x32_n30")
    a, b, V = oracle.execute_lanczos(H, 30)
    v0 = oracle.start_vector(1024, 99)
    # the warm-up step is overwritten: V[0] is normalise((H - alpha0 I) v0), orthogonal to v0
    assert abs(V[0] @ v0) < 1e-14
    w = H * v0 - (v0 @ (H * v0)) * v0
    np.testing.assert_allclose(V[0], w / np.linalg.norm(w), rtol=0, atol=1e-14)
    with pytest.raises(ValueError, match="n cannot be larger than M!"):
        oracle.execute_lanczos(H, 1025)
    with pytest.raises(IndexError):
        oracle.execute_lanczos(H, 1)  # beta = zeros(0); beta[-1] = ...


def test_virtual_ranks_match_single_rank():
    d, H = load_golden("lap2d_32x32_n30")
    a, b, V = oracle.execute_lanczos(H, 30)
    for bounds in ([0, 512, 1024], [0, 100, 333, 1024], [0, 256, 512, 768, 1024]):
        a2, b2, V2 = oracle.execute_lanczos_partitioned(H, 30, bounds)
        np.testing.assert_allclose(a2, a, rtol=0, atol=1e-12)
        np.testing.assert_allclose(b2, b, rtol=0, atol=1e-12)
        np.testing.assert_allclose(V2[:10], V[:10], rtol=0, atol=1e-10)
