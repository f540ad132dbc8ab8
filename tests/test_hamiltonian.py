"""The regular-grid Hamiltonian builder: the ``Hamiltonian`` mirror and the device assembly
kernel against matrices produced by the reference's own builder (tests/golden/hamiltonian_N6.npz and the N = 12
27-point H stored in deuteron3d_N12_27pt_n100.npz).  Bit-exact: structure AND values."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, load_golden
from lanczos_amd import Hamiltonian, Lanczos


def deuteron_potential(x, y, z):
    r = np.sqrt(x**2 + y**2 + z**2)
    eWell = 54.531
    return 40.0 * eWell * np.exp(-((r / 0.25) ** 4.0)) - 65.4823128982115 * np.exp(-((r / 1.7) ** 4.0))


@pytest.fixture()
def fix():
    return dict(np.load(os.path.join(GOLDEN_DIR, "hamiltonian_N6.npz")))


def make(N, tmp_path, monkeypatch, T_factor=None):
    monkeypatch.chdir(tmp_path)  # the builder creates ./T_matrices like the reference
    dx = 25.0 / N
    Tf = 197.327**2 / (2 * 469.4592) / dx**2 if T_factor is None else T_factor
    Hamiltonian.verbose = False
    return Hamiltonian(N, 25, deuteron_potential, Tf)


def test_host_side_matches_reference_builder(fix, tmp_path, monkeypatch):
    ham = make(6, tmp_path, monkeypatch, float(fix["T_factor"]))
    assert os.path.isdir("T_matrices")
    assert np.array_equal(ham.x, fix["x"]) and ham.dx == 25.0 / 6
    assert np.array_equal(ham.weights_27point, fix["weights_27point"]) and np.array_equal(ham.weights_7point, fix["weights_7point"])
    assert np.array_equal(ham.get_weights_27point(), fix["weights_27point"])
    assert list(ham.Laplacian_7point(0)[0]) == list(fix["neighbors_7_of_row_0"])
    assert list(ham.Laplacian_27point(215)[0]) == list(fix["neighbors_27_of_row_215"])
    assert ham.ravel_i(ham.unravel_xyz(3, 4, 5)) == (3, 4, 5)
    # point-by-point evaluation reproduces the reference's loop bit for bit; the one-call array form agrees to 1e-12
    assert np.array_equal(ham.potential_on_grid(), fix["V_diag"])
    np.testing.assert_allclose(ham.potential_on_grid(vectorized=True), fix["V_diag"], rtol=1e-12, atol=0)
    ham.create_sparse_V()
    assert np.array_equal(ham.V_sparse.diagonal(), fix["V_diag"]) and ham.V_sparse.nnz == 216


@pytest.mark.gpu
@pytest.mark.parametrize("pts", ["7", "27"])
def test_device_assembly_bit_exact(fix, tmp_path, monkeypatch, pts):
    ham = make(6, tmp_path, monkeypatch, float(fix["T_factor"]))
    ham.create_sparse_T(pts)
    T = ham.T_sparse
    assert T.has_sorted_indices and T.indices.dtype == np.int32
    assert np.array_equal(T.indptr, fix[f"T{pts}_rowptr"]) and np.array_equal(T.indices, fix[f"T{pts}_colidx"])
    assert np.array_equal(T.data, fix[f"T{pts}_vals"])
    # on-disk cache: same name and format as the reference (scipy.sparse.save_npz), and it is used on the next call
    f = f"T_matrices/T_N=6_Laplace={pts}.npz"
    assert os.path.isfile(f)
    import scipy.sparse

    assert (scipy.sparse.load_npz(f) != T).nnz == 0
    ham2 = make(6, tmp_path, monkeypatch, float(fix["T_factor"]))
    ham2._device_csr = None  # would raise if the cache were ignored
    ham2.create_sparse_T(pts)
    assert np.array_equal(ham2.T_sparse.data, T.data)
    H = ham.build_H(pts)
    assert np.array_equal(H.indptr, fix[f"H{pts}_rowptr"]) and np.array_equal(H.indices, fix[f"H{pts}_colidx"])
    assert np.array_equal(H.data, fix[f"H{pts}_vals"])


@pytest.mark.gpu
def test_mini_3ddeuteron_end_to_end(tmp_path, monkeypatch):
    """3Ddeuteron.py scaled to N = 12: builder on the device -> Lanczos on the device, against the reference's H and
    Ritz values (fixture iii)."""
    d, Href = load_golden("deuteron3d_N12_27pt_n100")
    ham = make(12, tmp_path, monkeypatch)
    H = ham.build_H("27")
    assert np.array_equal(H.indptr, Href.indptr) and np.array_equal(H.indices, Href.indices) and np.array_equal(H.data, Href.data)
    Lanczos.verbose = False
    s = Lanczos(H)
    s.execute_Lanczos(100, seed=78)
    low = s.H_eigvals[:4]
    np.testing.assert_allclose(low, d["H_eigvals"][:4], rtol=0, atol=1e-10 * np.abs(d["H_eigvals"]).max())


@pytest.mark.gpu
def test_device_assembly_large_properties(tmp_path, monkeypatch):
    """N = 96 (M = 884 736, 27-point: 2.4e7 entries): symmetric, sorted, row sums of T are zero to rounding."""
    ham = make(96, tmp_path, monkeypatch)
    h = ham
    T = h._device_csr("27", None, False)
    assert T.shape == (96**3, 96**3) and T.nnz == 27 * 96**3 and T.has_sorted_indices
    assert np.abs(np.asarray(T.sum(axis=1)).ravel()).max() < 1e-9 * abs(ham.T_factor)
    assert abs(T - T.T).max() == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("dims", [(9, 8, 7), (5, 12, 6)])
def test_noncubic_stencil_assembly_is_bit_exact(dims):
    """lz_build_stencil3d_block on a Nx x Ny x Nz grid (whole matrix on one rank) against the NumPy generator of the
    BASELINE C4 matrix (synthetic.laplacian_3d_7pt: 6 on the diagonal, -1 on the six periodic neighbours)."""
    from lanczos_amd import _capi, synthetic

    A = synthetic.laplacian_3d_7pt(*dims)
    h = _capi.Handle(0)
    h.build_stencil3d_block(dims, 7, 1.0, [-6.0, 1.0, 0.0, 0.0], 0, A.shape[0], (), negate_T=True)
    rowptr, colidx, vals = h.get_csr()
    assert np.array_equal(rowptr, A.rowptr) and np.array_equal(colidx, A.colidx) and np.array_equal(vals, A.vals)
    assert h.spmv_plan() == "fixed-k"
    h.close()


@pytest.mark.gpu
def test_device_potential_matches_the_host_evaluation(tmp_path, monkeypatch):
    """Hamiltonian.device_potential: the deuteron hard core + well evaluated inside the assembly kernel (no N^3 host
    array) against the reference-faithful host evaluation - same structure, values within the libm difference."""
    from lanczos_amd import Hamiltonian, synthetic

    monkeypatch.chdir(tmp_path)
    Hamiltonian.verbose = False
    N, L = 14, 25
    pot = synthetic.DeuteronPotential()
    x = np.linspace(-3, 3, 7)
    assert np.array_equal(pot(x, x[::-1], 0.3 * x), synthetic.deuteron_potential(x, x[::-1], 0.3 * x))
    ham = Hamiltonian(N, L, pot, 197.327**2 / (2 * 469.4592) / (float(L) / N) ** 2)
    H_host = ham.build_H("27")
    ham.device_potential = True
    H_dev = ham.build_H("27")
    assert np.array_equal(H_host.indptr, H_dev.indptr) and np.array_equal(H_host.indices, H_dev.indices)
    off = H_host.indices != np.repeat(np.arange(N**3), 27)
    assert np.array_equal(H_host.data[off], H_dev.data[off])  # the kinetic part is the same arithmetic
    np.testing.assert_allclose(H_dev.diagonal(), H_host.diagonal(), rtol=1e-12, atol=1e-12 * np.abs(H_host.diagonal()).max())
