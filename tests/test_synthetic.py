"""Integer bookkeeping of the synthetic generators: bit-exact against a SciPy COO->CSR build."""
import numpy as np
import pytest
import scipy.sparse

from lanczos_amd import synthetic


def scipy_stencil(dims):
    dims = tuple(dims)
    M = int(np.prod(dims))
    idx = np.arange(M).reshape(dims[::-1])  # slowest axis first -> flat index x + Nx*y (+ Nx*Ny*z)
    rows, cols, vals = [np.arange(M)], [np.arange(M)], [np.full(M, 2.0 * len(dims))]
    for ax in range(len(dims)):
        for sh in (1, -1):
            rows.append(np.arange(M))
            cols.append(np.roll(idx, sh, axis=ax).reshape(-1))
            vals.append(np.full(M, -1.0))
    A = scipy.sparse.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(M, M)).tocsr()
    A.sort_indices()
    return A


@pytest.mark.parametrize("dims", [(7, 5), (3, 3), (16, 9), (5, 4, 3), (6, 6, 6)])
def test_stencils_bit_exact(dims):
    A = synthetic.laplacian_2d_5pt(*dims) if len(dims) == 2 else synthetic.laplacian_3d_7pt(*dims)
    R = scipy_stencil(dims)
    assert A.rowptr.dtype == np.int32 and A.colidx.dtype == np.int32 and A.vals.dtype == np.float64
    assert np.array_equal(A.rowptr, R.indptr) and np.array_equal(A.colidx, R.indices) and np.array_equal(A.vals, R.data)
    S = A.to_scipy()
    assert (S != S.T).nnz == 0
    assert np.array_equal(np.asarray(S.sum(axis=1)).ravel(), np.zeros(S.shape[0]))


def test_stencil_row_slices_concatenate():
    full = synthetic.laplacian_2d_5pt(40, 30)
    parts = [synthetic.laplacian_2d_5pt(40, 30, rows=(lo, hi)) for lo, hi in [(0, 400), (400, 401), (401, 1200)]]
    assert np.array_equal(np.concatenate([p.colidx for p in parts]), full.colidx)
    assert np.array_equal(np.concatenate([p.vals for p in parts]), full.vals)
    assert np.array_equal(full.row_slice(400, 1200).rowptr, np.arange(801) * 5)


def test_small_grids_rejected():
    with pytest.raises(ValueError):
        synthetic.laplacian_2d_5pt(2, 8)


@pytest.mark.parametrize("M,E", [(1000, 3500), (100000, 350000)])
def test_random_graph_laplacian_bit_exact(M, E):
    A = synthetic.random_graph_laplacian(M, E, seed=1234)
    e = np.random.default_rng(1234).integers(0, M, size=(E, 2), dtype=np.int64)
    e = e[e[:, 0] != e[:, 1]]
    adj = scipy.sparse.coo_matrix((np.ones(len(e)), (e[:, 0], e[:, 1])), shape=(M, M)).tocsr()
    adj = adj + adj.T
    adj.data[:] = 1.0  # duplicates merged: simple graph
    deg = np.asarray(adj.sum(axis=1)).ravel()
    L = (scipy.sparse.diags(deg) - adj).tocsr()
    L.sort_indices()
    assert np.array_equal(A.rowptr, L.indptr) and np.array_equal(A.colidx, L.indices) and np.array_equal(A.vals, L.data)
    # degree vector: the diagonal entries are exact integers
    assert np.array_equal(A.to_scipy().diagonal(), deg)
    assert abs(A.nnz / M - (2 * E / M + 1)) < 0.1


def test_dense_and_start_vector_conventions():
    A = synthetic.dense_symmetric(64, seed=0)
    B = np.random.default_rng(0).standard_normal((64, 64))
    assert np.array_equal(A, (B + B.T) / 2)
    np.random.seed(99)
    assert np.array_equal(synthetic.reference_start_vector(1000, 99), np.random.uniform(-1, 1, size=1000))


def test_dense_symmetric_hashed_is_symmetric_and_row_sliceable():
    A = synthetic.dense_symmetric_hashed(300, seed=2)
    assert A.shape == (300, 300) and np.array_equal(A, A.T)
    assert A.min() >= -1 and A.max() < 1 and abs(A.mean()) < 0.02 and 0.5 < A.std() < 0.65  # ~uniform(-1, 1)
    B = synthetic.dense_symmetric_hashed(300, rows=(117, 260), seed=2)
    assert np.array_equal(B, A[117:260])
    assert not np.array_equal(A, synthetic.dense_symmetric_hashed(300, seed=3))
