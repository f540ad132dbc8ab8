"""Synthetic symmetric test/benchmark matrices.

All generators emit fp64 values with int32, column-sorted CSR structure - what
SciPy produces for the reference's builders (Python/Regular/Hamiltonian.py:62-68
followed by ``H.sort_indices()`` in 3Ddeuteron.py:81).  Grid conventions follow
Hamiltonian.py:73-99: periodic wrap, flat index ``x + Nx*y (+ Nx*Ny*z)``.

The structure (rowptr / colidx / degrees) is integer bookkeeping and is tested
bit-exact against a SciPy COO->CSR construction in tests/test_synthetic.py.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

__all__ = [
    "CSR",
    "laplacian_2d_5pt",
    "laplacian_3d_7pt",
    "random_graph_laplacian",
    "dense_symmetric",
    "dense_symmetric_hashed",
    "deuteron_potential",
    "DeuteronPotential",
    "reference_start_vector",
]


@dataclass
class CSR:
    """Minimal CSR holder (host side).  ``shape`` is (M, M)."""

    rowptr: np.ndarray  # int32 (M+1)
    colidx: np.ndarray  # int32 (nnz), sorted within each row
    vals: np.ndarray  # float64 (nnz)
    shape: tuple

    @property
    def nnz(self):
        return int(self.rowptr[-1])

    def to_scipy(self):
        import scipy.sparse

        return scipy.sparse.csr_matrix((self.vals, self.colidx, self.rowptr), shape=self.shape)

    def row_slice(self, lo, hi):
        """Rows [lo, hi) with GLOBAL column indices (what one rank owns)."""
        a, b = int(self.rowptr[lo]), int(self.rowptr[hi])
        return CSR((self.rowptr[lo : hi + 1] - self.rowptr[lo]).astype(np.int32), self.colidx[a:b], self.vals[a:b], (hi - lo, self.shape[1]))


def _stencil(dims, rows=None):
    """-Laplacian with periodic wrap on a grid of ``dims`` (fastest axis first):
    diagonal 2*d, each of the 2*d neighbours -1.  ``rows=(lo, hi)`` builds only
    that row block (global column indices)."""
    dims = tuple(int(d) for d in dims)
    if min(dims) < 3:
        raise ValueError("each grid dimension must be >= 3 (wrap neighbours would coincide)")
    M = int(np.prod(dims))
    if M * (2 * len(dims) + 1) >= 2**31:
        raise ValueError("nnz does not fit int32 CSR indices")
    lo, hi = (0, M) if rows is None else rows
    r = np.arange(lo, hi, dtype=np.int64)
    k = 2 * len(dims) + 1
    cols = np.empty((hi - lo, k), dtype=np.int32)
    cols[:, 0] = r
    stride = 1
    c = 1
    for d in dims:
        coord = (r // stride) % d
        cols[:, c] = r + np.where(coord == d - 1, -(d - 1) * stride, stride)
        cols[:, c + 1] = r + np.where(coord == 0, (d - 1) * stride, -stride)
        c += 2
        stride *= d
    cols.sort(axis=1)  # in place; the diagonal is wherever the row's own index lands
    vals = np.where(cols == r[:, None].astype(np.int32), float(2 * len(dims)), -1.0)
    rowptr = (np.arange(hi - lo + 1, dtype=np.int64) * k).astype(np.int32)
    return CSR(rowptr, np.ascontiguousarray(cols).reshape(-1), np.ascontiguousarray(vals).reshape(-1), (hi - lo, M))


def laplacian_2d_5pt(nx, ny, rows=None):
    """2-D periodic 5-point ``4I - shifts`` (PSD, row sums 0), M = nx*ny."""
    return _stencil((nx, ny), rows)


def laplacian_3d_7pt(nx, ny, nz, rows=None):
    """3-D periodic 7-point ``6I - shifts``, M = nx*ny*nz (cf. Hamiltonian.py:87-99, sign flipped)."""
    return _stencil((nx, ny, nz), rows)


def random_graph_laplacian(M, n_edges, seed=1234):
    """Irregular-graph Laplacian ``D - Adj`` of a random simple graph.

    ``n_edges`` undirected edges are drawn ``default_rng(seed).integers(0, M, (E, 2))``;
    self loops are dropped and duplicate edges merged.
    Values are integer-valued fp64.
    """
    M = int(M)
    rng = np.random.default_rng(seed)
    e = rng.integers(0, M, size=(int(n_edges), 2), dtype=np.int64)
    e = e[e[:, 0] != e[:, 1]]
    key = np.unique(np.minimum(e[:, 0], e[:, 1]) * M + np.maximum(e[:, 0], e[:, 1]))
    u, v = key // M, key % M
    del e, key
    deg = np.bincount(u, minlength=M) + np.bincount(v, minlength=M)
    diag = np.flatnonzero(deg)  # isolated vertices get an empty row (no explicit zero), like SciPy's D - Adj
    rows = np.concatenate([u, v, diag])
    cols = np.concatenate([v, u, diag])
    vals = np.concatenate([np.full(2 * len(u), -1.0), deg[diag].astype(np.float64)])
    order = np.argsort(rows * M + cols, kind="stable")
    nnz = len(order)
    if nnz >= 2**31:
        raise ValueError("nnz does not fit int32 CSR indices")
    rowptr = np.zeros(M + 1, dtype=np.int64)
    np.cumsum(deg + (deg > 0), out=rowptr[1:])
    return CSR(rowptr.astype(np.int32), cols[order].astype(np.int32), vals[order], (M, M))


def dense_symmetric(M, seed=0):
    """Config C1: ``A = standard_normal((M, M)); (A + A.T) / 2`` with ``default_rng(seed)``."""
    A = np.random.default_rng(seed).standard_normal((M, M))
    return (A + A.T) / 2


def dense_symmetric_hashed(M, rows=None, seed=0):
    """Rows ``[lo, hi)`` of a dense symmetric ``M x M`` matrix whose entry (i, j) is a counter-based hash of
    (min(i, j), max(i, j), seed) mapped to [-1, 1): every rank of a row partition generates exactly its block, no rank
    ever holds the whole matrix (the large synthetic dense workloads of bench.py; C1 keeps ``dense_symmetric``)."""
    lo, hi = (0, M) if rows is None else rows
    out = np.empty((hi - lo, M))
    j = np.arange(M, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        for r0 in range(lo, hi, 2048):
            r1 = min(hi, r0 + 2048)
            i = np.arange(r0, r1, dtype=np.uint64)[:, None]
            a, b = np.minimum(i, j), np.maximum(i, j)
            h = (a * np.uint64(0x9E3779B97F4A7C15)) ^ (b * np.uint64(0xC2B2AE3D27D4EB4F)) ^ np.uint64(seed * 2 + 1)
            h ^= h >> np.uint64(29)
            h *= np.uint64(0xBF58476D1CE4E5B9)
            h ^= h >> np.uint64(32)
            out[r0 - lo : r1 - lo] = (h >> np.uint64(11)).astype(np.float64) * (2.0 / 2**53) - 1.0
    return out


def deuteron_potential(x, y, z):
    """The potential of the reference's deuteron drivers (constants of 3Ddeuteron.py:51-61): hard core + well."""
    r = np.sqrt(x**2 + y**2 + z**2)
    eWell = 54.531
    return 40.0 * eWell * np.exp(-((r / 0.25) ** 4.0)) - 65.4823128982115 * np.exp(-((r / 1.7) ** 4.0))


class DeuteronPotential:
    """The same hard core + well as a callable with its constants exposed (``eCores exp(-(r/rCore)^fPow) - eWells
    exp(-(r/rWell)^fPow)``, 3Ddeuteron.py:51-61): ``lanczos_amd.Hamiltonian`` can then evaluate it on the device instead
    of point by point on the host (``Hamiltonian.device_potential = True``)."""

    def __init__(self, eCores=40.0 * 54.531, rCore=0.25, eWells=65.4823128982115, rWell=1.7, fPow=4.0):
        self.eCores, self.rCore, self.eWells, self.rWell, self.fPow = float(eCores), float(rCore), float(eWells), float(rWell), float(fPow)

    def __call__(self, x, y, z):
        r = np.sqrt(x**2 + y**2 + z**2)
        return self.eCores * np.exp(-((r / self.rCore) ** self.fPow)) - self.eWells * np.exp(-((r / self.rWell) ** self.fPow))

    def device_params(self, L):
        """the 8 parameters lz_build_stencil3d_block takes for potential_kind 2 on a cubic box of side L"""
        return np.array([self.eCores, self.rCore, self.eWells, self.rWell, self.fPow, L, L, L], dtype=np.float64)


def reference_start_vector(M, seed=99):
    """The reference's default start vector before normalisation
    (Python/Regular/Lanczos.py:93-97): global legacy RNG, uniform(-1, 1)."""
    return np.random.RandomState(seed).uniform(-1, 1, size=M)
