"""CPU oracle for the reference's two-sided (bi-orthogonal) Lanczos variant.

TEST INFRASTRUCTURE ONLY (same rules as oracle/lanczos_ref.py: imported by
``tests/`` and ``oracle/gen_golden.py`` only, never by ``lanczos_amd/``).

Parity status: PINNED.  ``oracle/gen_golden.py`` runs the reference's own
``IrrLanczos.execute_Lanczos(..., use_cuda=False)`` here and this restatement
reproduces its ``H_eff`` and ``V`` bit for bit on the ``two_sided_*`` fixtures
under ``tests/golden/`` (field ``ref_vs_oracle_maxabs``).

What it restates (paths relative to /root/reference):

* ``execute_two_sided``   <- Python/Irregular/IrrLanczos.py:77-187 (CPU branch)
* ``bireorthogonalize``   <- Python/Irregular/IrrLanczos.py:390-443 (default
                             ``mem_safe=False`` branch, the only one the driver loop uses)
* ``bireorthogonalize_mem_safe`` <- Python/Irregular/IrrLanczos.py:398-407 (pinned by
                             tests/golden/bireorth_mem_safe.npz, made by the reference's static method)
* ``build_h_eff``         <- Python/Irregular/IrrLanczos.py:165-174

Semantics kept as they are in the reference, quirks included:

* only ``v0=None`` works (the second start vector ``v1`` exists only on that
  branch, :98-102): two consecutive draws from the legacy global RNG;
* start vectors are scaled so that ``q0 . p0 = +-1`` (:104-106);
* ``gamma[j-1]`` / ``beta[j-1]`` / ``q[j-1]`` at ``j = 0`` are Python negative
  indices: the still-zero last entries / last row, i.e. a no-op (:128-129);
* every new pair is first projected (modified Gram-Schmidt, sequential) on the
  ORTHONORMALISED copies of the other side's basis, rescaled so ``q.p = +-1``,
  and the orthonormal copies are extended by one vector each (:408-441);
* ``alpha[n-1] = q[n-1] . r`` with the LAST loop iteration's residual (:163);
* ``H_eff`` is the non-symmetric tridiagonal with sub-diagonal ``beta`` and
  super-diagonal ``gamma`` where row ``i >= 1`` holds ``gamma[i-1]`` (not
  ``gamma[i]``) to the right of the diagonal (:165-174); ``get_H_eigs`` then calls
  ``eigh`` on it, which reads the lower triangle only (:292).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse

__all__ = ["start_pair", "bireorthogonalize", "bireorthogonalize_mem_safe", "execute_two_sided", "build_h_eff"]


def start_pair(M, seed=99):
    """IrrLanczos.py:96-106."""
    np.random.seed(seed)
    v0 = np.random.uniform(-1, 1, size=(M))
    v1 = np.random.uniform(-1, 1, size=(M))
    dot = np.sqrt(np.abs(np.dot(v0, v1)))
    v0 = v0 / dot
    v1 = v1 / dot * np.sign(np.dot(v0, v1))
    return v0, v1


def bireorthogonalize(Q, P, Qb, Pb, j):
    """IrrLanczos.py:408-441, in place on row ``j`` of the four (n, M) arrays."""
    for i in range(j):
        Q[j] = Q[j] - np.dot(Q[j], Pb[i]) / np.dot(Pb[i], Pb[i]) * Pb[i]
        P[j] = P[j] - np.dot(P[j], Qb[i]) / np.dot(Qb[i], Qb[i]) * Qb[i]
    scale = np.sqrt(np.abs(np.dot(Q[j], P[j])))
    Q[j] = Q[j] / scale
    P[j] = P[j] / scale * np.sign(np.dot(Q[j], P[j]))
    Qb[j] = Q[j] / np.linalg.norm(Q[j])
    Pb[j] = P[j] / np.linalg.norm(P[j])
    for i in range(j):
        Qb[j] = Qb[j] - np.dot(Qb[j], Qb[i]) / np.dot(Qb[i], Qb[i]) * Qb[i]
        Pb[j] = Pb[j] - np.dot(Pb[j], Pb[i]) / np.dot(Pb[i], Pb[i]) * Pb[i]
    Qb[j] = Qb[j] / np.linalg.norm(Qb[j])
    Pb[j] = Pb[j] / np.linalg.norm(Pb[j])


def bireorthogonalize_mem_safe(V1, V2, j):
    """IrrLanczos.py:398-407 (``mem_safe=True``; the reference never calls it), in place on row ``j`` of the two (n, M)
    arrays: one sweep over ALL n rows, rows at and past ``j`` with unit divisor, row ``j`` itself skipped."""
    for X, B in ((V1, V2), (V2, V1)):
        uv = np.sum(X[j] * B, axis=1)
        uu = np.sum(B * B, axis=1)
        uu[j:] = 1
        uv[j] = 0
        X[j] = X[j] - np.sum((uv / uu)[:, None] * B, axis=0)


def build_h_eff(alpha, beta, gamma):
    """IrrLanczos.py:165-174."""
    n = len(alpha)
    T = np.zeros((n, n))
    T[0, 0] = alpha[0]
    T[0, 1] = gamma[0]
    T[-1, -2] = beta[-1]
    T[-1, -1] = alpha[-1]
    for i in range(1, n - 1):
        T[i, i - 1] = beta[i - 1]
        T[i, i] = alpha[i]
        T[i, i + 1] = gamma[i - 1]
    return T


def execute_two_sided(H, n, seed=99, start=None, return_all=False):
    """IrrLanczos.py:77-187.  Returns ``(alpha, beta, gamma, Q)`` with ``Q`` the (n, M) right basis
    (the reference publishes ``Q.T`` as ``V``); ``return_all`` adds ``P, Qb, Pb``.
    ``start=(q0, p0)`` bypasses the RNG (used by the tests to perturb the start pair)."""
    H = scipy.sparse.csr_matrix(H, dtype=np.float64)
    HT = scipy.sparse.csr_matrix(H.transpose(), dtype=np.float64)
    M = H.shape[0]
    if n > M:
        raise ValueError("n cannot be larger than M!")
    q0, p0 = start_pair(M, seed) if start is None else (np.array(start[0]), np.array(start[1]))
    Q = np.zeros((n, M))
    P = np.zeros((n, M))
    Q[0], P[0] = q0, p0
    Qb, Pb = Q.copy(), P.copy()
    Qb[0] = Qb[0] / np.linalg.norm(Qb[0])
    Pb[0] = Pb[0] / np.linalg.norm(Pb[0])
    alpha = np.zeros(n)
    beta = np.zeros(n - 1)
    gamma = np.zeros(n - 1)
    for j in range(n - 1):
        r = H * Q[j]
        s = HT * P[j]
        r = r - gamma[j - 1] * Q[j - 1]
        s = s - beta[j - 1] * P[j - 1]
        alpha[j] = (np.dot(P[j], r) + np.dot(Q[j], s)) / 2
        r = r - alpha[j] * Q[j]
        s = s - alpha[j] * P[j]
        w = np.dot(r, s)
        beta[j] = np.sqrt(np.abs(w))
        gamma[j] = w / beta[j]
        Q[j + 1] = r / beta[j]
        P[j + 1] = s / gamma[j]
        Qb[j + 1] = Q[j + 1]
        Pb[j + 1] = P[j + 1]
        bireorthogonalize(Q, P, Qb, Pb, j + 1)
    alpha[n - 1] = np.dot(Q[n - 1], r)
    if return_all:
        return alpha, beta, gamma, Q, P, Qb, Pb
    return alpha, beta, gamma, Q
