"""CPU oracle: NumPy restatement of the reference's symmetric Lanczos path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``lanczos_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and only as the checker / the reported CPU baseline.

Parity status: PINNED.  ``oracle/gen_golden.py`` ran the reference's own
``use_cuda=False`` path (imported from /root/reference with empty ``cupy``
stubs) in the build container and this restatement reproduced its ``alpha``,
``beta``, ``H_eff``, ``V`` and Ritz values bit for bit on every fixture under
``tests/golden/`` (see the ``ref_vs_oracle_maxabs`` field stored in each
fixture and tests/test_oracle_golden.py).

What it restates (reference paths are relative to /root/reference):

* ``execute_lanczos``  <- Python/Regular/Lanczos.py:75-141
                          (== Python/Irregular/IrrLanczos.py:193-260 on CPU)
* ``reorthogonalize``  <- Python/Regular/Lanczos.py:233-251 (CPU branch :247-249)
* ``build_h_eff``      <- Python/Regular/Lanczos.py:121-130
* ``ritz_pairs``       <- Python/Regular/Lanczos.py:145-163
* ``is_normalized`` / ``is_orthogonal`` <- Python/Regular/Lanczos.py:288-323

The arithmetic is deliberately the reference's, including its quirks:
the warm-up step whose result is overwritten at j = 0, ``beta[j-1]`` landing in
``beta[-1]`` at j = 0, ``V[j-1]`` being the still-zero last row at j = 0, the
single-pass classical Gram-Schmidt written as ``2*V[j] - sum_i c_i V[i]`` with the
self term included, and no re-normalisation after it.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse

__all__ = [
    "as_operator",
    "start_vector",
    "reorthogonalize",
    "execute_lanczos",
    "build_h_eff",
    "ritz_pairs",
    "is_normalized",
    "is_orthogonal",
    "execute_lanczos_partitioned",
    "execute_lanczos_reordered",
    "stable_masks",
    "converged_ritz",
]


def as_operator(H):
    """Return ``H`` in a form whose ``H * v`` is a matvec.

    Reference Lanczos.py:108 uses ``H*V[0]``; for SciPy sparse input that is
    sparsetools' matvec.  A dense ndarray only works on the reference's GPU
    branch, where it is converted to CSR first (Lanczos.py:88); the oracle does
    the same conversion so dense inputs have a defined CPU answer.
    """
    if scipy.sparse.issparse(H):
        return H
    return scipy.sparse.csr_matrix(np.asarray(H, dtype=np.float64))


def start_vector(M, seed=99, v0=None):
    """Lanczos.py:93-100: seed the *global legacy* NumPy RNG, draw or copy v0, normalise."""
    np.random.seed(seed)
    if v0 is None:
        v0 = np.random.uniform(-1, 1, size=(M))
    else:
        v0 = np.array(v0)
    return v0 / np.linalg.norm(v0)


def reorthogonalize(V, j, rows=None):
    """Lanczos.py:247-249 (CPU branch), in place on row ``j``.

    ``rows=None`` sweeps all ``n`` rows exactly like the reference (two n x M
    temporaries).  ``rows=j+1`` restricts the sweep to the rows that can be
    non-zero; the result is identical bit for bit because the skipped rows are
    exactly zero (their coefficients are +0.0 and ``x + 0.0 == x``).
    """
    W = V if rows is None else V[:rows]
    inner_prods = np.sum(V[j] * W, axis=1)
    V[j] = 2 * V[j] - np.sum(inner_prods[:, None] * W, axis=0)
    return inner_prods


def execute_lanczos(H, n, seed=99, v0=None, economy=False, return_coeffs=False):
    """Lanczos.py:75-130.  Returns ``(alpha, beta, V)`` with ``V`` of shape (n, M),
    basis vector j being row j (the reference publishes ``V.T``, Lanczos.py:139).

    ``economy=True`` limits the re-orthogonalisation sweep to rows 0..j (same
    bits, half the work); the CPU-baseline timing in bench.py uses the faithful
    ``economy=False`` form.
    """
    H = as_operator(H)
    M = H.shape[0]
    if n > M:
        raise ValueError("n cannot be larger than M!")
    v0 = start_vector(M, seed, v0)

    V = np.zeros((n, M))
    V[0] = v0
    alpha = np.zeros(n)
    beta = np.zeros(n - 1)
    coeffs = []
    r = H * V[0]
    alpha[0] = np.dot(r, V[0])
    r = r - alpha[0] * V[0]
    for j in range(0, n):
        beta[j - 1] = np.linalg.norm(r)
        V[j] = r / beta[j - 1]
        c = reorthogonalize(V, j, rows=(j + 1) if economy else None)
        if return_coeffs:
            coeffs.append(c[: j + 1].copy())
        r = H * V[j]
        alpha[j] = np.dot(V[j], r)
        r = r - V[j] * alpha[j] - V[j - 1] * beta[j - 1]
    if return_coeffs:
        return alpha, beta, V, coeffs
    return alpha, beta, V


def build_h_eff(alpha, beta):
    """Lanczos.py:121-130: dense symmetric tridiagonal from alpha (n) and beta (n-1)."""
    n = len(alpha)
    H_eff = np.zeros((n, n))
    H_eff[0, 0] = alpha[0]
    H_eff[0, 1] = beta[0]
    H_eff[-1, -2] = beta[-1]
    H_eff[-1, -1] = alpha[-1]
    for i in range(1, n - 1):
        H_eff[i, i - 1] = beta[i - 1]
        H_eff[i, i] = alpha[i]
        H_eff[i, i + 1] = beta[i]
    return H_eff


def ritz_pairs(H_eff, V_rows):
    """Lanczos.py:151-156: ``eigh(H_eff)`` then ``Y[:, i] = V @ S[:, i]``.

    ``V_rows`` is the (n, M) row-vector basis; the reference multiplies with its
    transposed view, which is what ``V_rows.T`` is here.
    """
    theta, S = np.linalg.eigh(H_eff)
    Vt = V_rows.T
    M, n = Vt.shape
    Y = np.zeros((M, n))
    for i in range(n):
        Y[:, i] = np.dot(Vt, S[:, i])
    return theta, S, Y


def is_normalized(Y):
    """Lanczos.py:288-305 with no_assert=True: the column norm closest to 1."""
    norms = np.linalg.norm(Y, axis=0)
    return norms[np.argmin(np.abs(norms - 1))]


def is_orthogonal(Y):
    """Lanczos.py:307-323 with no_assert=True: sqrt of the largest off-diagonal
    |Gram| entry."""
    G = np.abs(Y.T @ Y - np.eye(Y.shape[1]) * np.linalg.norm(Y, axis=0) ** 2)
    return float(np.sqrt(G.max()))


# --------------------------------------------------------------------------
# "virtual ranks": the same recurrence with every vector row-block partitioned
# over P ranks and every inner product formed as an explicit sum of per-rank
# partials.  Used by the tests to check the partition / halo logic of the
# product's host code without a multi-GPU node.
# --------------------------------------------------------------------------
def execute_lanczos_partitioned(H, n, bounds, seed=99, v0=None, allreduce=None, spmv_local=None):
    """Row-partitioned restatement.

    ``bounds`` is the list of row offsets ``[0, ..., M]`` (P + 1 entries).  If
    ``allreduce``/``spmv_local`` are given the function runs as ONE rank of a
    real process group (``bounds`` then holds just this rank's ``[lo, hi]``):
    ``allreduce(np.ndarray) -> np.ndarray`` sums over ranks and
    ``spmv_local(x_local) -> y_local`` applies the distributed matvec.
    Otherwise all ranks are simulated in-process.
    Returns ``(alpha, beta, V)`` with V the local (n, rows) block or the
    concatenated global basis.
    """
    H = as_operator(H).tocsr()
    M = H.shape[0]
    if n > M:
        raise ValueError("n cannot be larger than M!")
    v0g = start_vector(M, seed, v0)

    if allreduce is not None:
        lo, hi = bounds
        parts = [(lo, hi)]
    else:
        parts = [(bounds[p], bounds[p + 1]) for p in range(len(bounds) - 1)]

        def allreduce(x):
            return x

    def gsum(partials):
        # partials: list over local parts of scalars / 1-D arrays
        s = partials[0]
        for p in partials[1:]:
            s = s + p
        return allreduce(np.atleast_1d(np.asarray(s, dtype=np.float64)))

    if spmv_local is None:
        blocks = [H[lo:hi] for lo, hi in parts]

        def matvec(xs):
            x = np.concatenate(xs)
            return [B * x for B in blocks]
    else:
        def matvec(xs):
            return [spmv_local(xs[0])]

    Vs = [np.zeros((n, hi - lo)) for lo, hi in parts]
    for V, (lo, hi) in zip(Vs, parts):
        V[0] = v0g[lo:hi]
    alpha = np.zeros(n)
    beta = np.zeros(n - 1)
    rs = matvec([V[0] for V in Vs])
    alpha[0] = gsum([np.dot(r, V[0]) for r, V in zip(rs, Vs)])[0]
    rs = [r - alpha[0] * V[0] for r, V in zip(rs, Vs)]
    for j in range(n):
        beta[j - 1] = np.sqrt(gsum([np.dot(r, r) for r in rs])[0])
        for V, r in zip(Vs, rs):
            V[j] = r / beta[j - 1]
        c = gsum([np.sum(V[j] * V[: j + 1], axis=1) for V in Vs])
        for V in Vs:
            V[j] = 2 * V[j] - np.sum(c[:, None] * V[: j + 1], axis=0)
        rs = matvec([V[j] for V in Vs])
        alpha[j] = gsum([np.dot(V[j], r) for r, V in zip(rs, Vs)])[0]
        rs = [r - V[j] * alpha[j] - V[j - 1] * beta[j - 1] for r, V in zip(rs, Vs)]
    V = Vs[0] if len(Vs) == 1 else np.concatenate(Vs, axis=1)
    return alpha, beta, V


def execute_lanczos_one_reduce(H, n, bounds, seed=99, v0=None, allreduce=None, spmv_local=None, partial=False, kappa=4.0):
    """The build's ONE-collective-per-step loops, restated (test infrastructure; nothing in the reference to mirror beyond the
    recurrence of Lanczos.py:104-119 they compute): LZ_FLAG_ONE_REDUCE (``partial=False``, run_loop_onereduce) and
    LZ_FLAG_ONE_REDUCE | LZ_FLAG_REORTH_PARTIAL (``partial=True``, run_loop_partial_onereduce, lanczos_amd/csrc/lz_loops.hip).

    State entering step j: r'' = A u - beta v_{j-2} with u = v_{j-1} (j = 0: r'' = A v0, u = v0) and the local partial of
    alpha = u.(A u).  ONE all-reduce carries [p_i = V_i.r'' (i < j), r''.r'' | q_i = V_i.u, u.u, u.r'', alpha]; then
    alpha, |r|^2 = (r''.r'' - 2 alpha u.r'') + alpha^2 u.u, w = (r'' - alpha u) / beta and - when the step sweeps - c_i = (p_i - alpha
    q_i) / beta, c_j = |r|^2 / beta^2, V[j] = 2 w - sum_{i<=j} c_i V_i (row j being w); else V[j] = w.  With ``partial`` the sweep runs
    only where the look-ahead gate (oracle/partial_gates.LookaheadGate: engine 8's device logic) says so, and p, q are zero otherwise.
    Calling convention as ``execute_lanczos_partitioned``.  Returns ``(alpha, beta, V, gates, allreduce_calls)``."""
    from . import partial_gates as pg

    H = as_operator(H).tocsr()
    M = H.shape[0]
    if n > M:
        raise ValueError("n cannot be larger than M!")
    v0g = start_vector(M, seed, v0)
    calls = [0]
    if allreduce is not None:
        lo, hi = bounds
        parts = [(lo, hi)]
        ar0 = allreduce

        def allreduce(x):
            calls[0] += 1
            return ar0(x)
    else:
        parts = [(bounds[p], bounds[p + 1]) for p in range(len(bounds) - 1)]

        def allreduce(x):
            calls[0] += 1
            return x

    def gsum(partials):
        s_ = partials[0]
        for p_ in partials[1:]:
            s_ = s_ + p_
        return allreduce(np.atleast_1d(np.asarray(s_, dtype=np.float64)))

    if spmv_local is None:
        blocks = [H[lo:hi] for lo, hi in parts]

        def matvec(xs):
            x = np.concatenate(xs)
            return [B * x for B in blocks]
    else:
        def matvec(xs):
            return [spmv_local(xs[0])]

    Vs = [np.zeros((n, hi - lo)) for lo, hi in parts]
    for V, (lo, hi) in zip(Vs, parts):
        V[0] = v0g[lo:hi]
    alpha = np.zeros(n)
    beta = np.zeros(n - 1)
    gate = pg.LookaheadGate(n, kappa) if partial else None
    rs = matvec([V[0] for V in Vs])  # r'' = A v0
    apart = [np.dot(r, V[0]) for r, V in zip(rs, Vs)]
    for j in range(n):
        m, urow = j, (j - 1 if j > 0 else 0)
        sweep = gate.gates[j] if partial else True
        us = [V[urow].copy() for V in Vs]
        loc = []
        for V, r, u, ap in zip(Vs, rs, us, apart):
            p_ = V[:m] @ r if sweep else np.zeros(m)
            q_ = V[:m] @ u if sweep else np.zeros(m)
            loc.append(np.concatenate([p_, [np.dot(r, r)], q_, [np.dot(u, u), np.dot(u, r), ap]]))
        buf = gsum(loc)  # THE collective of this iteration
        p_, rr, q_, uu, ur, a = buf[:m], buf[m], buf[m + 1:2 * m + 1], buf[2 * m + 1], buf[2 * m + 2], buf[2 * m + 3]
        alpha[urow] = a
        vv = (rr - 2.0 * a * ur) + a * a * uu
        bj = np.sqrt(vv)
        beta[j - 1] = bj
        for V, r, u in zip(Vs, rs, us):
            w = (r - u * a) / bj
            if sweep:
                c = np.concatenate([(p_ - a * q_) / bj, [vv / (bj * bj)]])
                V[j] = w
                V[j] = 2 * w - np.sum(c[:, None] * V[: j + 1], axis=0)
            else:
                V[j] = w
        if partial:
            gate.step(j, a, float(bj))
        rs = matvec([V[j] for V in Vs])
        apart = [np.dot(V[j], r) for r, V in zip(rs, Vs)]
        if j == n - 1:
            alpha[j] = gsum(apart)[0]  # the last alpha has no pass to ride on
            break
        if j > 0:
            rs = [r - V[j - 1] * bj for r, V in zip(rs, Vs)]
    V = Vs[0] if len(Vs) == 1 else np.concatenate(Vs, axis=1)
    gates = list(gate.gates) if partial else [True] * n
    return alpha, beta, V, gates, calls[0]


# --------------------------------------------------------------------------
# Conditioning probe.  Late Lanczos coefficients are ill-conditioned functions
# of the start vector once Ritz values converge or the Krylov space is nearly
# exhausted (highly symmetric grids): ANY change of summation order then moves
# them by O(1).  Parity tests therefore compare only what the reference
# arithmetic itself determines: quantities that do not move when the same
# recurrence is evaluated with reversed-order BLAS inner products.
# --------------------------------------------------------------------------
def execute_lanczos_reordered(H, n, seed=99, v0=None):
    """Same recurrence as ``execute_lanczos``; inner products summed in another order."""
    H = as_operator(H)
    M = H.shape[0]
    v = start_vector(M, seed, v0)
    V = np.zeros((n, M))
    V[0] = v
    alpha = np.zeros(n)
    beta = np.zeros(n - 1)
    r = H * V[0]
    alpha[0] = r[::-1] @ V[0][::-1]
    r = r - alpha[0] * V[0]
    for j in range(n):
        beta[j - 1] = np.sqrt(r[::-1] @ r[::-1])
        V[j] = r / beta[j - 1]
        c = V[: j + 1] @ V[j]
        V[j] = 2 * V[j] - np.sum(c[:, None] * V[: j + 1], axis=0)
        r = H * V[j]
        alpha[j] = V[j][::-1] @ r[::-1]
        r = r - V[j] * alpha[j] - V[j - 1] * beta[j - 1]
    return alpha, beta, V


def stable_masks(H, n, alpha, beta, seed=99, v0=None, tol=1e-12):
    """(prefix, ritz_mask): number of leading alpha/beta entries and the mask of
    (sorted) Ritz values that are reproduced to ``tol`` (relative to the spectral
    scale) by the reordered evaluation."""
    a2, b2, _ = execute_lanczos_reordered(H, n, seed, v0)
    th = np.linalg.eigvalsh(build_h_eff(alpha, beta))
    th2 = np.linalg.eigvalsh(build_h_eff(a2, b2))
    scale = np.abs(th).max()
    bad = np.abs(a2 - alpha) > tol * scale
    bad[:-1] |= np.abs(b2 - beta) > tol * scale
    prefix = int(np.argmax(bad)) if bad.any() else n
    return prefix, np.abs(th2 - th) <= tol * scale


def stable_basis_rows(H, n, V, seed=99, v0=None, tol=1e-11):
    """Number of leading basis rows that the reordered evaluation reproduces to ``tol`` (absolute; rows are unit
    vectors).  The vectors lose determinacy a few steps BEFORE the coefficients do: near the end of the stable
    coefficient prefix they already differ by 1e-8 between two summation orders of the reference arithmetic itself."""
    _, _, V2 = execute_lanczos_reordered(H, n, seed, v0)
    bad = np.abs(V2 - V).max(axis=1) > tol
    return int(np.argmax(bad)) if bad.any() else n


def converged_ritz(alpha, beta, tol=1e-9):
    """Ritz values of T(alpha, beta) whose residual bound max(beta)*|S[n-1, i]| is below
    ``tol`` times the spectral scale: these approximate true eigenvalues and are
    insensitive to the rounding path that produced the late coefficients."""
    theta, S = np.linalg.eigh(build_h_eff(alpha, beta))
    scale = np.abs(theta).max()
    return theta[np.abs(S[-1]) * np.abs(beta).max() <= tol * scale]
