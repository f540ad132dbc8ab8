"""GPU: the two-sided (bi-orthogonal) Lanczos of the Irregular copy through the drop-in class surface, against the
reference's golden vectors and the CPU oracle.

The recurrence is numerically unstable in the reference itself (a 1e-16 relative perturbation of the start pair grows
to 1e-7 within 20-40 steps, see `sensitivity`), so coefficients are compared on the prefix the reference arithmetic
determines to 1e-12, with the north-star bar of 1e-10 relative there; single bi-orthogonalisation steps are compared
tightly."""
import os

import numpy as np
import scipy.sparse
import pytest

from conftest import GOLDEN_DIR, load_golden, two_sided_names
from lanczos_amd import IrrLanczos, synthetic, _capi
from oracle import two_sided_ref as ts

pytestmark = pytest.mark.gpu


def sensitivity(H, n, seed):
    """|change| of (alpha, beta, gamma) under a 1e-16 relative perturbation of the start pair, per step, relative."""
    M = H.shape[0]
    q0, p0 = ts.start_pair(M, seed)
    a, b, g, _ = ts.execute_two_sided(H, n, start=(q0, p0))
    rng = np.random.default_rng(0)
    a2, b2, g2, _ = ts.execute_two_sided(H, n, start=(q0 * (1 + 1e-16 * rng.standard_normal(M)), p0 * (1 + 1e-16 * rng.standard_normal(M))))
    scale = max(np.abs(a).max(), np.abs(b).max())
    da = np.abs(a - a2) / scale
    dbg = np.maximum(np.abs(b - b2), np.abs(g - g2)) / scale
    return da, dbg, scale


@pytest.mark.parametrize("name", two_sided_names())
def test_golden(name):
    d, H = load_golden(name)
    n, seed, M = int(d["n"]), int(d["seed"]), int(d["M"])
    IrrLanczos.verbose = False
    s = IrrLanczos(H)
    s.execute_Lanczos(n, seed=seed)
    da, dbg, scale = sensitivity(H, n, seed)
    ka = n if (da <= 1e-12).all() else int(np.argmax(da > 1e-12))
    kb = n - 1 if (dbg <= 1e-12).all() else int(np.argmax(dbg > 1e-12))
    assert ka >= min(n, 8) and kb >= min(n - 1, 7)
    assert np.abs(s._alpha - d["alpha"])[:ka].max() <= 1e-10 * scale
    assert np.abs(s._beta - d["beta"])[:kb].max() <= 1e-10 * scale
    assert np.abs(s._gamma - d["gamma"])[:kb].max() <= 1e-10 * scale
    # the published H_eff has the reference's layout
    assert np.array_equal(s.H_eff, ts.build_h_eff(s._alpha, s._beta, s._gamma))
    assert s.V.shape == (M, n)
    if "V" in d:
        k = min(ka, kb, 6)
        np.testing.assert_allclose(s.V[:, :k], d["V"][:k].T, rtol=0, atol=1e-9 * np.abs(d["V"][:k]).max())
    # get_H_eigs of the Irregular copy (no asserts): eigh on the lower triangle, Y = V S on the device
    s.get_H_eigs()
    theta, S = np.linalg.eigh(s.H_eff)
    assert np.array_equal(s.H_eigvals, theta)
    Y = s.V @ S
    np.testing.assert_allclose(s.H_eigvecs, Y, rtol=0, atol=1e-10 * max(1.0, np.abs(Y).max()))
    if ka == n:
        # get_H_eigs is eigh on the lower triangle, i.e. the symmetric tridiagonal (alpha, beta): by Weyl's inequality the
        # Ritz values move by at most |dT|_2 <= max|d alpha| + 2 max|d beta|, which the coefficient checks above already
        # hold to the north-star bar - so the same 1e-10 bar applies to the Ritz values (a factor 3 is the inequality's).
        delta = max(np.abs(s._alpha - d["alpha"]).max(), np.abs(s._beta - d["beta"]).max())
        assert np.abs(s.H_eigvals - d["H_eigvals"]).max() <= max(3 * delta, 1e-13 * scale)
        assert np.abs(s.H_eigvals - d["H_eigvals"]).max() <= 1e-10 * max(np.abs(d["H_eigvals"]).max(), scale)


def test_bireorthogonalize_step_matches_oracle():
    rng = np.random.default_rng(3)
    n, M, j = 9, 3001, 6
    Q, P = rng.standard_normal((n, M)), rng.standard_normal((n, M))
    Qb, _ = np.linalg.qr(rng.standard_normal((M, n)))
    Pb, _ = np.linalg.qr(rng.standard_normal((M, n)))
    Qb, Pb = np.ascontiguousarray(Qb.T), np.ascontiguousarray(Pb.T)
    Qb[j:], Pb[j:] = 0.0, 0.0
    ref = [a.copy() for a in (Q, P, Qb, Pb)]
    ts.bireorthogonalize(*ref, j)
    got = [a.copy() for a in (Q, P, Qb, Pb)]
    IrrLanczos.bireorthogonalize(*got, j)
    for r, g in zip(ref, got):
        assert np.array_equal(r[:j], g[:j]) and np.array_equal(r[j + 1:], g[j + 1:])  # only row j is touched
        assert np.abs(r[j] - g[j]).max() <= 1e-13 * np.abs(r[j]).max()
    q, p = got[0][j], got[1][j]
    assert abs(abs(q @ p) - 1) < 1e-13
    assert np.abs(got[3][:j] @ q).max() < 1e-12 * np.linalg.norm(q) and np.abs(got[2][:j] @ p).max() < 1e-12 * np.linalg.norm(p)
    assert abs(np.linalg.norm(got[2][j]) - 1) < 1e-14 and np.abs(got[2][:j] @ got[2][j]).max() < 1e-13


def test_bireorthogonalize_j0_behaves_like_the_reference():
    """``bireorthogonalize(V1, V2, q_basis, p_basis, 0)`` (IrrLanczos.py:408-441 with both projection loops empty): the reference
    rescales the pair, seeds both orthonormal bases IN PLACE and then raises ValueError from its closing ``np.max`` over an empty
    array.  Held to the reference's own outputs (tests/golden/bireorth_default_j0.npz: the four rows and the exception)."""
    d = np.load(os.path.join(GOLDEN_DIR, "bireorth_default_j0.npz"))
    arrs = [d[k].copy() for k in ("V1", "V2", "q_basis", "p_basis")]
    with pytest.raises(ValueError) as ei:
        IrrLanczos.bireorthogonalize(*arrs, 0)
    assert str(d["error_type"]) == "ValueError" and str(ei.value) == str(d["error_text"])
    for k, got in zip(("V1", "V2", "q_basis", "p_basis"), arrs):
        want = d[k + "_out0"]
        assert np.abs(got[0] - want).max() <= 1e-13 * np.abs(want).max(), k
        assert np.array_equal(got[1:], d[k][1:])  # only row 0 is touched
    assert d["V1"][0] @ d["V2"][0] < 0 and abs(arrs[0][0] @ arrs[1][0] - 1) < 1e-13  # the sign factor of :420 flipped V2[0]
    with pytest.raises(IndexError):
        IrrLanczos.bireorthogonalize(*arrs, 4)


def test_bireorthogonalize_mem_safe_matches_the_reference():
    """the static method's other branch (IrrLanczos.py:398-407) against outputs of the reference's own static method
    (tests/golden/bireorth_mem_safe.npz): rows past j filled or zero, j = 0, and a longer random case against the oracle"""
    d = np.load(os.path.join(GOLDEN_DIR, "bireorth_mem_safe.npz"))
    for tag in "abc":
        V1, V2, j = d[tag + "_V1"].copy(), d[tag + "_V2"].copy(), int(d[tag + "_j"])
        IrrLanczos.bireorthogonalize(V1, V2, None, None, j, mem_safe=True)
        for got, want, src in ((V1, d[tag + "_out1"], d[tag + "_V1"]), (V2, d[tag + "_out2"], d[tag + "_V2"])):
            assert np.abs(got[j] - want).max() <= 1e-13 * np.abs(want).max()
            assert np.array_equal(np.delete(got, j, 0), np.delete(src, j, 0))  # only row j is touched
    rng = np.random.default_rng(8)
    n, M, j = 40, 70001, 17
    V1, V2 = rng.standard_normal((n, M)), rng.standard_normal((n, M))
    V1[j + 1:] = 0
    V2[j + 1:] = 0
    R1, R2 = V1.copy(), V2.copy()
    ts.bireorthogonalize_mem_safe(R1, R2, j)
    IrrLanczos.bireorthogonalize(V1, V2, None, None, j, mem_safe=True)
    assert np.abs(V1[j] - R1[j]).max() <= 1e-13 * np.abs(R1[j]).max() and np.abs(V2[j] - R2[j]).max() <= 1e-13 * np.abs(R2[j]).max()
    with pytest.raises(ValueError, match="0 <= j < n"):
        IrrLanczos.bireorthogonalize(V1, V2, None, None, n, mem_safe=True)


def test_float32_dtype_is_accepted_with_a_notice(capsys):
    """``execute_Lanczos(..., dtype=np.float32)`` (IrrLanczos.py:77, no caller in the reference): inputs rounded to float32 where
    the reference rounds them, recurrence in float64, results published as float32 - within float32 rounding of the float64 run
    on the float32-rounded matrix, and V comes back as a float32 array like the reference's ``q``."""
    rng = np.random.default_rng(12)
    A = scipy.sparse.random(400, 400, density=0.02, random_state=rng, format="csr") + scipy.sparse.diags(np.linspace(1, 3, 400))
    IrrLanczos.verbose = False
    s32 = IrrLanczos(A)
    s32.execute_Lanczos(6, dtype=np.float32)
    assert "dtype=float32" in capsys.readouterr().out
    assert s32.V.dtype == np.float32 and s32.V.shape == (400, 6) and s32.H_eff.dtype == np.float64
    s64 = IrrLanczos(scipy.sparse.csr_matrix(A, dtype=np.float32))  # the same rounded matrix, float64 run
    s64.execute_Lanczos(6)
    # (the start pair is rounded to float32 as well, and the two-sided recurrence amplifies a perturbation ~2x per step)
    assert np.abs(s32.H_eff - s64.H_eff).max() <= 1e-4 * np.abs(s64.H_eff).max()
    assert np.array_equal(s32.H_eff, s32.H_eff.astype(np.float32).astype(np.float64))  # coefficients are float32 values


def test_nonsymmetric_uses_the_transpose():
    """s = H^T p must run on the uploaded transpose: alpha_0 = (p0.Hq0 + q0.H^T p0)/2 and the first beta/gamma."""
    d, H = load_golden("two_sided_nonsym_M300_n10")
    assert (abs(H - H.T)).nnz > 0
    IrrLanczos.verbose = False
    s = IrrLanczos(H)
    s.execute_Lanczos(4, seed=int(d["seed"]))
    a, b, g, Q = ts.execute_two_sided(H, 4, seed=int(d["seed"]))
    scale = np.abs(a).max()
    assert np.abs(s._alpha - a).max() <= 1e-11 * scale and np.abs(s._beta - b).max() <= 1e-11 * scale
    assert np.abs(s._gamma - g).max() <= 1e-11 * scale
    np.testing.assert_allclose(s.V, Q.T, rtol=0, atol=1e-10 * np.abs(Q).max())


def test_large_biorthogonality_and_recurrence():
    """Size-independent properties at M = 4e5: q_i . p_j = +-delta_ij on the early pairs, and the two three-term
    relations H q_j = gamma_{j-1} q_{j-1} + alpha_j q_j + beta_j q~_{j+1} hold before bi-orthogonalisation corrections
    become visible (first steps)."""
    A = synthetic.laplacian_2d_5pt(800, 500).to_scipy()
    IrrLanczos.verbose = False
    s = IrrLanczos(A)
    n = 10
    s.execute_Lanczos(n, seed=7)
    h = s._handle
    Q = np.stack([h.bi_get_row(0, i) for i in range(n)])
    P = np.stack([h.bi_get_row(1, i) for i in range(n)])
    G = Q @ P.T
    assert np.abs(np.abs(np.diag(G)) - 1).max() < 1e-12
    assert np.abs(G - np.diag(np.diag(G))).max() < 1e-9
    Qb = np.stack([h.bi_get_row(2, i) for i in range(n)])
    assert np.abs(Qb @ Qb.T - np.eye(n)).max() < 1e-12
    # j = 0: H q0 - alpha0 q0 is parallel to q1 up to the projection on p_basis[0] removed by bireorthogonalize
    r = A @ Q[0] - s._alpha[0] * Q[0]
    assert abs(np.sqrt(abs((A @ Q[0] - s._alpha[0] * Q[0]) @ (A.T @ P[0] - s._alpha[0] * P[0]))) - s._beta[0]) <= 1e-12 * s._beta[0]
    assert np.isfinite(s.H_eff).all() and r.shape == (A.shape[0],)


def test_error_behaviour():
    A = synthetic.laplacian_2d_5pt(8, 8).to_scipy()
    IrrLanczos.verbose = False
    s = IrrLanczos(A)
    with pytest.raises(ValueError, match="n cannot be larger than M!"):
        s.execute_Lanczos(65)
    with pytest.raises(UnboundLocalError):  # the reference defines the second start vector only for v0=None
        s.execute_Lanczos(4, v0=np.ones(64))
    with pytest.raises(UnboundLocalError):
        s.execute_Lanczos(1)
    s.strict_use_cuda = True
    with pytest.raises(NotImplementedError):
        s.execute_Lanczos(4, use_cuda=False)
    with pytest.raises(ValueError, match="Lanczos Algorithm has not been called."):
        s.H_eff
    with pytest.raises(NotImplementedError, match="float64 or float32"):
        s.execute_Lanczos(4, dtype=np.float16)
    s.strict_use_cuda = False
    s.execute_Lanczos(4, use_cuda=False)  # accepted: one notice line, then the HIP path
    t = IrrLanczos(A)
    t.execute_Lanczos(4)
    assert np.array_equal(s.H_eff, t.H_eff)


def test_run_to_run_bit_reproducibility_and_handle_reuse():
    """Fixed-order reductions: two runs give identical bits; the same object can switch between the two-sided and the
    symmetric solver (the bases are re-planned per run)."""
    A = synthetic.random_graph_laplacian(5000, 17000, seed=2).to_scipy()
    IrrLanczos.verbose = False
    s = IrrLanczos(A)
    s.execute_Lanczos(12, seed=4)
    H1, V1 = s.H_eff.copy(), s.V.copy()
    s.execute_LanczosOld(10, seed=4)
    assert s.H_eff.shape == (10, 10) and np.array_equal(s.H_eff, s.H_eff.T)
    s.execute_Lanczos(12, seed=4)
    assert np.array_equal(H1, s.H_eff) and np.array_equal(V1, s.V)


@pytest.mark.parametrize("build,n", [(lambda: synthetic.random_graph_laplacian(5000, 17000, seed=2).to_scipy(), 14),
                                     (lambda: synthetic.laplacian_2d_5pt(300, 300).to_scipy(), 20),
                                     (lambda: synthetic.laplacian_2d_5pt(1000, 700).to_scipy(), 10)])
def test_deferred_fold_links_are_bit_identical(build, n, kb):
    """A/B arm of the Gram-Schmidt links (tuning knob 11 = 3: one launch each, the fold of a link's four sums rides in the next
    link's prologue, in the fold kernel's own order) against the default two-launch links: identical H_eff and bases; timed
    (device time) - the arm measured 0.81-0.92x, which is why it is not the default."""
    A = build()
    IrrLanczos.verbose = False
    out = []
    for knob in (0, 3):
        s = IrrLanczos(A)
        if knob:  # the retired arm lives in the kernel-bench build: this object's handle is created on that library
            s._handle = kb.Handle(0)
            s._handle_devices = ((0,), s.comm_backend)
        s.execute_Lanczos(3, seed=4)  # creates the handle (and loads the code objects)
        s._handle.set_tuning(_capi.TUNE_BI_LINKS, knob)
        s.execute_Lanczos(n, seed=4)
        s.execute_Lanczos(n, seed=4)
        dt = s._timings["total_ms"] * 1e-3  # device time of that run (host-side set-up of the class excluded)
        out.append((s.H_eff.copy(), s.V.copy(), dt))
    (H1, V1, t1), (H0, V0, t0) = out
    assert np.array_equal(H0, H1) and np.array_equal(V0, V1)
    print(f"\n[two-sided links] M={A.shape[0]} n={n}: two launches per link {1e3 * t1:.2f} ms, one launch (deferred fold) {1e3 * t0:.2f} ms, x{t1 / t0:.2f}")
