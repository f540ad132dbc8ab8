"""End-to-end parity of the HIP path against the reference's golden vectors and
against the CPU oracle on seeded inputs, through the drop-in class surface.

Tolerance (BASELINE.json north_star): Ritz values within 1e-10 relative - taken
relative to the spectral scale max|theta| because the test Laplacians are
singular (lambda_min = 0).
"""
import os

import numpy as np
import pytest
import scipy.sparse

from conftest import golden_names, load_golden
from lanczos_amd import IrrLanczos, Lanczos, _capi, synthetic
from oracle import lanczos_ref as oracle

pytestmark = pytest.mark.gpu

RTOL = 1e-10


def ritz_close(theta, ref, mask=None):
    scale = np.abs(ref).max()
    err = np.abs(theta - ref)
    if mask is not None:
        err = err[mask]
    return err.max() <= RTOL * scale


@pytest.mark.parametrize("name", [g for g in golden_names() if "n1001" not in g])
def test_golden(name):
    d, H = load_golden(name)
    n, seed = int(d["n"]), int(d["seed"])
    v0 = d["v0"] if "v0" in d else None
    Lanczos.verbose = False
    Hin = H.toarray() if name.startswith(("c1_dense", "box1d")) else H  # dense ndarray input where the reference scripts use one
    s = Lanczos(Hin)
    s.execute_Lanczos(n, seed=seed, v0=v0)
    alpha, beta = np.diag(s.H_eff), np.diag(s.H_eff, 1)
    scale = max(np.abs(d["alpha"]).max(), np.abs(d["beta"]).max())
    # compare what the reference arithmetic itself determines (see oracle.stable_masks)
    prefix, mask = oracle.stable_masks(H, n, d["alpha"], d["beta"], seed=seed, v0=v0)
    assert prefix >= min(n, 20) and mask.sum() >= min(n, 10)
    assert np.abs(alpha - d["alpha"])[:prefix].max() <= 1e-10 * scale
    assert np.abs(beta - d["beta"])[: max(prefix - 1, 1)].max() <= 1e-10 * scale
    assert np.array_equal(s.H_eff, s.H_eff.T)
    if prefix == n:  # well-conditioned to the end: every Ritz value must match
        assert ritz_close(s.H_eigvals, d["H_eigvals"])
    else:  # Krylov space (nearly) exhausted: late coefficients are rounding noise in the reference itself
        conv = oracle.converged_ritz(d["alpha"], d["beta"])
        assert len(conv) >= 5
        nearest = np.abs(s.H_eigvals[None, :] - conv[:, None]).min(axis=1)
        assert nearest.max() <= RTOL * np.abs(d["H_eigvals"]).max()
    assert s.V.shape == (int(d["M"]), n) and s.H_eigvecs.shape == (int(d["M"]), n)
    if "V" in d:
        # Lanczos vectors are only determined up to the growth of rounding differences; compare the projector
        # single-pass Gram-Schmidt loses orthogonality like eps/beta when beta collapses (Krylov space exhausted):
        # hold the device basis to the reference's own level on this fixture
        ref_orth = np.abs(d["V"] @ d["V"].T - np.eye(n)).max()
        assert np.abs(s.V.T @ s.V - np.eye(n)).max() < max(1e-12, 100 * ref_orth)
        # the golden basis on the WHOLE prefix of rows the reference arithmetic determines (to 1e-11 under a change of
        # summation order; the vectors lose determinacy a few steps before the coefficients do)
        vrows = oracle.stable_basis_rows(H, n, d["V"], seed=seed, v0=v0)
        assert vrows >= min(prefix, 20) - 5 and vrows >= min(n, 10)
        ref = d["V"][:vrows].T  # entries of size ~ 1 / sqrt(M): the bar is RELATIVE to the largest of them
        assert np.abs(s.V[:, :vrows] - ref).max() <= 1e-10 * np.abs(ref).max()
    if "H_eigvecs_first3" in d:
        # Ritz vectors of the three lowest Ritz values against the reference's, sign-fixed.  A Ritz vector is only as well
        # determined as its Ritz value is isolated in T (sin(angle) ~ |dT| / gap), and - when the run outlives the stable
        # prefix - only if it has converged; the others are skipped, never loosened.
        ref_th, ref_Y = d["H_eigvals"], d["H_eigvecs_first3"]
        conv = oracle.converged_ritz(d["alpha"], d["beta"]) if prefix < n else ref_th
        compared = 0
        for i in range(ref_Y.shape[1]):
            gap = np.abs(np.delete(ref_th, i) - ref_th[i]).min()
            if gap <= 1e-6 * np.abs(ref_th).max() or np.abs(conv - ref_th[i]).min() > 1e-12 * np.abs(ref_th).max():
                continue
            y, yr = s.H_eigvecs[:, i], ref_Y[:, i]
            y = y * np.sign(y @ yr)
            # sin(angle) ~ |dT| / gap: the bar scales with how isolated the Ritz value is (never looser than 1e-9 relative)
            tol = min(1e-9, 1e-10 * max(1.0, 1e-2 * np.abs(ref_th).max() / gap)) * np.abs(yr).max()
            assert np.abs(y - yr).max() <= tol, (i, np.abs(y - yr).max(), tol)
            compared += 1
        assert compared >= 1 or n <= 2
    # same checks get_H_eigs ran in the reference
    assert abs(Lanczos.test_is_normalized(s.H_eigvecs, no_assert=True) - float(d["norm_closest_to_1"])) < 1e-9


def test_golden_n_equals_M_edge():
    """1Ddeuteron.py: N = n = 1001.  The Krylov space is exhausted, late beta are rounding
    noise; the reference survives it, and so must we (finite output, well-separated low Ritz values agree)."""
    d, H = load_golden("deuteron1d_N1001_n1001")
    Lanczos.verbose = False
    s = Lanczos(H)
    s.execute_Lanczos(1001)
    assert np.isfinite(s.H_eff).all()
    th = np.linalg.eigvalsh(s.H_eff)
    scale = np.abs(d["H_eigvals"]).max()
    # North-star bar (1e-10 of the spectral scale) on every Ritz value the reference arithmetic itself determines
    # (unchanged to 1e-12 when its inner products are summed in another order, oracle.stable_masks): on this fixture
    # that is the whole spectrum - the tridiagonal 1-D problem keeps beta = O(1) to the last step.
    prefix, mask = oracle.stable_masks(H, 1001, d["alpha"], d["beta"])
    assert mask[:20].all() and mask.sum() >= 900
    assert ritz_close(th, d["H_eigvals"], mask)


def test_irregular_facade_matches_regular():
    d, H = load_golden("lap2d_32x32_n30")
    IrrLanczos.verbose = Lanczos.verbose = False
    a = Lanczos(H)
    a.execute_Lanczos(30)
    b = IrrLanczos(H.tocsc())
    b.execute_LanczosOld(30)
    assert np.array_equal(a.H_eff, b.H_eff)  # symmetric CSC == CSR arrays, deterministic kernels
    assert np.array_equal(a.V, b.V)
    b.get_H_eigsOld()
    assert ritz_close(b.H_eigvals, d["H_eigvals"])


@pytest.mark.parametrize(
    "build,n",
    [
        (lambda: synthetic.laplacian_2d_5pt(300, 200), 60),
        (lambda: synthetic.laplacian_3d_7pt(40, 30, 20), 50),
        (lambda: synthetic.random_graph_laplacian(50000, 175000, seed=1234), 60),
    ],
)
def test_seeded_vs_oracle(build, n):
    H = build().to_scipy()
    Lanczos.verbose = False
    s = Lanczos(H)
    s.execute_Lanczos(n)  # default seed 99, legacy RNG start vector
    a, b, V = oracle.execute_lanczos(H, n, economy=True)
    theta = np.linalg.eigvalsh(oracle.build_h_eff(a, b))
    assert ritz_close(s.H_eigvals, theta)
    scale = max(np.abs(a).max(), np.abs(b).max())
    assert np.abs(np.diag(s.H_eff) - a).max() <= 1e-10 * scale and np.abs(np.diag(s.H_eff, 1) - b).max() <= 1e-10 * scale
    # Lanczos invariants on the device result
    Vd = s.V
    assert np.abs(Vd.T @ Vd - np.eye(n)).max() < 1e-12
    R = H @ Vd - Vd @ s.H_eff
    assert np.abs(R[:, :-1]).max() < 1e-11 * scale
    # Ritz vectors: Y = V S
    th, S = np.linalg.eigh(s.H_eff)
    np.testing.assert_allclose(s.H_eigvecs, Vd @ S, rtol=0, atol=1e-13)


def test_c2_full_size_properties():
    """BASELINE config C2 at full size (2-D 5-pt, M = 1e6, k = 100): size-independent properties
    (orthonormal basis, Lanczos relation, run-to-run bit reproducibility) + Ritz values vs the oracle."""
    H = synthetic.laplacian_2d_5pt(1000, 1000)
    Hs = H.to_scipy()
    n = 100
    Lanczos.verbose = False
    s = Lanczos(Hs)
    s.execute_Lanczos(n)
    H_eff1 = s.H_eff.copy()
    V = s.V
    G = V.T @ V
    assert np.abs(G - np.eye(n)).max() < 1e-12
    R = Hs @ V - V @ s.H_eff
    assert np.abs(R[:, :-1]).max() < 1e-11 * 8
    s2 = Lanczos(Hs)
    s2.execute_Lanczos(n)
    assert np.array_equal(H_eff1, s2.H_eff)  # deterministic reductions
    a, b, _ = oracle.execute_lanczos(Hs, n, economy=True)
    assert ritz_close(s.H_eigvals, np.linalg.eigvalsh(oracle.build_h_eff(a, b)))


def test_error_surface_on_device():
    Lanczos.verbose = False
    s = Lanczos(synthetic.laplacian_2d_5pt(8, 8).to_scipy())
    with pytest.raises(ValueError, match="n cannot be larger than M!"):
        s.execute_Lanczos(65)
    with pytest.raises(ValueError, match="Lanczos Algorithm has not been called."):
        s.H_eff
    with pytest.raises(IndexError):
        s.execute_Lanczos(1)
    s.execute_Lanczos(2)
    d, _ = load_golden("lap2d_8x8_n2")
    assert ritz_close(s.H_eigvals, d["H_eigvals"])


def test_integer_and_list_start_vectors_are_accepted_like_the_reference():
    """Lanczos.py:99-100 copies the caller's v0 and divides OUT of place: an int array or a plain list becomes float64 there.
    (an in-place divide raised UFuncTypeError here.)"""
    Lanczos.verbose = False
    H = synthetic.laplacian_2d_5pt(20, 13).to_scipy()
    M, n = H.shape[0], 12
    vi = (np.arange(M) % 7 - 3).astype(np.int64)
    vi[0] = 5
    ref = Lanczos(H)
    ref.execute_Lanczos(n, v0=vi.astype(np.float64))
    for v0 in (vi, vi.tolist(), vi.astype(np.int32), vi.astype(np.float32)):
        keep = np.array(v0).copy()
        s = Lanczos(H)
        s.execute_Lanczos(n, v0=v0)
        if np.array(v0).dtype == np.float32:  # v0 / norm(v0) stays float32 in NumPy (in the reference too): a start vector rounded to 24 bits
            assert np.abs(s.H_eff - ref.H_eff).max() < 1e-5
        else:
            assert np.array_equal(s.H_eff, ref.H_eff)
        assert np.array_equal(np.array(v0), keep)  # the caller's vector is never written to
    a, b, _ = oracle.execute_lanczos(H, n, v0=vi)
    assert ritz_close(ref.H_eigvals, np.linalg.eigvalsh(oracle.build_h_eff(a, b)))


@pytest.mark.parametrize("flags", [2, 4, 16, 4 | 16, 32, 8, 16 | 256])
def test_kernel_variants_agree(flags, hip, kb):
    """VALU (4) Q^T w arm, fused-norm (16, also with VALU), generic CSR-stream (32) and scalar SpMV (8) arms, and the
    one-all-reduce-per-iteration loop (256: pass 1 dots two columns at once, alpha and |r|^2 come out of the same reduced
    buffer) against the default path (4x4x4-MFMA Q^T w, fixed-K SpMV); the retired 16x16x4-MFMA Q^T w arm (2) in the
    kernel-bench build."""
    _capi = kb if flags == 2 else hip

    A = synthetic.laplacian_2d_5pt(300, 200)
    M = A.shape[0]
    v0 = synthetic.reference_start_vector(M)
    v0 /= np.linalg.norm(v0)
    out = []
    for f in (0, flags):
        h = _capi.Handle(0)
        h.set_options(f)
        h.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
        a, b = h.run(60, v0)
        V = h.get_basis()
        out.append((a, b, V))
        h.close()
    (a0, b0, V0), (a1, b1, V1) = out
    assert np.abs(a1 - a0).max() < 1e-11 and np.abs(b1 - b0).max() < 1e-11
    assert np.abs(V1[:10] - V0[:10]).max() < 1e-10
    assert np.abs(V1 @ V1.T - np.eye(60)).max() < 1e-12


def test_device_gram_and_quality_match_numpy():
    """lz_ritz_gram / lz_ritz_quality (device-side versions of test_is_normalized / test_is_orthogonal and the
    print_good_eigs residual loop, Lanczos.py:157-158,166-175) against NumPy on the fetched Ritz vectors."""
    H = synthetic.laplacian_2d_5pt(211, 97).to_scipy()
    n = 37
    Lanczos.verbose = False
    s = Lanczos(H)
    s.execute_Lanczos(n)
    s.get_H_eigs()  # runs both asserts on the device Gram matrix
    Y = s.H_eigvecs
    G = s._handle.ritz_gram()
    np.testing.assert_allclose(G, Y.T @ Y, rtol=0, atol=1e-13)
    q = s._handle.ritz_quality()
    HY = H @ Y
    ref = np.einsum("ij,ij->j", HY / np.linalg.norm(HY, axis=0), Y) ** 2
    np.testing.assert_allclose(q, ref, rtol=1e-11, atol=1e-13)
    assert abs(Lanczos.test_is_normalized(Y, no_assert=True) - np.sqrt(np.diag(G))[np.argmin(np.abs(np.sqrt(np.diag(G)) - 1))]) < 1e-13
    import contextlib, io

    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        s.print_good_eigs()
    assert "Eigvec InnerProd" in buf.getvalue() and len(buf.getvalue().splitlines()) == 22


@pytest.mark.parametrize("M,n,chunk", [(4096, 48, 0), (70001, 100, 0), (30011, 200, 0), (20000, 208, 0), (50000, 117, 0), (30011, 200, 8000), (5000, 33, 0), (5000, 32, 0), (3000, 100, 0),
                                       (9000, 209, 0), (8200, 353, 0), (6000, 500, 0), (12000, 420, 5000),
                                       # round 5: group sizes 7 .. 11 (n = 224: 2 x 7, 300: 2 x 10, 400: 4 x 7, 500: 4 x 8, 520: 3 x 11, 700: 4 x 11, 290: 2 x 10 ragged),
                                       # even n: operands staged through LDS; odd n: the register-ring kernels
                                       (9000, 224, 0), (9001, 300, 0), (8200, 400, 0), (8300, 520, 0), (5000, 700, 0), (9000, 290, 0), (9000, 399, 0), (4100, 50, 0),
                                       (4099, 112, 0), (8193, 176, 4096)])
def test_symmetric_gram_kernel(hip, M, n, chunk):
    """Round 4: the accumulator-stationary symmetric Gram kernel (k_gram_sym: upper 16 x 16 tiles of Y^T Y kept in the
    accumulators, Y streamed once, result mirrored) against NumPy and against the split-K TN GEMM it replaces (knob 19 = 1),
    on a Y that is NOT orthonormal (random S) so every entry is exercised: ragged last column tile, ragged last k-step,
    resident and chunked Y; more than 208 columns go through groups of 11 column tiles (diagonal units + halves of the
    off-diagonal pairs, padded last group).  Fewer than three column tiles (n <= 32) or fewer than 4096 rows stay on the old path."""
    A = synthetic.random_graph_laplacian(M, 3 * M, seed=5)
    v0 = synthetic.reference_start_vector(M)
    v0 /= np.linalg.norm(v0)
    out = {}
    S = np.random.default_rng(3).standard_normal((n, n))
    for knob in (0, 1, 2):  # 0: symmetric kernel, operands through LDS (round 5); 1: split-K TN GEMM; 2: symmetric kernel, register rings (round 4)
        h = hip.Handle(0)
        h.set_tuning(hip.TUNE_GRAM_KERNEL, knob)
        if chunk:
            h.set_tuning(hip.TUNE_RITZ_CHUNK_ROWS, chunk)
        h.set_options(hip.FLAG_FUSED_NORM)
        h.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
        h.run(n, v0)
        V = h.get_basis()
        h.ritz_vectors(S, fetch=False)
        G = h.ritz_gram()
        info = h.gram_info()
        assert (info["ksteps"] > 0) == (knob != 1 and 32 < n <= 208 and min(M, chunk or M) >= 4096)  # (the clock record: single-group form only)
        out[knob] = G
        h.close()
    Y = V.T @ S
    ref = Y.T @ Y
    scale = np.abs(ref).max()
    assert np.array_equal(out[0], out[0].T)  # mirrored, not computed twice
    assert np.abs(out[0] - ref).max() <= 1e-12 * scale
    assert np.abs(out[1] - ref).max() <= 1e-12 * scale
    # the LDS-staged and the register-ring form issue the same MFMAs in the same order per tile and slice: the same bits
    assert np.array_equal(out[0], out[2])


def test_get_H_eigs_asserts_fire_on_device_gram():
    """A basis that is not orthonormal must trip the reference's asserts (evaluated on the device Gram)."""
    from lanczos_amd import _capi

    H = synthetic.laplacian_2d_5pt(40, 30).to_scipy()
    Lanczos.verbose = False
    s = Lanczos(H)
    s.execute_Lanczos(12)
    S_bad = np.linalg.eigh(s.H_eff)[1].copy()
    S_bad[:, 3] = S_bad[:, 2]  # two identical Ritz vectors
    s._handle.ritz_vectors(S_bad, fetch=False)
    G = s._handle.ritz_gram()
    off = np.abs(G - np.diag(np.diag(G)))
    assert np.sqrt(off.max()) > 0.9


def test_headline_full_size_properties():
    """BASELINE headline size (2-D 5-pt, M = 1e7, k = 200): the first 12 recurrence coefficients against the CPU
    oracle (a k = 12 oracle run produces the same leading coefficients), all 200 against a full-size run of the reference
    itself (golden fixture), and size-independent properties evaluated
    on the device: Ritz vectors orthonormal (Gram), every Ritz value within its residual bound of an analytic
    eigenvalue 4 - 2cos(2 pi p/Nx) - 2cos(2 pi q/Ny) of the periodic Laplacian, spectrum inside the Gershgorin
    interval, bit-reproducible reruns."""
    nx, ny, n = 4000, 2500, 200
    H = synthetic.laplacian_2d_5pt(nx, ny).to_scipy()
    Lanczos.verbose = False
    s = Lanczos(H)
    s.execute_Lanczos(n)
    H_eff = s.H_eff.copy()
    a, b, _ = oracle.execute_lanczos(H, 12, economy=True)
    assert np.abs(np.diag(H_eff)[:12] - a).max() <= 1e-12 * 8
    assert np.abs(np.diag(H_eff, 1)[:11] - b).max() <= 1e-12 * 8
    # ALL 200 coefficients against the reference itself, run at this size on the CPU (oracle/gen_golden_headline.py, 85 min
    # per run; the fixture holds alpha / beta and how far a second run with another BLAS thread count moved each of them):
    # north-star bar 1e-10 of the spectral scale wherever the reference's own arithmetic determines the coefficient to 1e-12
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "headline_lap2d_4000x2500_n200.npz"), allow_pickle=False)
    assert int(gold["M"]) == nx * ny and int(gold["n"]) == n
    st_a, st_b = gold["alpha_moved"] <= 1e-12 * 8, gold["beta_moved"] <= 1e-12 * 8
    assert st_a.sum() >= 150 and st_b.sum() >= 150, (int(st_a.sum()), int(st_b.sum()))
    assert np.abs(np.diag(H_eff) - gold["alpha"])[st_a].max() <= 1e-10 * 8
    assert np.abs(np.diag(H_eff, 1) - gold["beta"])[st_b].max() <= 1e-10 * 8
    s.get_H_eigs()  # device back-transform + the reference's two asserts on the device Gram matrix
    G = s._handle.ritz_gram()
    assert np.abs(G - np.eye(n)).max() < 1e-12
    q = s._handle.ritz_quality()
    theta = s.H_eigvals
    assert theta.min() > -1e-12 and theta.max() < 8 + 1e-12
    lam = (4 - 2 * np.cos(2 * np.pi * np.arange(nx) / nx))[:, None] - 2 * np.cos(2 * np.pi * np.arange(ny) / ny)[None, :]
    lam = np.unique(np.round(lam.ravel(), 13))
    hi = np.searchsorted(lam, theta).clip(0, len(lam) - 1)
    dist = np.minimum(np.abs(lam[hi] - theta), np.abs(lam[(hi - 1).clip(0)] - theta))
    resid = 8.0 * np.sqrt(np.clip(1 - q, 0, None))  # ||A y - theta y|| <= ||A y|| sin(angle(A y, y))
    assert np.all(dist <= resid * (1 + 1e-6) + 1e-10)
    assert q[-1] > 1 - 1e-5 and q[-1] >= q[len(q) // 2]  # the top of the spectrum is the best converged
    s2 = Lanczos(H)
    s2.execute_Lanczos(n)
    assert np.array_equal(H_eff, s2.H_eff)
    # The partial ("selective") re-orthogonalisation mode at the same full size (round 4: decided on the device, the SpMV forms
    # r / beta itself): no host synchronisation inside lz_run, one sweep (j = 0: nothing converges in 200 steps at this size),
    # and - every coefficient being determined here - ALL 200 alpha / beta against the full-size run of the reference itself at the
    # north-star bar, although the basis is only semi-orthogonal by design (held to 1e-7 on the device Gram matrix of its Ritz
    # vectors; measured 1.3e-13: no loss of orthogonality to speak of has built up in 200 steps)
    p = Lanczos(H)
    p.reorth = "partial"
    p.execute_Lanczos(n)
    hp = p._get_handle()
    assert hp.last_engine() == "partial-device" and hp.last_host_syncs() == 0 and p.sweeps == 1
    assert np.abs(np.diag(p.H_eff) - gold["alpha"])[st_a].max() <= 1e-10 * 8
    assert np.abs(np.diag(p.H_eff, 1) - gold["beta"])[st_b].max() <= 1e-10 * 8
    assert np.abs(p.H_eigvals - theta).max() <= 1e-10 * 8
    assert np.abs(hp.ritz_gram() - np.eye(n)).max() < 1e-7
    s2.close()
    # ... and the same with ONE all-reduce per step (round 5, engine 8: look-ahead sweep decision; on one GPU the collective is a no-op,
    # the arithmetic - two-column dots, three-sum |r|^2 - is the partitioned run's): all 200 coefficients against the reference's own
    # full-size run at the north-star bar, no look-ahead miss, no host synchronisation
    q = Lanczos(H)
    q.reorth = "partial"
    q.options = _capi.FLAG_ONE_REDUCE
    q.execute_Lanczos(n)
    hq = q._get_handle()
    assert hq.last_engine() == "partial-one-reduce" and hq.last_host_syncs() == 0 and hq.last_sweep_misses() == 0 and q.sweeps == 1
    assert np.abs(np.diag(q.H_eff) - gold["alpha"])[st_a].max() <= 1e-10 * 8
    assert np.abs(np.diag(q.H_eff, 1) - gold["beta"])[st_b].max() <= 1e-10 * 8
    assert np.abs(q.H_eigvals - theta).max() <= 1e-10 * 8
    for obj in (s, p, q):
        obj.close()


@pytest.mark.parametrize(
    "build,n",
    [
        (lambda: synthetic.laplacian_2d_5pt(300, 200).to_scipy(), 200),      # nothing converges: almost no sweeps
        (lambda: synthetic.laplacian_3d_7pt(20, 18, 16).to_scipy(), 150),    # Ritz values converge: sweeps are needed
        (lambda: load_golden("box1d_N500_n50")[1], 50),
        (lambda: load_golden("deuteron3d_N12_27pt_n100")[1], 100),
    ],
)
def test_partial_reorthogonalisation_opt_in(build, n):
    """LZ_FLAG_REORTH_PARTIAL (opt-in; the reference only has the full sweep): same sweep kernels, run only when the
    omega-recurrence says semi-orthogonality (sqrt(eps)) is about to be lost.  Converged/stable Ritz values must equal
    the full-sweep run's to 1e-10 and the basis must stay semi-orthogonal."""
    H = build()
    Lanczos.verbose = False
    full = Lanczos(H)
    full.execute_Lanczos(n)
    part = Lanczos(H)
    part.reorth = "partial"
    part.execute_Lanczos(n)
    assert full.sweeps == n and 1 <= part.sweeps < n
    scale = np.abs(full.H_eigvals).max()
    conv = oracle.converged_ritz(np.diag(full.H_eff), np.diag(full.H_eff, 1), tol=1e-9)
    th_p = np.linalg.eigvalsh(part.H_eff)
    if len(conv):
        assert np.abs(th_p[None, :] - conv[:, None]).min(axis=1).max() <= RTOL * scale
    a_f, b_f = np.diag(full.H_eff), np.diag(full.H_eff, 1)
    prefix, mask = oracle.stable_masks(H, n, a_f, b_f)
    if prefix == n:  # well-conditioned run: every Ritz value is determined
        assert np.abs(th_p - full.H_eigvals).max() <= RTOL * scale
    V = part.V
    # Round 4: the omega-recurrence and the sweep decision live on the device (engine "partial-device"): lz_run makes no host
    # synchronisation between its first and its last launch.  Decisions, coefficients and basis are bit-identical to the
    # host-decided loop (knob 18 = 1, two scalars read back per step) and to the device loop without the fused r / beta
    # (knob 18 = 2; with an ELL-ordered stencil matrix the SpMV forms v_j = r / beta itself).
    h = part._get_handle()
    assert h.last_engine() == "partial-device" and h.last_host_syncs() == 0
    for knob in (1, 2, 3):  # 3: device loop with the separate second-stage kernel of pass 1 (default: folded into pass 1's last block)
        other = Lanczos(H)
        other.reorth = "partial"
        other._get_handle().set_tuning(_capi.TUNE_PARTIAL_LOOP, knob)
        other.execute_Lanczos(n)
        ho = other._get_handle()
        assert ho.last_engine() == ("kernels" if knob == 1 else "partial-device")
        assert (ho.last_host_syncs() > n) if knob == 1 else (ho.last_host_syncs() == 0)
        assert other.sweeps == part.sweeps
        assert np.array_equal(other.H_eff, part.H_eff)
        assert np.array_equal(other.V, V)
        other.close()
    assert np.abs(V.T @ V - np.eye(n)).max() < 1e-6  # semi-orthogonal (sqrt(eps) level), not eps like the full sweep
    # a sweep removes components of size <= sqrt(eps) from v_j, so A V = V T + ... holds to that level (not eps)
    R = H @ V - V @ part.H_eff
    assert np.abs(R[:, :-1]).max() < 1e-6 * max(scale, 1.0)


def test_partial_loops_agree_bit_for_bit_on_awkward_shapes(hip):
    """The device-decided partial re-orthogonalisation loop against the host-decided one (knob 18 = 1) through the C ABI on 24
    random shapes: stencils whose row count is no multiple of any block size (the fused r / beta over the ELL copy, ragged last
    block), ragged CSR, dense, n from 2 to 60 - alpha, beta, basis, number of sweeps and the sweep log's count, all np.array_equal;
    no host synchronisation in the device loop."""
    rng = np.random.default_rng(2024)
    for trial in range(24):
        kind = ("lap2d", "lap3d", "ragged", "dense")[trial % 4]
        if kind == "lap2d":
            A = synthetic.laplacian_2d_5pt(int(rng.integers(5, 90)), int(rng.integers(5, 90)))
            M, setm = A.shape[0], lambda h, A=A: h.set_csr(A.shape[0], 0, A.rowptr, A.colidx, A.vals)
        elif kind == "lap3d":
            A = synthetic.laplacian_3d_7pt(int(rng.integers(3, 20)), int(rng.integers(3, 20)), int(rng.integers(3, 20)))
            M, setm = A.shape[0], lambda h, A=A: h.set_csr(A.shape[0], 0, A.rowptr, A.colidx, A.vals)
        elif kind == "ragged":
            M = int(rng.integers(40, 3000))
            R = scipy.sparse.random(M, M, density=min(0.3, 8.0 / M), random_state=rng, format="csr")
            S = (R + R.T + scipy.sparse.diags(rng.standard_normal(M))).tocsr()
            S.sort_indices()
            setm = lambda h, S=S, M=M: h.set_csr(M, 0, S.indptr, S.indices, S.data)
        else:
            M = int(rng.integers(20, 400))
            D = synthetic.dense_symmetric(M, seed=trial)
            setm = lambda h, D=D: h.set_dense(D)
        n = int(min(M, rng.integers(2, 61)))
        v0 = rng.uniform(-1, 1, M)
        v0 /= np.linalg.norm(v0)
        out = []
        for knob in (0, 1):
            h = hip.Handle(0)
            h.set_tuning(hip.TUNE_PARTIAL_LOOP, knob)
            h.set_options(hip.FLAG_REORTH_PARTIAL | hip.FLAG_FUSED_NORM)
            setm(h)
            a, b = h.run(n, v0)
            out.append((a, b, h.get_basis(), h.last_sweeps(), h.last_engine(), h.last_host_syncs()))
            h.close()
        (a0, b0, V0, s0, e0, y0), (a1, b1, V1, s1, e1, y1) = out
        tag = f"{kind} M={M} n={n}"
        assert e0 == "partial-device" and y0 == 0 and e1 == "kernels" and y1 >= n, tag
        assert s0 == s1 and 1 <= s0 <= n, tag
        assert np.array_equal(a0, a1, equal_nan=True) and np.array_equal(b0, b1, equal_nan=True) and np.array_equal(V0, V1, equal_nan=True), tag


def test_repeated_default_start_vector_is_cached_and_the_global_rng_ends_where_the_reference_leaves_it():
    """Lanczos.py:93-97 seeds the GLOBAL legacy RNG and draws M doubles on every call.  A repeated call with the same (seed, M)
    reuses the cached normalised vector (the draw is the whole overhead of a second call at the headline size) - same results bit
    for bit - and restores the RNG state saved right after the first draw, so the caller's next np.random draw is what it would be
    after the reference's call."""
    H = synthetic.laplacian_2d_5pt(60, 50).to_scipy()
    M = H.shape[0]
    Lanczos.verbose = False
    s = Lanczos(H)
    s.execute_Lanczos(20, seed=7)
    after_first = np.random.uniform()
    first = s.H_eff.copy()
    np.random.seed(12345)  # the caller does something else with the global RNG in between
    s.execute_Lanczos(20, seed=7)
    after_second = np.random.uniform()
    np.random.seed(7)
    np.random.uniform(-1, 1, size=(M))
    expected = np.random.uniform()
    assert after_first == expected and after_second == expected
    assert np.array_equal(s.H_eff, first)
    s.execute_Lanczos(20, seed=8)  # another seed: a fresh draw
    assert not np.array_equal(s.H_eff, first)
    s.execute_Lanczos(20, seed=7, v0=np.ones(M))  # explicit v0: the RNG is seeded, nothing is drawn (Lanczos.py:93, 98-99)
    x = np.random.uniform()
    np.random.seed(7)
    assert x == np.random.uniform()
    s.close()


def test_randomised_shapes_against_oracle():
    """30 random symmetric matrices of awkward shapes (M = 5 ... 12345 incl. non-multiples of every tile size, n up to
    M, dense / banded / ragged sparse): recurrence coefficients and Ritz values vs the CPU oracle, Y = V S."""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location("stress_parity", os.path.join(os.path.dirname(os.path.abspath(__file__)), "stress_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(seed=1, trials=30, quiet=True) == 0


def test_breakdown_behaves_like_the_reference():
    """Invariant subspace (H = I: the first residual is exactly zero): the reference divides by beta = 0 and carries
    NaNs forward (Lanczos.py:113, no breakdown test); the device path must do the same - and says so with a warning."""
    import scipy.sparse

    H = scipy.sparse.identity(64, format="csr")
    a, b, V = None, None, None
    with np.errstate(all="ignore"):
        a, b, V = oracle.execute_lanczos(H, 5)
    assert not np.isfinite(a).all()
    Lanczos.verbose = False
    s = Lanczos(H)
    with pytest.warns(RuntimeWarning, match="breakdown"):
        s.execute_Lanczos(5)
    assert not np.isfinite(s.H_eff).all()
    assert np.array_equal(np.isfinite(np.diag(s.H_eff)), np.isfinite(a))


def test_many_basis_rows_with_a_long_slice():
    """n = 900 at M = 2.56e6: pass 1 then needs more than 64 KiB of dynamic LDS per block (40 KiB slice of w + four
    coefficient runs of 912 doubles), which has to be allowed per kernel.  Size-independent checks only."""
    A = synthetic.laplacian_2d_5pt(2000, 1280).to_scipy()
    Lanczos.verbose = False
    s = Lanczos(A)
    s.execute_Lanczos(900)
    h = s._handle
    rows = [0, 1, 450, 700, 898, 899]
    Vs = np.stack([h.basis_get_row(i) for i in rows])
    assert np.abs(Vs @ Vs.T - np.eye(len(rows))).max() < 1e-12
    al, be = np.diag(s.H_eff), np.diag(s.H_eff, 1)
    j = 700
    res = A @ Vs[3] - be[j - 1] * h.basis_get_row(j - 1) - al[j] * Vs[3] - be[j] * h.basis_get_row(j + 1)
    assert np.abs(res).max() < 1e-14 and np.isfinite(s.H_eff).all()


@pytest.mark.parametrize("dims,n", [((400, 300), 4200), ((300, 200), 5200)])
def test_thousands_of_basis_rows(dims, n):
    """The default pass-1 kernel parks four coefficient runs of n doubles in LDS: at n = 4200 the slice of w shrinks to
    make room, at n = 5200 the plan falls back to the VALU kernel (partials in HBM)."""
    A = synthetic.laplacian_2d_5pt(*dims).to_scipy()
    Lanczos.verbose = False
    s = Lanczos(A)
    s.execute_Lanczos(n)
    h = s._handle
    rows = [0, 7, n // 2, n - 2, n - 1]
    Vs = np.stack([h.basis_get_row(i) for i in rows])
    assert np.abs(Vs @ Vs.T - np.eye(len(rows))).max() < 1e-12 and np.isfinite(s.H_eff).all()


def test_breakdown_status_through_the_c_abi(hip):
    """lz_run returns the distinct positive status LZ_WARN_BREAKDOWN (and says which coefficient) when a residual norm
    underflows; the coefficients are still delivered as the reference would leave them (Lanczos.py:113 divides blindly)."""
    import ctypes as C

    M = 96
    ptr = np.arange(M + 1, dtype=np.int32)
    h = hip.Handle(0)
    h.set_csr(M, 0, ptr, ptr[:-1], np.ones(M))  # H = I: the first residual is exactly zero
    v0 = np.ones(M) / np.sqrt(M)
    alpha, beta = np.zeros(5), np.zeros(4)
    st = h.lib.lz_run(h._h, 5, hip.dptr(v0), hip.dptr(alpha), hip.dptr(beta))
    assert st == hip.LZ_WARN_BREAKDOWN == 1
    assert b"breakdown" in h.lib.lz_last_error(h._h)
    assert not (np.isfinite(beta).all() and beta.min() > 64 * np.finfo(float).eps)  # noise-level or non-finite residual norm
    # a healthy run on the same handle reports LZ_OK again
    A = synthetic.laplacian_2d_5pt(12, 8)
    h.set_csr(96, 0, A.rowptr, A.colidx, A.vals)
    a, b = h.run(5, synthetic.reference_start_vector(96) / np.linalg.norm(synthetic.reference_start_vector(96)))
    assert not h.breakdown and np.isfinite(a).all() and np.isfinite(b).all()
    h.close()


def test_ablation_knobs_are_rejected_by_the_product_library(hip, kb):
    h = hip.Handle(0)
    # timing-only ablations (1 >= 20, 3, 9 >= 10: deleted from BOTH builds in round 5), out-of-range knobs, and the A/B arms retired in round 3: the one-kernel /
    # one-launch-per-step engines (15 = 2, 3, 5), the persistent / LDS-staged / forced S-stationary Ritz GEMM arms (9 >= 2),
    # the ticket / deferred-fold two-sided links (11 >= 2)
    for idx, val in [(1, 21), (1, 27), (1, 31), (1, 37), (3, 1), (3, 15), (24, 0), (-1, 0), (0, -5),
                     (15, 2), (15, 3), (15, 5), (9, 2), (9, 3), (9, 4), (9, 5), (9, 6), (9, 21), (11, 2), (11, 3), (22, 2), (22, 5)]:
        assert h.lib.lz_set_tuning(h._h, idx, val) == -1, (idx, val)  # LZ_ERR_ARG
    assert b"lz_set_tuning" in h.lib.lz_last_error(h._h)
    for idx, val in [(1, 0), (1, 10), (1, 13), (8, 2), (7, 8), (13, 1), (13, 0), (15, 0), (15, 1), (9, 0), (9, 1), (11, 0), (11, 1), (16, 4096), (16, 0)]:  # (9, 6) is retired too
        assert h.lib.lz_set_tuning(h._h, idx, val) == 0, (idx, val)
    assert h.lib.lz_set_options(h._h, hip.FLAG_QTW_MFMA) == -1 and b"retired" in h.lib.lz_last_error(h._h)  # the 16x16x4 Q^T w arm
    assert h.lib.lz_set_options(h._h, hip.FLAG_QTW_VALU | hip.FLAG_FUSED_NORM) == 0  # (the VALU kernel is the fallback for > 5000 basis rows)
    h.close()
    k = kb.Handle(0)  # the kernel-bench build still takes the retired (bit-identity-tested) arms ...
    for idx, val in [(15, 2), (15, 5), (9, 3), (11, 3), (22, 4)]:
        assert k.lib.lz_set_tuning(k._h, idx, val) == 0, (idx, val)
    for idx, val in [(1, 21), (3, 1), (9, 21), (9, 31)]:  # ... but the timing-only ablation arms are gone from it too
        assert k.lib.lz_set_tuning(k._h, idx, val) == -1 and b"removed in round 5" in k.lib.lz_last_error(k._h), (idx, val)
    assert k.lib.lz_set_options(k._h, hip.FLAG_QTW_MFMA) == 0
    k.close()


def _ritz_case(H_handle, dims, n):
    A = synthetic.laplacian_2d_5pt(*dims)
    M = A.shape[0]
    v0 = synthetic.reference_start_vector(M)
    v0 /= np.linalg.norm(v0)
    h = H_handle
    h.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
    a, b = h.run(n, v0)
    V = h.get_basis()
    T = np.diag(a) + np.diag(b, 1) + np.diag(b, -1)
    S = np.linalg.eigh(T)[1]
    return V, S


@pytest.mark.parametrize("variant", [2, 3, 4, 6])
@pytest.mark.parametrize("dims,n", [((300, 250), 50), ((512, 256), 200), ((331, 211), 197), ((300, 250), 100), ((331, 211), 37)])
def test_retired_ritz_gemm_arms_in_the_kernel_bench_build(kb, variant, dims, n):
    """2: the 32-row tile walked by persistent waves; 3: two waves per SIMD with 16-row tiles, S staged through LDS; 4: one wave
    per SIMD, 32-row tiles, S through LDS; 6 (n <= 128): the 16-row-tile form of the S-in-LDS kernel, superseded by the 32-row
    form - all measured slower than the defaults, kept correct."""
    h = kb.Handle(0)
    h.set_tuning(_capi.TUNE_RITZ_KERNEL, variant)
    V, S = _ritz_case(h, dims, n)
    np.testing.assert_allclose(h.ritz_vectors(S), V.T @ S, rtol=0, atol=1e-13)
    h.close()


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("dims,n", [((300, 250), 50), ((331, 211), 37), ((512, 256), 200), ((331, 211), 197), ((331, 211), 193),
                                    ((300, 250), 49), ((300, 250), 64), ((300, 250), 65), ((300, 250), 100), ((331, 211), 111),
                                    ((300, 250), 150), ((331, 211), 192), ((300, 250), 201), ((300, 250), 126), ((257, 256), 178),
                                    ((300, 250), 80), ((300, 250), 85), ((300, 250), 96), ((300, 250), 117), ((300, 250), 133),
                                    ((300, 250), 141), ((300, 250), 158), ((300, 250), 165), ((300, 250), 173),
                                    ((300, 250), 3), ((300, 250), 16), ((300, 250), 22), ((300, 250), 30), ((331, 211), 44), ((300, 250), 58),
                                    ((300, 250), 70), ((300, 250), 128), ((300, 250), 129)])
def test_ritz_backtransform_kernels(hip, variant, dims, n):
    """Y = V S (Lanczos.py:153-156) by the FP64-MFMA kernels - 0: automatic choice: S resident in LDS, Y-stationary waves
    without barriers for 32 < n <= 128; the S-stationary kernel (S held in registers, 16-row tiles of V through LDS) for
    129 <= n <= 200 - every (column tiles, k-steps) instantiation of either is hit by some case here - else (n > 200) the
    one-workgroup-per-128-rows kernel; 1: the latter always (one wave per SIMD with a 32-row x n tile) - against NumPy on the
    fetched basis, ragged row and column counts included."""
    A = synthetic.laplacian_2d_5pt(*dims)
    M = A.shape[0]
    v0 = synthetic.reference_start_vector(M)
    v0 /= np.linalg.norm(v0)
    h = hip.Handle(0)
    h.set_tuning(_capi.TUNE_RITZ_KERNEL, variant)
    h.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
    a, b = h.run(n, v0)
    V = h.get_basis()
    T = np.diag(a) + np.diag(b, 1) + np.diag(b, -1)
    S = np.linalg.eigh(T)[1]
    Y = h.ritz_vectors(S)
    np.testing.assert_allclose(Y, V.T @ S, rtol=0, atol=1e-13)
    G = h.ritz_gram()
    assert np.abs(G - np.eye(n)).max() < 1e-12
    info = h.ritz_info()
    assert info["chunk_rows"] == 0
    if variant == 0 and 32 < n <= 200:  # the S-in-LDS (n <= 128) or the S-stationary kernel ran and left its clock record
        # (cycles per tile = workgroup 0's span / the tiles of its first wave: with only a handful of tiles per wave, as here,
        # the other waves of the workgroup may own one tile less, so the figure can dip a few per cent under the floor)
        assert info["tiles"] > 0 and 500 < info["clock_mhz"] < 3000 and info["cycles_per_tile"] >= 0.85 * info["mfma_floor_cycles_per_tile"] > 0, info
    else:
        assert info["tiles"] == 0
    # the same vectors re-formed in row chunks (what BASELINE C4 needs on one GPU: no room for a second M x n array)
    h.set_tuning(_capi.TUNE_RITZ_CHUNK_ROWS, 20000)
    h.ritz_vectors(S, fetch=False)
    assert h.ritz_info()["chunk_rows"] == 20000
    np.testing.assert_allclose(h.ritz_fetch(), Y, rtol=0, atol=1e-13)
    lo, hi = 19993, 40011  # a window across a chunk boundary, not tile aligned
    np.testing.assert_allclose(h.ritz_fetch_rows(lo, hi), Y[lo:hi], rtol=0, atol=1e-13)
    Gc = h.ritz_gram()
    assert np.abs(Gc - G).max() < 1e-13
    q_chunked = h.ritz_quality()
    h.set_tuning(_capi.TUNE_RITZ_CHUNK_ROWS, 0)
    h.ritz_vectors(S, fetch=False)
    assert np.abs(q_chunked - h.ritz_quality()).max() <= 1e-12
    h.close()


def test_matrix_stays_on_the_device_across_calls_and_use_cuda_false_runs_the_hip_path(capsys):
    """(a) `execute_Lanczos(400, use_cuda=False, seed=78)` is how the reference's largest script calls the solver
    (3Ddeuteron.py:95): accepted unedited - one notice line, then the HIP path, held to the reference's NumPy output of the
    same call (golden fixture) at the north-star bar.  (b) A second call on the same object repacks / uploads nothing while
    H's content hash is unchanged (C3 paid ~1 GB H2D + a 2.6 GB layout build per call); an in-place edit of H is seen."""
    d, H = load_golden("deuteron3d_N12_27pt_n100")
    n = int(d["n"])
    Lanczos.verbose = False
    s = Lanczos(H)
    s.execute_Lanczos(n, use_cuda=False, seed=int(d["seed"]))
    assert "use_cuda=False: lanczos_amd has no NumPy path; running the HIP path" in capsys.readouterr().out
    prefix, mask = oracle.stable_masks(H, n, d["alpha"], d["beta"], seed=int(d["seed"]))
    scale = max(np.abs(d["alpha"]).max(), np.abs(d["beta"]).max())
    assert np.abs(np.diag(s.H_eff) - d["alpha"])[:prefix].max() <= 1e-10 * scale
    conv = oracle.converged_ritz(d["alpha"], d["beta"])
    assert np.abs(s.H_eigvals[None, :] - conv[:, None]).min(axis=1).max() <= RTOL * np.abs(d["H_eigvals"]).max()
    assert s._handle.matrix_uploads == 1
    H_eff1 = s.H_eff.copy()
    s.execute_Lanczos(n, seed=int(d["seed"]))  # self.H is now the CSR the first call published (Lanczos.py:137): same arrays, same hash
    assert s._handle.matrix_uploads == 1 and np.array_equal(H_eff1, s.H_eff)
    s.H.data[3] += 0.5  # edited in place: the hash moves, the matrix goes up again
    s.execute_Lanczos(12, seed=1)
    assert s._handle.matrix_uploads == 2 and not np.array_equal(s.H_eff, H_eff1[:12, :12])
    s.strict_use_cuda = True
    with pytest.raises(NotImplementedError, match="device path only"):
        s.execute_Lanczos(12, use_cuda=False)
    # a dense ndarray stays the resident dense operator although self.H becomes a CSR copy after the first call
    A = synthetic.dense_symmetric(512, seed=0)
    t = Lanczos(A)
    t.execute_Lanczos(20)
    He = t.H_eff.copy()
    assert t._handle.spmv_plan() == "dense" and t._handle.matrix_uploads == 1
    t.execute_Lanczos(20)
    assert t._handle.spmv_plan() == "dense" and t._handle.matrix_uploads == 1 and np.array_equal(He, t.H_eff)
    t.cache_matrix = False
    t.execute_Lanczos(20)
    assert t._handle.matrix_uploads == 2


def test_one_reduce_cancellation_guard(hip):
    """LZ_FLAG_ONE_REDUCE forms |r|^2 = r''.r'' - 2 alpha u.r'' + alpha^2 u.u, whose relative error grows like
    eps (alpha^2 + beta^2) / beta^2.  On a strongly shifted operator (|alpha| ~ 1e4 >> beta ~ 2: the three sums
    cancel to 1e-8 of their size) the guard in k_onereduce_prepare fires and lz_run repeats the solve on the default loop:
    no NaN, no silent loss of the 1e-10 bar - and lz_last_engine says so.  Unshifted: the one-reduce loop itself runs."""
    import scipy.sparse

    A = synthetic.laplacian_2d_5pt(120, 100).to_scipy()
    M = A.shape[0]
    v0 = synthetic.reference_start_vector(M)
    v0 /= np.linalg.norm(v0)
    for shift, engine in ((0.0, "one-reduce"), (1.0e4, "one-reduce-repeated")):
        H = (A + shift * scipy.sparse.identity(M, format="csr")).tocsr()
        H.sort_indices()
        out = []
        for flags in (hip.FLAG_FUSED_NORM, hip.FLAG_FUSED_NORM | hip.FLAG_ONE_REDUCE):
            h = hip.Handle(0)
            h.set_options(flags)
            h.set_csr(M, 0, H.indptr, H.indices, H.data)
            a, b = h.run(40, v0)
            out.append((a, b, h.last_engine(), h.breakdown))
            h.close()
        (a0, b0, e0, bd0), (a1, b1, e1, bd1) = out
        assert e1 == engine and e0 in ("three-term-fused", "fused") and not bd0 and not bd1
        assert np.isfinite(a1).all() and np.isfinite(b1).all() and b1.min() > 0
        if engine == "one-reduce-repeated":
            assert np.array_equal(a0, a1) and np.array_equal(b0, b1)  # the repeat IS the default loop
        else:
            assert np.abs(a1 - a0).max() < 1e-11 and np.abs(b1 - b0).max() < 1e-11


@pytest.mark.parametrize("build,n,well", [(lambda: synthetic.laplacian_3d_7pt(20, 18, 16), 150, True),        # Ritz values converge: several sweeps
                                          (lambda: synthetic.laplacian_2d_5pt(96, 80), 60, True),
                                          (lambda: synthetic.laplacian_2d_5pt(331, 211), 90, True),          # rows no multiple of a slice
                                          (lambda: synthetic.random_graph_laplacian(50000, 175000, seed=3), 150, False)])
def test_partial_one_reduce_loop(hip, build, n, well):
    """LZ_FLAG_REORTH_PARTIAL | LZ_FLAG_ONE_REDUCE (round 5, lz_last_engine 8): the device-decided partial re-orthogonalisation with
    ONE all-reduce per step.  The sweep decision is a one-step look-ahead of Simon's recurrence (its exact test
    needs the sums the step's only all-reduce delivers), so sweeps may come a step earlier than in the three-collective loop; held to:
    coefficients of a well-conditioned run within 1e-10 of the scale of the default (full-sweep) loop's, converged Ritz values to
    1e-10 whatever the conditioning, a semi-orthogonal basis (sqrt(eps) level), no look-ahead miss, no host synchronisation, and a
    sweep count in the neighbourhood of the exact loop's."""
    A = build()
    M = A.shape[0]
    v0 = synthetic.reference_start_vector(M)
    v0 /= np.linalg.norm(v0)
    res = {}
    for tag, flags in (("full", hip.FLAG_FUSED_NORM), ("partial", hip.FLAG_REORTH_PARTIAL), ("onered", hip.FLAG_REORTH_PARTIAL | hip.FLAG_ONE_REDUCE)):
        h = hip.Handle(0)
        h.set_options(flags)
        h.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
        a, b = h.run(n, v0)
        res[tag] = dict(a=a, b=b, V=h.get_basis(), sweeps=h.last_sweeps(), misses=h.last_sweep_misses(), engine=h.last_engine(),
                        syncs=h.last_host_syncs(), bd=h.breakdown)
        h.close()
    f, p1, o = res["full"], res["partial"], res["onered"]
    assert o["engine"] == "partial-one-reduce" and o["syncs"] == 0 and o["misses"] == 0 and not o["bd"]
    assert p1["engine"] == "partial-device" and p1["misses"] == 0
    assert 1 <= o["sweeps"] < n and abs(o["sweeps"] - p1["sweeps"]) <= max(3, p1["sweeps"] // 2), (o["sweeps"], p1["sweeps"])
    scale = max(np.abs(f["a"]).max(), np.abs(f["b"]).max())
    th_f = np.linalg.eigvalsh(oracle.build_h_eff(f["a"], f["b"]))
    th_o = np.linalg.eigvalsh(oracle.build_h_eff(o["a"], o["b"]))
    conv = oracle.converged_ritz(f["a"], f["b"], tol=1e-9)
    if len(conv):
        assert np.abs(th_o[None, :] - conv[:, None]).min(axis=1).max() <= RTOL * np.abs(th_f).max()
    if well:
        prefix, mask = oracle.stable_masks(A.to_scipy(), n, f["a"], f["b"], v0=v0)
        assert prefix >= n // 2
        assert np.abs(o["a"] - f["a"])[:prefix].max() <= 1e-10 * scale and np.abs(o["b"] - f["b"])[: prefix - 1].max() <= 1e-10 * scale
        if prefix == n:
            assert np.abs(th_o - th_f).max() <= RTOL * np.abs(th_f).max()
    assert np.abs(o["V"] @ o["V"].T - np.eye(n)).max() < 1e-6  # semi-orthogonal, like the three-collective partial loop
    # a second run on a fresh handle: bit-identical (deterministic reductions, device-side decisions)
    h = hip.Handle(0)
    h.set_options(hip.FLAG_REORTH_PARTIAL | hip.FLAG_ONE_REDUCE)
    h.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
    a2, b2 = h.run(n, v0)
    assert np.array_equal(a2, o["a"]) and np.array_equal(b2, o["b"]) and np.array_equal(h.get_basis(), o["V"]) and h.last_sweeps() == o["sweeps"]
    # knob 18 = 3: the loop's first form - second-stage sums, the alpha subtraction, the sweep and the plain scale as kernels of their
    # own (10 launches per step instead of 6) - same arithmetic, same summation trees: same bits
    h.set_tuning(hip.TUNE_PARTIAL_LOOP, 3)
    a4, b4 = h.run(n, v0)
    assert np.array_equal(a4, o["a"]) and np.array_equal(b4, o["b"]) and np.array_equal(h.get_basis(), o["V"]) and h.last_sweeps() == o["sweeps"]
    h.set_tuning(hip.TUNE_PARTIAL_LOOP, 0)
    # the look-ahead's safety factor (knob 20): a larger kappa can only sweep earlier / more often, never lose the bar
    h.set_tuning(hip.TUNE_PARTIAL_LOOKAHEAD, 64)
    a3, b3 = h.run(n, v0)
    assert h.last_sweeps() >= o["sweeps"] and h.last_sweep_misses() == 0
    if well:
        assert np.abs(a3 - f["a"])[:prefix].max() <= 1e-10 * scale
    h.close()


def test_partial_one_reduce_cancellation_guard(hip):
    """The one-reduce partial loop forms |r|^2 from three sums like the full one-reduce loop and carries the same guard: on a
    strongly shifted operator the solve is repeated - on the three-collective device loop, whose coefficients it then delivers
    bit for bit (lz_last_engine 5)."""
    import scipy.sparse

    A = synthetic.laplacian_2d_5pt(120, 100).to_scipy()
    M = A.shape[0]
    v0 = synthetic.reference_start_vector(M)
    v0 /= np.linalg.norm(v0)
    H = (A + 1.0e4 * scipy.sparse.identity(M, format="csr")).tocsr()
    H.sort_indices()
    out = []
    for flags in (hip.FLAG_REORTH_PARTIAL, hip.FLAG_REORTH_PARTIAL | hip.FLAG_ONE_REDUCE):
        h = hip.Handle(0)
        h.set_options(flags)
        h.set_csr(M, 0, H.indptr, H.indices, H.data)
        a, b = h.run(40, v0)
        out.append((a, b, h.last_engine(), h.last_sweeps(), h.breakdown))
        h.close()
    (a0, b0, e0, s0, bd0), (a1, b1, e1, s1, bd1) = out
    assert e0 == "partial-device" and e1 == "one-reduce-repeated" and not bd0 and not bd1
    assert np.array_equal(a0, a1) and np.array_equal(b0, b1) and s0 == s1


@pytest.mark.parametrize("build,n1,n2", [(lambda: synthetic.laplacian_2d_5pt(64, 48).to_scipy(), 10, 31),             # three-launch loop
                                         (lambda: synthetic.laplacian_3d_7pt(40, 30, 20).to_scipy(), 17, 40),         # five-launch loop
                                         (lambda: synthetic.random_graph_laplacian(50000, 175000, seed=3).to_scipy(), 2, 12),
                                         (lambda: load_golden("box1d_N500_n50")[1], 5, 20),                           # M = 500: padded rows
                                         (lambda: synthetic.dense_symmetric(700, seed=2), 25, 26)])
@pytest.mark.parametrize("fused", [True, False])
def test_checkpoint_and_resume_is_bit_identical(tmp_path, build, n1, n2, fused):
    """SURVEY.md section 5 hook: (alpha, beta, j, V[:j]) + the residual is a checkpoint; `resume_Lanczos` continues it.  A run of
    n1 steps, saved, loaded into a NEW object and continued to n2 steps gives H_eff, V and the Ritz values of an uninterrupted
    n2-step run bit for bit - whichever loop structure (three / five / six launches per step) the uninterrupted run used."""
    H = build()
    Lanczos.verbose = False
    whole = Lanczos(H)
    whole.fused_norm = fused
    whole.execute_Lanczos(n2)
    first = Lanczos(H)
    first.fused_norm = fused
    first.execute_Lanczos(n1)
    assert np.array_equal(np.diag(first.H_eff)[: n1 - 1], np.diag(whole.H_eff)[: n1 - 1])  # (the last alpha of a run is final too)
    path = str(tmp_path / "ck.npz")
    first.save_checkpoint(path)
    ck = first.checkpoint()
    assert ck["V"].shape == (n1, H.shape[0]) and ck["r"].shape == (H.shape[0],) and len(ck["beta"]) == n1 - 1
    first.close()
    second = Lanczos(H)
    second._get_handle().set_tuning(_capi.TUNE_POISON_BASIS, 1)  # (the uploaded rows' padding is cleared, not inherited: round 5)
    second.resume_Lanczos(n2, path)
    assert second._handle.last_engine() == "kernels"
    assert np.array_equal(second.H_eff, whole.H_eff)
    assert np.array_equal(second.V, whole.V)
    assert np.array_equal(second.H_eigvals, whole.H_eigvals)
    third = Lanczos(H)  # from the in-memory dict, and the resumed run can itself be checkpointed
    third.resume_Lanczos(n2, ck)
    assert np.array_equal(third.H_eff, whole.H_eff)
    r2 = third.checkpoint()["r"]
    assert np.array_equal(r2, whole.checkpoint()["r"])
    with pytest.raises(ValueError, match="must exceed"):
        third.resume_Lanczos(n1, ck)
    # the checkpoint's matrix key is format-independent - the same operator held in another container resumes (CSC for
    # the sparse cases: symmetric, so the same CSR arrays and the same bits; the CSR form of the dense case: another SpMV kernel,
    # same operator, coefficients to rounding), and so does `Lanczos(old.H)` (the CSR copy execute_Lanczos leaves in self.H)
    if scipy.sparse.issparse(H):
        fourth = Lanczos(H.tocsc())
        fourth.resume_Lanczos(n2, ck)
        assert np.array_equal(fourth.H_eff, whole.H_eff)
    else:
        fourth = Lanczos(scipy.sparse.csr_matrix(H))
        fourth.resume_Lanczos(n2, ck)
        assert np.abs(fourth.H_eff - whole.H_eff).max() <= 1e-10 * np.abs(whole.H_eff).max()
    fourth.close()
    fifth = Lanczos(whole.H)
    fifth.resume_Lanczos(n2, ck)
    assert np.isfinite(fifth.H_eff).all()
    fifth.close()
    # the resumed run's norm order is the checkpoint's, but the object's own setting is left alone; a checkpoint
    # is refused by a different operator of the same size and by other options; a closed object says so clearly
    assert second.fused_norm is Lanczos.fused_norm
    H_other = H + (scipy.sparse.identity(H.shape[0], format="csr") if scipy.sparse.issparse(H) else np.eye(H.shape[0]))
    other = Lanczos(H_other)
    with pytest.raises(ValueError, match="different matrix"):
        other.resume_Lanczos(n2, ck)
    other.options = _capi.FLAG_SPMV_STREAM
    with pytest.raises(ValueError, match="options"):
        other.resume_Lanczos(n2, ck)
    other.close()
    third.close()
    with pytest.raises(_capi.LanczosHipError, match="released"):
        third.checkpoint()
    for s in (whole, second):
        s.close()


@pytest.mark.parametrize("build,n1,n2", [(lambda: synthetic.laplacian_3d_7pt(40, 30, 20).to_scipy(), 37, 120),       # ELL-fused r / beta; sweeps on both sides of the cut
                                         (lambda: load_golden("deuteron3d_N12_27pt_n100")[1], 50, 100),              # 27-point rows: scale kernel + CSR stream
                                         (lambda: load_golden("box1d_N500_n50")[1], 2, 50),                          # the shortest first leg
                                         (lambda: synthetic.dense_symmetric(700, seed=2), 61, 90)])
def test_checkpoint_and_resume_of_the_partial_loop_is_bit_identical(tmp_path, hip, build, n1, n2):
    """a reorth='partial' run can be continued too.  Its checkpoint also carries the omega-recurrence state of the
    device-decided loop (lz_get_omega_state); lz_run_resume_partial re-lays it for the longer run and takes the decision of the first
    new step from the refreshed ||r||^2 exactly as the uninterrupted run did: coefficients, basis AND sweep schedule equal an
    uninterrupted run's bit for bit, wherever the cut falls relative to a sweep pair."""
    H = build()
    Lanczos.verbose = False
    whole = Lanczos(H)
    whole.reorth = "partial"
    whole.execute_Lanczos(n2)
    hw = whole._get_handle()
    assert hw.last_engine() == "partial-device"
    log_whole = hw.last_sweep_log()
    first = Lanczos(H)
    first.reorth = "partial"
    first.execute_Lanczos(n1)
    assert list(first._get_handle().last_sweep_log()) == list(log_whole[:n1])
    path = str(tmp_path / "ck.npz")
    first.save_checkpoint(path)
    ck = first.checkpoint()
    assert ck["omega_state"].shape == (2 + (n1 + 2) + 3 * (n1 + 1),) and str(ck["reorth"]) == "partial"
    first.close()
    second = Lanczos(H)
    second.reorth = "partial"
    # (NaN-poisoned fresh basis: the padding of the uploaded rows must not be whatever the recycled allocation held - it enters
    # ||r||^2 through the kernels that stream rows_pad columns; found with this test at M = 500)
    second._get_handle().set_tuning(_capi.TUNE_POISON_BASIS, 1)
    second.resume_Lanczos(n2, path)
    hs = second._get_handle()
    assert hs.last_engine() == "partial-device" and hs.last_host_syncs() == 0
    assert np.array_equal(second.H_eff, whole.H_eff)
    assert np.array_equal(second.V, whole.V)
    assert list(hs.last_sweep_log()[n1:]) == list(log_whole[n1:])  # (the record of a resumed run covers its own steps)
    assert second.sweeps == int(np.sum(log_whole[n1:]))
    # ... and the resumed run can be checkpointed and continued again: same state as the uninterrupted run's
    ck2, ckw = second.checkpoint(), whole.checkpoint()
    assert np.array_equal(ck2["r"], ckw["r"]) and np.array_equal(ck2["omega_state"], ckw["omega_state"])
    if n2 + 5 <= H.shape[0]:
        longer = Lanczos(H)
        longer.reorth = "partial"
        longer.execute_Lanczos(n2 + 5)
        third = Lanczos(H)
        third.reorth = "partial"
        third.resume_Lanczos(n2 + 5, ck2)
        assert np.array_equal(third.H_eff, longer.H_eff) and np.array_equal(third.V, longer.V)
        third.close()
        longer.close()
    # refusals: a full-sweep object, a checkpoint without the state, a state of the wrong length, the one-reduce arm
    full = Lanczos(H)
    with pytest.raises(ValueError, match="reorth='partial' run"):
        full.resume_Lanczos(n2, ck)
    full.close()
    other = Lanczos(H)
    other.reorth = "partial"
    with pytest.raises(NotImplementedError, match="omega-recurrence state"):
        other.resume_Lanczos(n2, {k: v for k, v in ck.items() if k != "omega_state"})
    with pytest.raises(ValueError, match="does not belong"):
        other.resume_Lanczos(n2, dict(ck, omega_state=ck["omega_state"][:-1]))
    other.close()
    onered = Lanczos(H)
    onered.reorth = "partial"
    onered.options = _capi.FLAG_ONE_REDUCE
    with pytest.raises(ValueError, match="options"):
        onered.resume_Lanczos(n2, ck)
    onered.close()
    # at the C boundary: the state is only there after the device-decided loop
    hf = hip.Handle(0)
    Hs = scipy.sparse.csr_matrix(H)
    hf.set_csr(Hs.shape[0], 0, Hs.indptr, Hs.indices, Hs.data)
    hf.run(5, np.ones(Hs.shape[0]) / np.sqrt(Hs.shape[0]))
    with pytest.raises(_capi.LanczosHipError, match="engine 7"):
        hf.get_omega_state()
    hf.close()
    for s in (whole, second):
        s.close()


def test_ritz_quality_of_a_dense_matrix_runs_on_the_device(capsys):
    """print_good_eigs (Lanczos.py:166-185) on the dense-ndarray call path of 1Dbox.py: the quality sums come from the device
    (every Ritz vector multiplied by the resident dense operator) - there is no NumPy fallback in the product."""
    A = synthetic.dense_symmetric(700, seed=4)
    Lanczos.verbose = False
    s = Lanczos(A)
    s.execute_Lanczos(40)
    q = s._eigvec_quality()
    Y = s.H_eigvecs
    AY = A @ Y
    ref = np.einsum("ij,ij->j", AY / np.linalg.norm(AY, axis=0), Y) ** 2
    np.testing.assert_allclose(q, ref, rtol=1e-11, atol=1e-13)
    assert np.array_equal(s._handle.get_basis(), s.V.T)  # the basis row the kernel borrows is back
    s.print_good_eigs(print_nr=3)
    assert "Eigvec InnerProd" in capsys.readouterr().out


def test_large_results_come_back_through_the_staged_copy_unchanged():
    """Result publication (Lanczos.py:132-141,153-156): copies of >= 192 MB leave the device through a ring of pinned staging
    buffers and host copy threads (lz_xfer.hip).  The same data fetched in small pieces (row by row / window by window: the
    runtime's plain path) must be the same bytes - whole basis with a padded destination stride, an odd window of it, the
    Ritz vectors whole and as a row range."""
    from lanczos_amd import _capi

    M, n = 1_300_003, 24  # 250 MB per array; rows not a multiple of anything
    d = np.linspace(1.0, 2.0, M)
    ptr = np.arange(M + 1, dtype=np.int32)
    h = _capi.Handle(0)
    h.set_options(_capi.FLAG_FUSED_NORM)
    h.set_csr(M, 0, ptr, ptr[:-1], d)
    v0 = np.random.default_rng(3).standard_normal(M)
    a, b = h.run(n, v0 / np.linalg.norm(v0))
    V = h.get_basis()
    for j in (0, 7, n - 1):
        assert np.array_equal(V[j], h.basis_get_row(j))
    ld = M + 5
    Vp = np.full((n, ld), -7.0)
    h.check(h.lib.lz_get_basis(h._h, _capi.dptr(Vp), ld))
    assert np.array_equal(Vp[:, :M], V) and (Vp[:, M:] == -7.0).all()  # the padding of the caller's array is not touched
    blk = h.get_basis_block(11, M - 3)  # 24 rows x 10.4 MB: the 2-D staging path
    assert np.array_equal(blk, V[:, 11:M - 3])
    S = np.linalg.eigh(np.diag(a) + np.diag(b, 1) + np.diag(b, -1))[1]
    Y = h.ritz_vectors(S)
    assert np.abs(Y - V.T @ S).max() < 1e-13
    assert np.array_equal(h.ritz_fetch(), Y)
    lo, hi = 100_001, M - 77
    big = h.ritz_fetch_rows(lo, hi)
    assert np.array_equal(big, Y[lo:hi])
    for r0 in (lo, 600_000, hi - 1000):  # small windows: plain path
        assert np.array_equal(h.ritz_fetch_rows(r0, r0 + 1000), Y[r0:r0 + 1000])
    h.close()


def test_big_buffers_survive_allocation_churn(hip):
    """Round 5 (profiles/r05/vmm_address_reuse_fault.txt): on this stack a fresh VMM mapping that lands on just-freed addresses
    intermittently had holes - kernels faulted inside their own buffers.  `big_free` now frees a range and at once reserves the same
    addresses again, unmapped, so that no later buffer lands on them.  Handles with VMM-sized buffers of changing sizes (a 330 - 650 MB
    basis, Ritz vectors) are created, run and destroyed in turn while another one stays alive; every run of one size must give the
    bits of the first run of that size, and the memory must be back at the end."""
    probe = hip.Handle(0)
    free0, _ = probe.device_memory()
    A = synthetic.laplacian_2d_5pt(1300, 800)  # M = 1.04e6: 8.3 MB per basis row
    M = A.shape[0]
    v0 = synthetic.reference_start_vector(M)
    v0 /= np.linalg.norm(v0)
    keeper = hip.Handle(0)
    keeper.set_options(hip.FLAG_FUSED_NORM)
    keeper.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
    first = {}
    for cyc, n in enumerate((40, 78, 52, 64, 40, 78, 45, 52, 64, 70, 40, 78)):
        h = hip.Handle(0)
        h.set_options(hip.FLAG_FUSED_NORM)
        h.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
        a, b = h.run(n, v0)
        S = np.linalg.eigh(np.diag(a) + np.diag(b, 1) + np.diag(b, -1))[1]
        h.ritz_vectors(S, fetch=False)
        G = h.ritz_gram()
        assert np.abs(G - np.eye(n)).max() < 1e-10
        if n in first:
            assert np.array_equal(a, first[n][0]) and np.array_equal(b, first[n][1]), (cyc, n)
        else:
            first[n] = (a, b)
        if cyc % 3 == 0:  # the long-lived handle allocates and drops big buffers in between
            ka, kb = keeper.run(48 + cyc, v0)
            assert np.array_equal(ka[:39], first[40][0][:39])  # (a longer run's leading coefficients are the shorter run's)
        h.close()
    keeper.close()
    free1, _ = probe.device_memory()
    probe.close()
    assert abs(free1 - free0) < 96 << 20, (free0, free1)


def test_closing_a_handle_gives_the_device_memory_back(hip):
    """Round 5: buffers of 256 MB and more are VMM ranges (reserve + create + map), everything else hipMalloc; `lz_destroy` must
    release both kinds - a plain hipFree of a mapped range fails silently and would leak it.  Several handles with a 400 MB basis,
    400 MB Ritz vectors, a 300 MB dense matrix and 320 MB work vectors are created, run and closed; the device's free memory comes
    back to where it started."""
    probe = hip.Handle(0)
    free0, total = probe.device_memory()
    rng = np.random.default_rng(0)
    for round_ in range(3):
        # (a) CSR problem with big basis / Ritz vectors (M = 2e6, n = 25: 400 MB each)
        A = synthetic.laplacian_2d_5pt(2000, 1000)
        M = A.shape[0]
        v0 = synthetic.reference_start_vector(M)
        v0 /= np.linalg.norm(v0)
        h = hip.Handle(0)
        h.set_options(hip.FLAG_FUSED_NORM)
        h.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
        a, b = h.run(25, v0)
        S = np.linalg.eigh(np.diag(a) + np.diag(b, 1) + np.diag(b, -1))[1]
        h.ritz_vectors(S, fetch=False)
        h.ritz_gram()
        used = free0 - h.device_memory()[0]
        assert used > 700 << 20  # the big buffers really are on the device now
        h.close()
        # (b) a dense matrix of 300 MB
        D = synthetic.dense_symmetric(6200, seed=round_)
        h = hip.Handle(0)
        h.set_dense(D)
        h.run(10, rng.standard_normal(6200) / 80.0)
        h.close()
    free1, _ = probe.device_memory()
    probe.close()
    assert abs(free1 - free0) < 96 << 20, (free0, free1)
