"""Kernel-level parity on the GPU, through the C ABI (lz_step_* entry points).

Element-wise kernels (SpMV row sums, re-orthogonalisation update, three-term
recurrence, r/beta scaling) are compared BIT-EXACT against NumPy/SciPy - they
follow the reference's expression order with no FMA.  Reductions (alpha, c,
||r||^2) are compared to 1e-13 relative to the sum of magnitudes.
"""
import numpy as np
import pytest
import scipy.sparse

from lanczos_amd import synthetic, _capi

pytestmark = pytest.mark.gpu


def _matrices():
    rng = np.random.default_rng(5)
    R = scipy.sparse.random(3000, 3000, density=0.004, random_state=rng, format="csr")
    R = (R + R.T + scipy.sparse.diags(rng.standard_normal(3000))).tocsr()
    R.sort_indices()
    L = R.tolil()
    L[17, :] = rng.standard_normal(3000)  # a 3000-entry row: exercises the short-row tile limit
    L[:, 17] = L[17, :].T
    wide = scipy.sparse.random(9000, 9000, density=0.0012, random_state=rng, format="csr")
    wide = (wide + wide.T).tolil()
    wide[100, :] = 1.0  # > 4096 entries: long-row path
    wide[:, 100] = 1.0
    return {
        "lap2d_37x29": synthetic.laplacian_2d_5pt(37, 29).to_scipy(),
        "lap3d_9x8x7": synthetic.laplacian_3d_7pt(9, 8, 7).to_scipy(),
        "graph_5000": synthetic.random_graph_laplacian(5000, 17000, seed=3).to_scipy(),
        "ragged_3000": L.tocsr(),
        "longrow_9000": wide.tocsr(),
        "empty_rows": scipy.sparse.csr_matrix(([1.0, 2.0, 2.0, 5.0], ([0, 1, 3, 3], [0, 3, 1, 3])), shape=(6, 6)),
        "lap2d_400x300": synthetic.laplacian_2d_5pt(400, 300).to_scipy(),
    }


MATS = _matrices()


@pytest.mark.parametrize("name", list(MATS))
@pytest.mark.parametrize("flags", [0, 8, 32])
def test_spmv_bit_exact(hip, name, flags):
    H = MATS[name]
    M = H.shape[0]
    h = hip.Handle(0)
    h.set_options(flags)
    h.set_csr(M, 0, H.indptr, H.indices, H.data)
    x = np.random.default_rng(1).uniform(-1, 1, M)
    y = h.spmv_host(x)
    ref = H * x
    # rows longer than the CSR-stream tile (2048 entries when rows average >= 12 entries, else 4096) get a block of
    # their own with a block-wide (not sequential) reduction: tolerance instead of bit-exactness for those rows
    cap = 2048 if H.nnz / M >= 12 else 4096
    long_rows = np.diff(H.indptr) > cap
    if long_rows.any() and flags in (0, 32):
        assert long_rows.sum() == 1
        assert np.array_equal(y[~long_rows], ref[~long_rows])
        i = int(np.flatnonzero(long_rows)[0])
        np.testing.assert_allclose(y[long_rows], ref[long_rows], rtol=0, atol=1e-13 * np.abs(H[i]).dot(np.abs(x)).max())
    else:
        assert np.array_equal(y, ref), f"max diff {np.abs(y - ref).max()}"
    h.close()


def _fixed_k_random(M, K, seed):
    """M rows of exactly K entries at random (sorted, distinct) columns with real values: fixed-K without any stencil structure"""
    rng = np.random.default_rng(seed)
    W = min(M, 256)  # K distinct offsets inside a window of W columns placed anywhere in the row
    offs = np.argsort(rng.random((M, W)), axis=1)[:, :K]
    cols = np.sort(rng.integers(0, M - W + 1, (M, 1)) + offs, axis=1)
    return scipy.sparse.csr_matrix((rng.standard_normal(M * K), cols.ravel().astype(np.int32), np.arange(0, M * K + 1, K, dtype=np.int32)), shape=(M, M))


@pytest.mark.parametrize("K,M", [(5, 1300), (7, 2049), (27, 777), (5, 40000), (7, 30011), (27, 5000)])
def test_spmv_fixed_k_layouts_bit_exact(hip, K, M):
    """Fixed-K rows: the ELL-ordered copy (round 4; knob 17 = 2 one row per lane, 3 two adjacent rows per lane; by default only the
    partial re-orthogonalisation loop's fused SpMV uses it) against the CSR-order kernel (knob 17 = 0 / 1) and SciPy - y bit for bit in every layout, ragged last block included; the
    one-row-per-lane layout also groups the alpha partials like the CSR-order kernel (same bits, K in {5, 7})."""
    H = _fixed_k_random(M, K, seed=K * M)
    x = np.random.default_rng(1).uniform(-1, 1, M)
    ref = H * x
    alphas = {}
    for knob in (0, 1, 2, 3):
        h = hip.Handle(0)
        h.set_tuning(hip.TUNE_FIXED_LAYOUT, knob)
        h.set_csr(M, 0, H.indptr, H.indices, H.data)
        y = h.spmv_host(x)
        assert np.array_equal(y, ref), (knob, np.abs(y - ref).max())
        h.basis_alloc(3)
        h.basis_set_row(1, x)
        alphas[knob] = h.step_spmv(1)
        assert np.array_equal(h.r_get(), ref)
        assert abs(alphas[knob] - np.dot(x, ref)) <= 1e-13 * np.dot(np.abs(x), np.abs(H) * np.abs(x))
        h.close()
    if K == 27:
        assert alphas[0] == alphas[2]  # 27-point rows: auto = ELL (no CSR-order fixed-K kernel exists for them; ELL beats CSR-stream)
    else:
        assert alphas[0] == alphas[1]  # auto = the CSR-order kernel; the ELL copy serves the partial loop's fused SpMV only
        assert alphas[1] == alphas[2]  # 512-row blocks, rows t and t + 256 per lane in both


def _stencil_cases():
    lap2 = synthetic.laplacian_2d_5pt(37, 29).to_scipy()       # 9 classes (periodic: interior, 4 edges, 4 corners), ragged last block
    lap2b = synthetic.laplacian_2d_5pt(400, 300).to_scipy()
    lap3 = synthetic.laplacian_3d_7pt(21, 17, 13).to_scipy()   # 27 classes
    rng = np.random.default_rng(5)

    def with_values(H, what):
        G = H.copy()
        if what == "random":
            G.data = rng.standard_normal(G.nnz)
        else:  # a potential on the diagonal: the reference's own operators (Hamiltonian.py) - offsets and off-diagonal values repeat, the diagonal streams
            G = (G + scipy.sparse.diags(rng.uniform(-1, 1, G.shape[0]))).tocsr()
        G.sort_indices()
        return G

    return {"lap2d_37x29": (lap2, "offsets+values", 9), "lap2d_400x300": (lap2b, "offsets+values", 9), "lap3d_21x17x13": (lap3, "offsets+values", 27),
            "lap2d_37x29_random_values": (with_values(lap2, "random"), "offsets", 9),
            "lap3d_21x17x13_potential": (with_values(lap3, "potential"), "offsets+values, diagonal streamed", 27),
            "lap2d_400x300_potential": (with_values(lap2b, "potential"), "offsets+values, diagonal streamed", 9)}


@pytest.mark.parametrize("name", ["lap2d_37x29", "lap2d_400x300", "lap3d_21x17x13", "lap2d_37x29_random_values", "lap3d_21x17x13_potential",
                                  "lap2d_400x300_potential"])
def test_spmv_row_class_coding_bit_exact(hip, name):
    """Round 5: a stencil matrix's rows fall into a handful of classes up to translation (offsets col - row; with constant coefficients
    the values too).  lz_set_csr finds and verifies them on the device and the ELL-order kernel then streams ONE BYTE per row (+ the
    values when only the offsets repeat) instead of 12 bytes per entry.  y and the alpha partials are bit-identical to the CSR-order
    kernel (knob 17 = 1), the uncoded ELL copy (2) and SciPy; knob 4 keeps the values streamed (A/B)."""
    H, coding, classes = _stencil_cases()[name]
    M = H.shape[0]
    x = np.random.default_rng(1).uniform(-1, 1, M)
    ref = H * x
    alphas = {}
    for knob in (0, 1, 2, 4, 10, 30):  # 10 / 30: auto with knob 23 = 1 / 3 (the one-row-per-lane forms of the fully coded kernel, A/B)
        h = hip.Handle(0)
        h.set_tuning(hip.TUNE_FIXED_LAYOUT, knob if knob < 10 else 0)
        h.set_tuning(hip.TUNE_CLS_GROUP, knob // 10)
        h.set_csr(M, 0, H.indptr, H.indices, H.data)
        want = {0: (coding, classes), 1: ("none", 0), 2: ("none", 0), 4: ("offsets", classes), 10: (coding, classes), 30: (coding, classes)}[knob]
        assert h.spmv_coding() == want, (knob, h.spmv_coding())
        assert h.spmv_plan() == "fixed-k"
        for _ in range(2):
            y = h.spmv_host(x)
            assert np.array_equal(y, ref), (knob, np.abs(y - ref).max())
        h.basis_alloc(3)
        h.basis_set_row(1, x)
        alphas[knob] = h.step_spmv(1)
        assert np.array_equal(h.r_get(), ref)
        h.close()
    assert alphas[0] == alphas[1] == alphas[2] == alphas[4] == alphas[10] == alphas[30]


def test_spmv_row_class_coding_gives_way(hip):
    """More than 256 classes (fixed-K rows without stencil structure; a stencil whose every row has its own coefficients still codes its
    OFFSETS), a value class count just over the limit, and options that bypass the ELL copy: the uncoded kernels run, same bits."""
    H = _fixed_k_random(3000, 5, seed=3)
    h = hip.Handle(0)
    h.set_csr(3000, 0, H.indptr, H.indices, H.data)
    assert h.spmv_coding() == ("none", 0)
    x = np.random.default_rng(2).uniform(-1, 1, 3000)
    assert np.array_equal(h.spmv_host(x), H * x)
    h.close()
    # 300 different diagonal values on a 5-point stencil: 9 x 300 > 256 classes of offsets + values -> the diagonal streams (9 classes)
    L = synthetic.laplacian_2d_5pt(60, 50).to_scipy()
    D = scipy.sparse.diags((np.arange(3000) % 300).astype(float))
    G = (L + D).tocsr()
    G.sort_indices()
    h = hip.Handle(0)
    h.set_csr(3000, 0, G.indptr, G.indices, G.data)
    assert h.spmv_coding() == ("offsets+values, diagonal streamed", 9)
    assert np.array_equal(h.spmv_host(x), G * x)
    h.set_options(_capi.FLAG_SPMV_STREAM)  # the CSR-stream kernel on request: the coding is not in play
    assert h.spmv_coding() == ("none", 0) and h.spmv_plan() == "csr-stream"
    assert np.array_equal(h.spmv_host(x), G * x)
    h.close()
    # 25 diagonal values: 9 x 25 = 225 classes of offsets + values (a table of 225 x 5 entries in LDS)
    G = (L + scipy.sparse.diags((np.arange(3000) % 25).astype(float))).tocsr()
    G.sort_indices()
    h = hip.Handle(0)
    h.set_csr(3000, 0, G.indptr, G.indices, G.data)
    kind, ncls = h.spmv_coding()
    assert kind == "offsets+values" and 25 <= ncls <= 225
    assert np.array_equal(h.spmv_host(x), G * x)
    h.close()
    # random OFF-diagonal values as well: only the offsets repeat
    R = L.copy().tocsr()
    R.data = np.random.default_rng(7).standard_normal(R.nnz)
    h = hip.Handle(0)
    h.set_csr(3000, 0, R.indptr, R.indices, R.data)
    assert h.spmv_coding() == ("offsets", 9)
    assert np.array_equal(h.spmv_host(x), R * x)
    h.close()
    # an explicit second entry on the diagonal (duplicate column) in a fixed-K matrix: the diagonal-streamed coding is refused
    rows = np.repeat(np.arange(3000), 5)
    cols = np.stack([np.arange(3000), np.arange(3000), (np.arange(3000) + 1) % 3000, (np.arange(3000) + 2) % 3000, (np.arange(3000) + 3) % 3000], axis=1).ravel()
    vals = np.tile([1.0, 1.0, -1.0, -1.0, -1.0], 3000) + np.repeat(np.arange(3000) * 1e-3, 5) * (np.tile([1, 0, 0, 0, 0], 3000))
    h = hip.Handle(0)
    h.set_csr(3000, 0, np.arange(0, 15001, 5, dtype=np.int32), cols.astype(np.int32), vals)
    assert h.spmv_coding()[0] in ("offsets", "none")
    ref = np.zeros(3000)
    np.add.at(ref, rows, vals * x[cols])  # (unordered accumulation: compare to rounding)
    assert np.abs(h.spmv_host(x) - ref).max() <= 1e-12
    h.close()


def _two_phase_matrices():
    rng = np.random.default_rng(11)
    big = synthetic.random_graph_laplacian(60000, 210000, seed=8).to_scipy()  # several row blocks x 118 column blocks
    vals = big.copy()
    vals.data = rng.standard_normal(big.nnz)  # non-integer values: products and partial sums really round
    too_long = scipy.sparse.random(40000, 40000, density=0.0002, random_state=rng, format="lil")
    too_long[7, :] = 1.0  # 40000 entries > the largest LDS tile: the layout does not apply, CSR-stream runs instead
    short = synthetic.random_graph_laplacian(50000, 40000, seed=3).to_scipy()  # 2.6 entries per row: row blocks end at their 2048-row limit, not at the LDS tile
    return {"graph_60000": big, "graph_60000_real": vals.tocsr(), "row_too_long": too_long.tocsr(), "short_rows_50000": short}


@pytest.mark.parametrize("name", ["graph_5000", "ragged_3000", "longrow_9000", "empty_rows", "lap2d_37x29", "lap2d_400x300",
                                  "graph_60000", "graph_60000_real", "row_too_long", "short_rows_50000"])
def test_spmv_two_phase_bit_exact(hip, name):
    """The column-blocked two-phase SpMV (lz_spmv_pb.hip; auto-selected for matrices without column locality, forced here
    with tuning knob 14 = 2): products staged through a column-block-major buffer, every row summed out of LDS in CSR
    order - bit-identical to SciPy's csr_matvec, including rows of thousands of entries (up to the ~13 000-product LDS tile)."""
    H = MATS[name] if name in MATS else _two_phase_matrices()[name]
    M = H.shape[0]
    h = hip.Handle(0)
    h.set_tuning(_capi.TUNE_SPMV_PLAN, 2)
    h.set_csr(M, 0, H.indptr, H.indices, H.data)
    assert h.spmv_plan() == ("csr-stream" if name == "row_too_long" else "two-phase")
    x = np.random.default_rng(1).uniform(-1, 1, M)
    ref = H * x
    for _ in range(2):  # the layout's scratch buffer is reused between calls
        y = h.spmv_host(x)
        if name == "row_too_long":
            long_rows = np.diff(H.indptr) > 4096
            assert np.array_equal(y[~long_rows], ref[~long_rows])
            np.testing.assert_allclose(y[long_rows], ref[long_rows], rtol=0, atol=1e-12)
        else:
            assert np.array_equal(y, ref), f"max diff {np.abs(y - ref).max()}"
    # fused alpha partials
    h.basis_alloc(3)
    v = np.random.default_rng(2).standard_normal(M)
    h.basis_set_row(1, v)
    a = h.step_spmv(1)
    r = h.r_get()
    if name != "row_too_long":
        assert np.array_equal(r, H * v)
    assert abs(a - np.dot(v, H * v)) <= 1e-13 * np.dot(np.abs(v), np.abs(H) * np.abs(v))
    h.close()


@pytest.mark.parametrize("groups", [2, 3, 7])
def test_two_phase_row_block_group_arm_in_the_kernel_bench_build(kb, groups):
    """Round 5 A/B arm (knob 22, kernel-bench build only: measured 8-110 % slower on C3, profiles/r05/ab_c3_row_block_groups.jsonl): the two
    phases interleaved over G groups of row blocks - products(g), rows(g), ... - on the same layout: the same products into the same slots,
    the same row sums: y and the alpha partial equal the default's and SciPy's bit for bit."""
    H = _two_phase_matrices()["graph_60000_real"]  # (dozens of row blocks x 118 column blocks, values that really round)
    M = H.shape[0]
    x = np.random.default_rng(1).uniform(-1, 1, M)
    ref = H * x
    out = []
    for g in (0, groups):
        h = kb.Handle(0)
        h.set_tuning(_capi.TUNE_SPMV_PLAN, 2)
        h.set_tuning(_capi.TUNE_PB_GROUPS, g)
        h.set_csr(M, 0, H.indptr, H.indices, H.data)
        assert h.spmv_plan() == "two-phase"
        y = h.spmv_host(x)
        h.basis_alloc(2)
        h.basis_set_row(1, x)
        out.append((y, h.step_spmv(1)))
        h.close()
    assert np.array_equal(out[0][0], ref) and np.array_equal(out[1][0], ref) and out[0][1] == out[1][1]


def test_two_phase_layout_on_a_second_device(hip):
    """the dynamic-LDS limit of the two-phase kernels is raised per DEVICE (hipFuncSetAttribute applies to the
    current device's function object): a handle on GPU 1 after one on GPU 0 in the same process must still launch with more
    than 64 KiB of dynamic LDS.  Needs two visible GPUs (skipped on a one-GPU box)."""
    import ctypes as C

    cnt = C.c_int()
    hip.load_library().lz_device_count(C.byref(cnt))
    if cnt.value < 2:
        pytest.skip("one visible GPU")
    H = synthetic.random_graph_laplacian(60000, 210000, seed=8).to_scipy()
    x = np.random.default_rng(1).uniform(-1, 1, H.shape[0])
    for dev in (0, 1):
        h = hip.Handle(dev)
        h.set_tuning(hip.TUNE_SPMV_PLAN, 2)
        h.set_csr(H.shape[0], 0, H.indptr, H.indices, H.data)
        assert h.spmv_plan() == "two-phase"
        assert np.array_equal(h.spmv_host(x), H * x)
        h.close()


def test_two_phase_spmv_in_the_run_loop(hip):
    """Whole Lanczos runs with and without the two-phase SpMV: r is bit-identical, alpha differs only by the grouping of
    its partial sums."""
    A = synthetic.random_graph_laplacian(60000, 210000, seed=8)
    M = A.shape[0]
    v0 = synthetic.reference_start_vector(M)
    v0 /= np.linalg.norm(v0)
    out = []
    for knob in (1, 2):
        h = hip.Handle(0)
        h.set_tuning(_capi.TUNE_SPMV_PLAN, knob)
        h.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
        a, b = h.run(50, v0)
        out.append((a, b, h.get_basis()))
        h.close()
    (a0, b0, V0), (a1, b1, V1) = out
    scale = np.abs(a0).max()
    assert np.abs(a1 - a0).max() < 1e-12 * scale and np.abs(b1 - b0).max() < 1e-12 * scale
    assert np.abs(V1[:10] - V0[:10]).max() < 1e-11
    assert np.abs(V1 @ V1.T - np.eye(50)).max() < 1e-12


def test_spmv_step_and_alpha(hip):
    H = MATS["graph_5000"]
    M = H.shape[0]
    h = hip.Handle(0)
    h.set_csr(M, 0, H.indptr, H.indices, H.data)
    h.basis_alloc(4)
    v = np.random.default_rng(2).standard_normal(M)
    h.basis_set_row(2, v)
    a = h.step_spmv(2)
    r = h.r_get()
    assert np.array_equal(r, H * v)
    assert abs(a - np.dot(v, r)) <= 1e-13 * np.dot(np.abs(v), np.abs(r))
    h.close()


def test_dense_gemv(hip):
    A = synthetic.dense_symmetric(517, seed=4)
    h = hip.Handle(0)
    h.set_dense(A)
    x = np.random.default_rng(3).standard_normal(517)
    y = h.spmv_host(x)
    np.testing.assert_allclose(y, A @ x, rtol=0, atol=1e-13 * (np.abs(A) @ np.abs(x)).max())
    h.close()


@pytest.mark.parametrize("M,n,j", [(1000, 6, 3), (70000, 9, 8), (5121, 5, 0), (33, 4, 2)])
def test_three_term_bit_exact(hip, M, n, j):
    rng = np.random.default_rng(M + j)
    h = hip.Handle(0)
    ptr = np.arange(M + 1, dtype=np.int32)
    h.set_csr(M, 0, ptr, ptr[:-1], np.ones(M))
    h.basis_alloc(n)
    r, vj, vm = rng.standard_normal((3, M))
    alpha, beta = rng.standard_normal(2)
    h.r_set(r)
    h.basis_set_row(j, vj)
    jm1 = j - 1
    if jm1 >= 0:
        h.basis_set_row(jm1, vm)
    nrm2 = h.step_three_term(j, jm1, alpha, beta)
    out = h.r_get()
    ref = r - vj * alpha - vm * beta if jm1 >= 0 else r - vj * alpha
    assert np.array_equal(out, ref)
    assert abs(nrm2 - ref.dot(ref)) <= 1e-13 * ref.dot(ref)
    h.close()


@pytest.mark.parametrize("M,n,j,nrows", [(1000, 6, 3, 4), (70000, 12, 11, 12), (5121, 7, 2, 7), (33, 4, 0, 1), (20000, 40, 39, 40)])
def test_reorth_matches_numpy(hip, M, n, j, nrows):
    """reference reorthogonalize(V, j): c = sum(V[j]*V, axis=1); V[j] = 2 V[j] - sum(c[:,None]*V, axis=0)."""
    rng = np.random.default_rng(M + 7 * j)
    V = np.zeros((n, M))
    V[:nrows] = rng.standard_normal((nrows, M)) / np.sqrt(M)
    h = hip.Handle(0)
    ptr = np.arange(M + 1, dtype=np.int32)
    h.set_csr(M, 0, ptr, ptr[:-1], np.ones(M))
    h.basis_alloc(n)
    for i in range(nrows):
        h.basis_set_row(i, V[i])
    _, c = h.step_reorth(j, nrows, scale=False)
    c_ref = np.sum(V[j] * V[:nrows], axis=1)
    scale = np.sum(np.abs(V[j]) * np.abs(V[:nrows]), axis=1)
    assert np.all(np.abs(c - c_ref) <= 1e-13 * scale)
    # the update is bit-exact GIVEN the device's own coefficients
    expect = 2 * V[j] - np.sum(c[:, None] * V[:nrows], axis=0)
    got = h.basis_get_row(j)
    assert np.array_equal(got, expect)
    for i in range(nrows):
        if i != j:
            assert np.array_equal(h.basis_get_row(i), V[i])
    h.close()


def test_scale_then_reorth(hip):
    M, n, j = 30011, 8, 5
    rng = np.random.default_rng(11)
    V = np.zeros((n, M))
    V[:j] = rng.standard_normal((j, M)) / np.sqrt(M)
    r = rng.standard_normal(M)
    h = hip.Handle(0)
    ptr = np.arange(M + 1, dtype=np.int32)
    h.set_csr(M, 0, ptr, ptr[:-1], np.ones(M))
    h.basis_alloc(n)
    for i in range(j):
        h.basis_set_row(i, V[i])
    h.r_set(r)
    beta, c = h.step_reorth(j, j + 1, scale=True)
    assert abs(beta - np.linalg.norm(r)) <= 1e-14 * np.linalg.norm(r)
    w = r / beta  # bit-exact division given the device's beta
    V[j] = w
    expect = 2 * w - np.sum(c[:, None] * V[: j + 1], axis=0)
    assert np.array_equal(h.basis_get_row(j), expect)
    assert np.array_equal(h.r_get(), r)
    h.close()


def test_reference_static_reorthogonalize(hip):
    from lanczos_amd import Lanczos

    rng = np.random.default_rng(0)
    V = rng.standard_normal((6, 999)) / 30
    W = V.copy()
    Lanczos.reorthogonalize(W, 3)
    c = np.sum(V[3] * V, axis=1)
    np.testing.assert_allclose(W[3], 2 * V[3] - np.sum(c[:, None] * V, axis=0), rtol=0, atol=1e-14)
    assert np.array_equal(np.delete(W, 3, 0), np.delete(V, 3, 0))


def test_two_phase_layouts_of_different_size_coexist(hip):
    """the dynamic-LDS limit of k_pb_products / k_pb_rows is a property of the KERNEL, shared by every layout in
    the process.  It used to be set to each new layout's own need, so a later, smaller layout (H^T of a two-sided run, a
    second handle) lowered it under the earlier one's launches.  Two layouts alive at once, the smaller built second; both
    must keep multiplying (bit-identical to SciPy)."""
    A1 = synthetic.random_graph_laplacian((1 << 20) + 4096, 3_600_000, seed=5)   # auto: two-phase, a full-LDS layout
    A2 = synthetic.random_graph_laplacian(300_000, 1_000_000, seed=6)            # forced, with a small tile capacity
    h1 = hip.Handle(0)
    h1.set_csr(A1.shape[0], 0, A1.rowptr, A1.colidx, A1.vals)
    assert h1.spmv_plan() == "two-phase"
    x1 = np.random.default_rng(1).standard_normal(A1.shape[0])
    y1 = A1.to_scipy() @ x1
    assert np.array_equal(h1.spmv_host(x1), y1)
    h2 = hip.Handle(0)
    h2.set_tuning(_capi.TUNE_SPMV_PLAN, 2)
    h2.set_tuning(_capi.TUNE_PB_ENTRIES, 3000)
    h2.set_csr(A2.shape[0], 0, A2.rowptr, A2.colidx, A2.vals)
    assert h2.spmv_plan() == "two-phase"
    x2 = np.random.default_rng(2).standard_normal(A2.shape[0])
    assert np.array_equal(h2.spmv_host(x2), A2.to_scipy() @ x2)
    assert np.array_equal(h1.spmv_host(x1), y1)  # the first layout's launches still fit their kernel's limit
    h2.close()
    assert np.array_equal(h1.spmv_host(x1), y1)
    h1.close()


def test_two_phase_spmv_with_duplicate_diagonal_entries(hip):
    """Unsummed duplicates are legal CSR.  A row that stores its diagonal twice cannot use the diagonal split of the two-phase
    layout (one LDS slot per row): the layout must fall back to sending every entry through the product stream - same bits as
    SciPy's csr_matvec, which adds the stored entries one by one."""
    import scipy.sparse

    A = synthetic.random_graph_laplacian(20000, 70000, seed=9).to_scipy().tocsr()
    # append a second diagonal entry (value 0.25) to 50 rows, keeping the columns sorted (the duplicate follows the original)
    rows = np.arange(0, 20000, 400)
    indptr, indices, data = A.indptr.copy(), A.indices.copy(), A.data.copy()
    out_idx, out_val, out_ptr = [], [], [0]
    for r in range(A.shape[0]):
        cols, vals = indices[indptr[r]:indptr[r + 1]], data[indptr[r]:indptr[r + 1]]
        if r in rows and (cols == r).any():
            k = int(np.flatnonzero(cols == r)[0]) + 1
            cols, vals = np.insert(cols, k, r), np.insert(vals, k, 0.25)
        out_idx.append(cols)
        out_val.append(vals)
        out_ptr.append(out_ptr[-1] + len(cols))
    B = scipy.sparse.csr_matrix((np.concatenate(out_val), np.concatenate(out_idx), np.array(out_ptr)), shape=A.shape)
    assert B.nnz == A.nnz + 50 and not B.has_canonical_format
    h = hip.Handle(0)
    h.set_tuning(_capi.TUNE_SPMV_PLAN, 2)
    h.set_csr(B.shape[0], 0, B.indptr, B.indices, B.data)
    assert h.spmv_plan() == "two-phase"
    x = np.random.default_rng(3).standard_normal(B.shape[0])
    y = np.zeros(B.shape[0])
    for r in range(B.shape[0]):  # csr_matvec's order: one stored entry after the other
        acc = 0.0
        for k in range(B.indptr[r], B.indptr[r + 1]):
            acc += B.data[k] * x[B.indices[k]]
        y[r] = acc
    assert np.array_equal(h.spmv_host(x), y)
    h.close()
