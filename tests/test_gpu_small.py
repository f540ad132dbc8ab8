"""Loop structures for the sizes of the reference's own scripts (1Dbox.py N = 500, 1Ddeuteron.py N = n = 1001) and BASELINE
config C1 (dense 512, k = 20).  Product library: the fused three-launch path (small problems) and the five-launch loop
(mid-size), each BIT FOR BIT equal to the plain six-launch loop.  Kernel-bench build (fixture `kb`,
liblanczos_kbench.so): the two RETIRED engines - the whole run as one cooperative kernel (lz_small.hip) and one launch per
step - which reproduce the same bits (same reduction trees, same MFMA sequence, same element-wise arithmetic) but
measured no faster on MI355X (the printed timings), so they left the product in round 3."""
import time

import numpy as np
import pytest

from conftest import load_golden
from lanczos_amd import Lanczos, synthetic, _capi

pytestmark = pytest.mark.gpu


def _cases():
    d1, box = load_golden("box1d_N500_n50")
    d2, deut = load_golden("deuteron1d_N1001_n1001")
    return {
        "c1_dense512_k20": (synthetic.dense_symmetric(512, seed=0), 20),
        "box1d_dense_N500_n50": (box.toarray(), 50),
        "deuteron1d_csr_N1001_n1001": (deut, 1001),
        "lap2d_32x32_fixedk_n30": (synthetic.laplacian_2d_5pt(32, 32).to_scipy(), 30),
        "lap3d_10x9x8_fixedk_n40": (synthetic.laplacian_3d_7pt(10, 9, 8).to_scipy(), 40),
        "odd_dense_M333_n333": (synthetic.dense_symmetric(333, seed=3), 333),
    }


def _run(hip, H, n, engine_off, flags=None, knob=None):
    import scipy.sparse

    h = hip.Handle(0)
    h.set_options(hip.FLAG_FUSED_NORM if flags is None else flags)
    # knob 15: 0 = default (fused three-launch path where it applies), 1 = plain six-launch path, 2 = one-kernel engine (opt-in)
    h.set_tuning(_capi.TUNE_LOOP, knob if knob is not None else (1 if engine_off else 2))
    if scipy.sparse.issparse(H):
        A = H.tocsr()
        h.set_csr(A.shape[0], 0, A.indptr, A.indices, A.data)
    else:
        h.set_dense(H)
    M = H.shape[0]
    v0 = synthetic.reference_start_vector(M)
    v0 /= np.linalg.norm(v0)
    a, b = h.run(n, v0)  # first run: code-object load etc.
    ts = []
    for _ in range(5):
        h.synchronize()
        t0 = time.perf_counter()
        a, b = h.run(n, v0)
        ts.append(time.perf_counter() - t0)
    out = (a, b, h.get_basis(), h.last_engine(), float(np.median(ts)))
    h.close()
    return out


@pytest.mark.parametrize("name", list(_cases()))
def test_engine_is_bit_identical_to_the_kernel_path(hip, kb, name):
    H, n = _cases()[name]
    a1, b1, V1, e1, t1 = _run(hip, H, n, engine_off=True)   # product library, plain six-launch loop
    a0, b0, V0, e0, t0 = _run(kb, H, n, engine_off=False)   # kernel-bench build, one-kernel engine
    assert e1 == "kernels" and e0 == "small"
    assert np.array_equal(a0, a1) and np.array_equal(b0, b1), (np.abs(a0 - a1).max(), np.abs(b0 - b1).max())
    assert np.array_equal(V0, V1), np.abs(V0 - V1).max()
    print(f"\n[small-engine] {name}: kernels {1e3 * t1:.3f} ms, engine {1e3 * t0:.3f} ms, x{t1 / t0:.2f}")


def _fused_cases():
    c = dict(_cases())
    c["lap2d_64x64_n60"] = (synthetic.laplacian_2d_5pt(64, 64).to_scipy(), 60)           # 4096 rows: 8 pass-1 slices
    c["graph_M3000_n50"] = (synthetic.random_graph_laplacian(3000, 10000, seed=7).to_scipy(), 50)
    c["ragged_M700_n25"] = (load_golden("ragged_M700_n25")[1], 25)                         # a 700-entry row
    c["dense_M2047_n30"] = (synthetic.dense_symmetric(2047, seed=5), 30)
    return c


@pytest.mark.parametrize("name", list(_fused_cases()))
def test_fused_launch_path_is_bit_identical_and_faster(hip, name):
    """Default path of small problems: three launches per step (second-stage reductions and the three-term recurrence
    folded into their consumer kernels) against the plain six-launch path: same bits, fewer microseconds."""
    H, n = _fused_cases()[name]
    a1, b1, V1, e1, t1 = _run(hip, H, n, engine_off=True, knob=1)
    a0, b0, V0, e0, t0 = _run(hip, H, n, engine_off=False, knob=0)
    assert e1 == "kernels" and e0 == "fused"
    assert np.array_equal(a0, a1) and np.array_equal(b0, b1), (np.abs(a0 - a1).max(), np.abs(b0 - b1).max())
    assert np.array_equal(V0, V1), np.abs(V0 - V1).max()
    print(f"\n[fused-launch] {name}: six launches {1e3 * t1:.3f} ms, three launches {1e3 * t0:.3f} ms, x{t1 / t0:.2f}")


@pytest.mark.parametrize("name", list(_fused_cases()))
def test_step_kernels_are_bit_identical(hip, kb, name):
    """A/B arm (tuning knob 15 = 5; padded rows <= 1280, n <= 64, short CSR rows or dense): ONE launch per step - every block
    redoes the vector work of the step and multiplies its share of the rows - against the plain six-launch path: same bits
    (and slower than the default three-launch path: the printed timings are the record).  Where the conditions do not hold
    the knob falls back to the fused-launch path."""
    H, n = _fused_cases()[name]
    a1, b1, V1, e1, t1 = _run(hip, H, n, engine_off=True, knob=1)
    a0, b0, V0, e0, t0 = _run(kb, H, n, engine_off=False, knob=5)
    M = H.shape[0]
    import scipy.sparse
    long_rows = scipy.sparse.issparse(H) and np.diff(H.tocsr().indptr).max() > 32
    applies = M <= 1280 and n <= 64 and not long_rows
    assert e1 == "kernels" and e0 == ("step" if applies else "fused"), (e0, applies)
    assert np.array_equal(a0, a1) and np.array_equal(b0, b1), (np.abs(a0 - a1).max(), np.abs(b0 - b1).max())
    assert np.array_equal(V0, V1), np.abs(V0 - V1).max()
    print(f"\n[step-kernels] {name} ({e0}): six launches {1e3 * t1:.3f} ms, knob 15 = 5 {1e3 * t0:.3f} ms, x{t1 / t0:.2f}")


def _mid_cases():
    return {
        "lap2d_80x80_n30": (synthetic.laplacian_2d_5pt(80, 80).to_scipy(), 30),                      # 6400 rows: 13 pass-1 slices
        "lap3d_24x20x18_n40": (synthetic.laplacian_3d_7pt(24, 20, 18).to_scipy(), 40),
        "graph_M50000_n25": (synthetic.random_graph_laplacian(50000, 175000, seed=11).to_scipy(), 25),
        "lap2d_700x500_n24": (synthetic.laplacian_2d_5pt(700, 500).to_scipy(), 24),                   # 350 000 rows
        "dense_M4500_n12": (synthetic.dense_symmetric(4500, seed=6), 12),
    }


@pytest.mark.parametrize("name", list(_mid_cases()))
def test_three_term_fused_loop_is_bit_identical(hip, name):
    """Default loop of everything that is not small: the three-term recurrence rides in the prologue of the next step's
    pass 1 (five launches per step) - against the plain six-launch loop: same bits."""
    H, n = _mid_cases()[name]
    a1, b1, V1, e1, t1 = _run(hip, H, n, engine_off=True, knob=1)
    a0, b0, V0, e0, t0 = _run(hip, H, n, engine_off=False, knob=0)
    assert e1 == "kernels" and e0 == "three-term-fused"
    assert np.array_equal(a0, a1) and np.array_equal(b0, b1), (np.abs(a0 - a1).max(), np.abs(b0 - b1).max())
    assert np.array_equal(V0, V1), np.abs(V0 - V1).max()
    print(f"\n[three-term-fused] {name}: six launches {1e3 * t1:.3f} ms, five launches {1e3 * t0:.3f} ms, x{t1 / t0:.2f}")


def test_device_scope_arm_is_bit_identical_too(hip, kb):
    """tuning knob 15 = 3: the same kernel on a plain grid with device-scope coherence (what it costs when the blocks do
    NOT share an XCD) - an A/B arm, same bits."""
    H, n = _cases()["c1_dense512_k20"]
    a1, b1, V1, e1, t1 = _run(hip, H, n, engine_off=True)
    h = kb.Handle(0)
    h.set_options(hip.FLAG_FUSED_NORM)
    h.set_tuning(_capi.TUNE_LOOP, 3)
    h.set_dense(H)
    v0 = synthetic.reference_start_vector(512)
    v0 /= np.linalg.norm(v0)
    a, b = h.run(n, v0)
    assert h.last_engine() == "small" and np.array_equal(a, a1) and np.array_equal(b, b1) and np.array_equal(h.get_basis(), V1)
    h.close()


def test_loop_selection(hip, kb):
    """choose_loop (lz_loops.hip) through lz_last_engine: the product library knows three loops + one-reduce; the retired
    engines answer only in the kernel-bench build, and only where they apply."""
    rag = load_golden("ragged_M700_n25")[1]  # a 700-entry row: one lane per row would be a chain of dependent loads
    assert _run(kb, rag, 25, engine_off=False)[3] == "kernels"
    H = synthetic.laplacian_2d_5pt(40, 40).to_scipy()  # 1600 rows > 1280
    assert _run(kb, H, 10, engine_off=False)[3] == "kernels"
    Hs = synthetic.laplacian_2d_5pt(20, 20).to_scipy()
    assert _run(kb, Hs, 10, engine_off=False, flags=0)[3] == "kernels"  # not fused-norm mode
    assert _run(kb, Hs, 10, engine_off=False, knob=5)[3] == "step"  # the one-launch-per-step arm
    assert _run(kb, Hs, 10, engine_off=False, flags=hip.FLAG_FUSED_NORM | hip.FLAG_REORTH_PARTIAL)[3] == "partial-device"  # round 4: its own loop
    assert _run(kb, Hs, 10, engine_off=False)[3] == "small"
    for lib in (hip, kb):
        assert _run(lib, Hs, 10, engine_off=True)[3] == "kernels"  # knob 15 = 1: the plain path
        assert _run(lib, Hs, 10, engine_off=False, knob=0)[3] == "fused"  # the default for small problems
        assert _run(lib, Hs, 10, engine_off=False, knob=0, flags=0)[3] == "kernels"  # reference order (scale, then dot): plain loop
        assert _run(lib, synthetic.laplacian_2d_5pt(80, 80).to_scipy(), 10, engine_off=False, knob=0)[3] == "three-term-fused"  # 6400 rows: 13 slices
        assert _run(lib, Hs, 10, engine_off=False, knob=0, flags=hip.FLAG_FUSED_NORM | hip.FLAG_ONE_REDUCE)[3] == "one-reduce"


def test_drop_in_class_on_the_reference_scripts_sizes():
    """Through `Lanczos`: 1Dbox.py's N = 500 dense matrix (fused-launch path by default); the golden coefficients hold."""
    d, H = load_golden("box1d_N500_n50")
    Lanczos.verbose = False
    s = Lanczos(H.toarray())
    s.execute_Lanczos(50)
    assert s._handle.last_engine() == "fused"
    scale = max(np.abs(d["alpha"]).max(), np.abs(d["beta"]).max())
    assert np.abs(np.diag(s.H_eff) - d["alpha"]).max() <= 1e-10 * scale
    assert np.abs(s.H_eigvals - d["H_eigvals"]).max() <= 1e-10 * np.abs(d["H_eigvals"]).max()
