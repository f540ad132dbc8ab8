#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE's CPU path here.

Run only in the build container (needs /root/reference):

    cd /tmp && MPLBACKEND=Agg python /root/repo/oracle/gen_golden.py

The reference imports ``cupy`` at module top level (Python/Regular/Lanczos.py:4-5)
but never touches it on the ``use_cuda=False`` path, so empty stub modules are
registered first.  For every case the script

  1. builds the input (own generator, or the reference's Hamiltonian builder),
  2. runs the reference ``Lanczos`` / ``IrrLanczos.execute_LanczosOld`` on CPU,
  3. runs oracle/lanczos_ref.py on the same input and records the max abs
     difference (expected: exactly 0.0),
  4. stores inputs + reference outputs as a compressed .npz (data only - no
     reference source travels).

Fixture shapes follow SURVEY.md section 8c (i)-(v).
"""
import contextlib
import io
import os
import sys
import types

import numpy as np
import scipy.sparse

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/Python"
OUT = os.path.join(REPO, "tests", "golden")

for _m in ("cupy", "cupyx", "cupyx.scipy", "cupyx.scipy.sparse"):
    sys.modules.setdefault(_m, types.ModuleType(_m))
sys.path.insert(0, os.path.join(REF, "Regular"))
sys.path.insert(0, os.path.join(REF, "Irregular"))
sys.path.insert(0, REPO)

import Lanczos as ref_regular  # noqa: E402  (reference, read-only)
import IrrLanczos as ref_irregular  # noqa: E402
import Hamiltonian as ref_hamiltonian  # noqa: E402

from oracle import lanczos_ref as oracle  # noqa: E402
from oracle import two_sided_ref  # noqa: E402
from lanczos_amd import synthetic  # noqa: E402


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        return fn(*a, **k)


def run_reference(H, n, seed, v0, irregular=False):
    if irregular:
        obj = ref_irregular.IrrLanczos(H)
        quiet(obj.execute_LanczosOld, n, seed=seed, use_cuda=False, v0=v0)
    else:
        obj = ref_regular.Lanczos(H)
        quiet(obj.execute_Lanczos, n, seed=seed, use_cuda=False, v0=v0)
    quiet(obj.get_H_eigs)
    return obj


def case(name, H, n, seed=99, v0=None, store_matrix=True, gen=None, keep_V=None):
    H = scipy.sparse.csr_matrix(H)
    H.sort_indices()
    M = H.shape[0]
    reg = run_reference(H, n, seed, v0)
    irr = run_reference(H, n, seed, v0, irregular=True)
    assert np.array_equal(reg.H_eff, irr.H_eff) and np.array_equal(reg.V, irr.V), "Regular != Irregular-Old"
    alpha = np.diag(reg.H_eff).copy()
    beta = np.diag(reg.H_eff, 1).copy()
    a, b, V = oracle.execute_lanczos(H, n, seed=seed, v0=v0)
    a2, b2, V2 = oracle.execute_lanczos(H, n, seed=seed, v0=v0, economy=True)
    assert np.array_equal(a, a2) and np.array_equal(b, b2) and np.array_equal(V, V2), "economy sweep changed bits"
    theta, S, Y = oracle.ritz_pairs(oracle.build_h_eff(a, b), V)
    diff = max(
        np.abs(a - alpha).max(),
        np.abs(b - beta).max(),
        np.abs(V.T - reg.V).max(),
        np.abs(theta - reg.H_eigvals).max(),
        np.abs(Y - reg.H_eigvecs).max(),
        np.abs(oracle.build_h_eff(a, b) - reg.H_eff).max(),
    )
    if keep_V is None:
        keep_V = M * n <= 200_000
    data = dict(
        name=name, M=M, n=n, seed=seed,
        alpha=alpha, beta=beta, H_eigvals=reg.H_eigvals,
        norm_closest_to_1=oracle.is_normalized(reg.H_eigvecs),
        max_offdiag_gram=oracle.is_orthogonal(reg.H_eigvecs),
        ref_vs_oracle_maxabs=diff,
        numpy_version=np.__version__, scipy_version=scipy.__version__,
    )
    if v0 is not None:
        data["v0"] = np.asarray(v0)
    if store_matrix:
        data.update(rowptr=H.indptr.astype(np.int32), colidx=H.indices.astype(np.int32), vals=H.data)
    if gen is not None:
        data["generator"] = gen
    if keep_V:
        data["V"] = np.ascontiguousarray(reg.V.T)  # (n, M): basis vector j is row j
        data["H_eigvecs_first3"] = reg.H_eigvecs[:, :3].copy()
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **data)
    print(f"{name:28s} M={M:6d} n={n:5d} ref-vs-oracle max|diff|={diff:.3e} eig[:3]={reg.H_eigvals[:3]}")


def two_sided_case(name, H, n, seed=99, gen=None, store_matrix=False):
    """IrrLanczos.execute_Lanczos (two-sided, IrrLanczos.py:77-187) on the reference's CPU branch; v0=None is the only
    start the reference supports there."""
    H = scipy.sparse.csr_matrix(H, dtype=np.float64)
    H.sort_indices()
    M = H.shape[0]
    obj = ref_irregular.IrrLanczos(H)
    quiet(obj.execute_Lanczos, n, seed=seed, use_cuda=False, v0=None)
    quiet(obj.get_H_eigs)
    a, b, g, Q, P, Qb, Pb = two_sided_ref.execute_two_sided(H, n, seed=seed, return_all=True)
    T = two_sided_ref.build_h_eff(a, b, g)
    diff = max(np.abs(T - obj._H_eff).max(), np.abs(Q.T - obj._V).max())
    q0, p0 = two_sided_ref.start_pair(M, seed)
    data = dict(
        name=name, M=M, n=n, seed=seed, alpha=a, beta=b, gamma=g, H_eff=obj._H_eff.copy(), H_eigvals=obj.H_eigvals.copy(),
        q0=q0, p0=p0, ref_vs_oracle_maxabs=diff, numpy_version=np.__version__, scipy_version=scipy.__version__,
    )
    if M * n <= 200_000:
        data["V"] = np.ascontiguousarray(obj._V.T)  # (n, M)
    if store_matrix:
        data.update(rowptr=H.indptr.astype(np.int32), colidx=H.indices.astype(np.int32), vals=H.data)
    if gen is not None:
        data["generator"] = gen
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **data)
    print(f"{name:28s} M={M:6d} n={n:5d} ref-vs-oracle max|diff|={diff:.3e} alpha[:3]={a[:3]} eig[:2]={obj.H_eigvals[:2]}")


def two_sided_main():
    two_sided_case("two_sided_lap2d_16x16_n12", synthetic.laplacian_2d_5pt(16, 16).to_scipy(), 12, gen="laplacian_2d_5pt(16, 16)")
    two_sided_case("two_sided_graph_M2000_n20", synthetic.random_graph_laplacian(2000, 7000, seed=1234).to_scipy(), 20,
                   gen="random_graph_laplacian(2000, 7000, seed=1234)")
    rng = np.random.default_rng(11)
    N = scipy.sparse.random(300, 300, density=0.03, random_state=rng, format="csr") + scipy.sparse.diags(np.linspace(1, 4, 300))
    two_sided_case("two_sided_nonsym_M300_n10", N, 10, seed=5, store_matrix=True)
    two_sided_case("two_sided_lap2d_8x8_n2", synthetic.laplacian_2d_5pt(8, 8).to_scipy(), 2, gen="laplacian_2d_5pt(8, 8)")


def mem_safe_main():
    """The static IrrLanczos.bireorthogonalize(..., mem_safe=True) (IrrLanczos.py:398-407) on the reference's CPU branch: three
    input pairs - zero rows past j (how the driver's arrays look), every row filled, and j = 0."""
    rng = np.random.default_rng(2024)
    data = dict(numpy_version=np.__version__)
    worst = 0.0
    for tag, (n, M, j, fill) in {"a": (8, 300, 5, False), "b": (6, 1000, 3, True), "c": (5, 130, 0, True)}.items():
        V1 = rng.uniform(-1, 1, (n, M))
        V2 = rng.uniform(-1, 1, (n, M))
        if not fill:
            V1[j + 1:] = 0
            V2[j + 1:] = 0
        R1, R2 = V1.copy(), V2.copy()
        quiet(ref_irregular.IrrLanczos.bireorthogonalize, R1, R2, None, None, j, use_cuda=False, mem_safe=True)
        O1, O2 = V1.copy(), V2.copy()
        two_sided_ref.bireorthogonalize_mem_safe(O1, O2, j)
        worst = max(worst, np.abs(O1 - R1).max(), np.abs(O2 - R2).max())
        assert np.array_equal(np.delete(R1, j, 0), np.delete(V1, j, 0)) and np.array_equal(np.delete(R2, j, 0), np.delete(V2, j, 0))
        data.update({f"{tag}_V1": V1, f"{tag}_V2": V2, f"{tag}_j": j, f"{tag}_out1": R1[j].copy(), f"{tag}_out2": R2[j].copy()})
    data["ref_vs_oracle_maxabs"] = worst
    np.savez_compressed(os.path.join(OUT, "bireorth_mem_safe.npz"), **data)
    print(f"bireorth_mem_safe            ref-vs-oracle max|diff|={worst:.3e}")


def bireorth_j0_main():
    """The static IrrLanczos.bireorthogonalize(V1, V2, q_basis, p_basis, j=0) - default branch (IrrLanczos.py:408-441) on the
    reference's CPU branch.  With j = 0 both projection loops are empty: the pair is rescaled to V1[0].V2[0] = +-1 and seeds the
    two orthonormal bases - and the LAST statement (:441, the maximum of an empty array) raises ValueError, after every array
    has been updated in place.  Recorded: the four row-0 outputs and the exception's type and text."""
    rng = np.random.default_rng(77)
    n, M = 4, 257
    arrs = [rng.uniform(-1, 1, (n, M)) for _ in range(4)]
    arrs[1][0] = -np.abs(arrs[1][0]) * np.sign(arrs[0][0])  # V1[0].V2[0] < 0: the sign factor of :420 matters
    ref = [a.copy() for a in arrs]
    err = None
    try:
        quiet(ref_irregular.IrrLanczos.bireorthogonalize, *ref, 0, use_cuda=False)
    except Exception as e:  # noqa: BLE001 - the point is to record what the reference raises
        err = e
    assert isinstance(err, ValueError), err
    ora = [a.copy() for a in arrs]
    two_sided_ref.bireorthogonalize(*ora, 0)
    worst = max(np.abs(o - r).max() for o, r in zip(ora, ref))
    for a, r in zip(arrs, ref):
        assert np.array_equal(a[1:], r[1:])
    data = dict(numpy_version=np.__version__, error_type=type(err).__name__, error_text=str(err), ref_vs_oracle_maxabs=worst)
    for name, a, r in zip(("V1", "V2", "q_basis", "p_basis"), arrs, ref):
        data[name] = a
        data[name + "_out0"] = r[0].copy()
    np.savez_compressed(os.path.join(OUT, "bireorth_default_j0.npz"), **data)
    print(f"bireorth_default_j0          ref-vs-oracle max|diff|={worst:.3e}; the reference raises {type(err).__name__}: {err}")


def deuteron_potential(x, y, z):
    # same functional form/constants the reference's driver uses (3Ddeuteron.py:51-61); data, not code of the path
    r = np.sqrt(x**2 + y**2 + z**2)
    eWell = 54.531
    return 40.0 * eWell * np.exp(-((r / 0.25) ** 4.0)) - 65.4823128982115 * np.exp(-((r / 1.7) ** 4.0))


def main():
    os.chdir(os.environ.get("TMPDIR", "/tmp"))  # the reference's Hamiltonian creates ./T_matrices
    if "--two-sided-only" in sys.argv:
        return two_sided_main()
    if "--mem-safe-only" in sys.argv:
        return mem_safe_main()
    if "--bireorth-j0-only" in sys.argv:
        return bireorth_j0_main()

    # (i) C1: dense 512 x 512 random symmetric, n = 20, explicit v0
    A = synthetic.dense_symmetric(512, seed=0)
    v0 = np.random.default_rng(0).uniform(-1, 1, 512)
    case("c1_dense512_n20", A, 20, v0=v0, store_matrix=False, gen="dense_symmetric(512, seed=0); v0=default_rng(0).uniform(-1,1,512)")

    # (ii) 2-D periodic 5-point 32 x 32, n = 30, default start vector (legacy RNG, seed 99)
    case("lap2d_32x32_n30", synthetic.laplacian_2d_5pt(32, 32).to_scipy(), 30, store_matrix=False, gen="laplacian_2d_5pt(32, 32)")

    # (iii) mini 3Ddeuteron: the reference's own builder, N = 12, 27-point, seed 78 (3Ddeuteron.py:63-95 scaled down)
    N, L = 12, 25
    dx = float(L) / N
    T_factor = 197.327**2 / (2 * 469.4592) / dx**2
    ham = ref_hamiltonian.Hamiltonian(N, L, deuteron_potential, T_factor)
    quiet(ham.create_sparse_T)
    quiet(ham.create_sparse_V)
    H = -ham.T_sparse + ham.V_sparse
    H.sort_indices()
    case("deuteron3d_N12_27pt_n100", H, 100, seed=78, keep_V=True)

    # (iii-b) the reference's builder itself at N = 6: T (7- and 27-point), V, H - pins the device assembly kernel
    N6 = 6
    dx6 = 25.0 / N6
    Tf6 = 197.327**2 / (2 * 469.4592) / dx6**2
    ham6 = ref_hamiltonian.Hamiltonian(N6, 25, deuteron_potential, Tf6)
    out = {"N": N6, "L": 25, "T_factor": Tf6, "weights_27point": ham6.weights_27point, "weights_7point": ham6.weights_7point,
           "x": ham6.x}
    for pts in ("7", "27"):
        for f in ("T_matrices/T_N=6_Laplace=%s.npz" % pts,):
            if os.path.exists(f):
                os.unlink(f)
        quiet(ham6.create_sparse_T, pts)
        T = ham6.T_sparse.copy()
        out["T%s_sorted_as_built" % pts] = bool(T.has_sorted_indices)
        T.sort_indices()
        out["T%s_rowptr" % pts], out["T%s_colidx" % pts], out["T%s_vals" % pts] = T.indptr, T.indices, T.data
        quiet(ham6.create_sparse_V)
        Hh = -ham6.T_sparse + ham6.V_sparse
        Hh.sort_indices()
        out["H%s_rowptr" % pts], out["H%s_colidx" % pts], out["H%s_vals" % pts] = Hh.indptr, Hh.indices, Hh.data
    out["V_diag"] = ham6.V_sparse.diagonal()
    out["neighbors_7_of_row_0"] = np.array(ham6.Laplacian_7point(0)[0])
    out["neighbors_27_of_row_215"] = np.array(ham6.Laplacian_27point(215)[0])
    np.savez_compressed(os.path.join(OUT, "hamiltonian_N6.npz"), **out)
    print("hamiltonian_N6 fixture written; T sorted as built:", out["T7_sorted_as_built"], out["T27_sorted_as_built"])

    # (iv) 1Dbox.py:5-22 matrix, N = 500, n = 50 (dense there; wrapped as CSR for the CPU path)
    Nb = 500
    pot = np.zeros(Nb)
    pot[Nb // 4 : (3 * Nb) // 4] = -10
    Hb = np.diag(2 + pot) - np.diag(np.ones(Nb - 1), 1) - np.diag(np.ones(Nb - 1), -1)
    case("box1d_N500_n50", Hb, 50)

    # (v) 1Ddeuteron.py:6-54, N = n = 1001: the n == M edge
    Nd = 1001
    dxd = 25.0 / Nd
    rr = np.linspace(0, 25, Nd)
    Vd = 40.0 * 54.531 * np.exp(-((rr / 0.25) ** 4.0)) - 65.4823128982115 * np.exp(-((rr / 1.7) ** 4.0))
    Vd[Nd - 1] = 0.0  # the script's loop stops at N-2
    Tf = 197.327**2 / (2 * 469.4592) / dxd**2
    T = scipy.sparse.diags([np.full(Nd - 1, Tf), np.r_[-Tf, np.full(Nd - 2, -2 * Tf), -Tf], np.full(Nd - 1, Tf)], [-1, 0, 1], format="csr")
    case("deuteron1d_N1001_n1001", -T + scipy.sparse.diags(Vd, format="csr"), 1001, keep_V=False)

    # extra structure coverage for the hot path: 3-D 7-point, random irregular graph, ragged rows
    case("lap3d_8x8x8_n40", synthetic.laplacian_3d_7pt(8, 8, 8).to_scipy(), 40, store_matrix=False, gen="laplacian_3d_7pt(8, 8, 8)")
    case("graph_M2000_E7000_n40", synthetic.random_graph_laplacian(2000, 7000, seed=1234).to_scipy(), 40, store_matrix=False, gen="random_graph_laplacian(2000, 7000, seed=1234)")
    rng = np.random.default_rng(7)
    R = scipy.sparse.random(700, 700, density=0.02, random_state=rng, format="csr")
    R = R + R.T + scipy.sparse.diags(np.linspace(-3, 3, 700))
    R = R.tolil()
    R[5, :] = 0.01  # one dense row/column and a few empty-off-diagonal rows: ragged CSR
    R[:, 5] = 0.01
    case("ragged_M700_n25", R.tocsr(), 25, seed=3)
    # n = 2: smallest n the reference survives (beta has one entry)
    case("lap2d_8x8_n2", synthetic.laplacian_2d_5pt(8, 8).to_scipy(), 2, store_matrix=False, gen="laplacian_2d_5pt(8, 8)")
    # the Irregular copy's two-sided variant (IrrLanczos.py:77-187)
    two_sided_main()
    mem_safe_main()
    bireorth_j0_main()


if __name__ == "__main__":
    main()
