"""The row-partitioned path on real kernels: several ranks share the one GPU of the test box
(RCCL refuses two ranks per device, so these use the host-staged collective backend over gloo;
the RCCL entry points are exercised with a 1-rank communicator).  Every rank must reproduce the
single-rank device result and the CPU oracle."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from lanczos_amd import _capi, distributed, synthetic

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    sys.path.insert(0, os.environ["LZ_ROOT"])
    import lanczos_amd
    from lanczos_amd import _capi, distributed, partition, synthetic
    from oracle import lanczos_ref as oracle
    # torch (when used at all) comes first: one HIP runtime in the process; the socket bootstrap needs no torch
    boot = distributed.SocketBootstrap() if os.environ.get("LZ_BOOT") == "socket" else distributed.TorchBootstrap()
    lanczos_amd.load_library()
    assert len(_capi.mapped_runtimes()["amdhip64"]) == 1, _capi.mapped_runtimes()
    out = {}
    cases = [("lap2d", lambda lo, hi: synthetic.laplacian_2d_5pt(96, 80, rows=(lo, hi)), 96 * 80, "auto", 40),
             ("lap3d", lambda lo, hi: synthetic.laplacian_3d_7pt(24, 20, 18, rows=(lo, hi)), 24 * 20 * 18, "halo", 30),
             ("graph", lambda lo, hi: synthetic.random_graph_laplacian(6000, 20000, seed=4).row_slice(lo, hi), 6000, "auto", 30),
             ("graph_halo", lambda lo, hi: synthetic.random_graph_laplacian(6000, 20000, seed=4).row_slice(lo, hi), 6000, "halo", 30),
             ("lap3d_partial", lambda lo, hi: synthetic.laplacian_3d_7pt(20, 18, 16, rows=(lo, hi)), 20 * 18 * 16, "halo", 120),
             ("dense", lambda lo, hi: synthetic.dense_symmetric_hashed(701, rows=(lo, hi), seed=3), 701, "auto", 25),
             # LZ_FLAG_ONE_REDUCE: one all-reduce (+ one halo exchange / all-gather) per iteration instead of two
             ("lap2d_onereduce", lambda lo, hi: synthetic.laplacian_2d_5pt(96, 80, rows=(lo, hi)), 96 * 80, "halo", 40),
             ("graph_onereduce", lambda lo, hi: synthetic.random_graph_laplacian(6000, 20000, seed=4).row_slice(lo, hi), 6000, "auto", 30),
             # LZ_FLAG_REORTH_PARTIAL | LZ_FLAG_ONE_REDUCE (round 5): the device-decided partial loop with ONE all-reduce per step
             ("lap3d_partial_onereduce", lambda lo, hi: synthetic.laplacian_3d_7pt(20, 18, 16, rows=(lo, hi)), 20 * 18 * 16, "halo", 120),
             ("graph_partial_onereduce", lambda lo, hi: synthetic.random_graph_laplacian(6000, 20000, seed=4).row_slice(lo, hi), 6000, "auto", 60),
             # the column-blocked two-phase SpMV (what config C3 runs) on every rank's row block against the all-gathered vector
             ("graph_twophase", lambda lo, hi: synthetic.random_graph_laplacian(6000, 20000, seed=4).row_slice(lo, hi), 6000, "auto", 30),
             # M = 1000 is not a multiple of world * 32: the last rank's all-gather chunk has a tail no kernel writes; the
             # fresh basis allocation is NaN-poisoned first (tuning knob 13), so the run only survives if the library
             # clears what the dense GEMV reads against zero-padded columns (0 * NaN = NaN)
             ("dense_poison", lambda lo, hi: synthetic.dense_symmetric_hashed(1000, rows=(lo, hi), seed=5), 1000, "auto", 25),
             # BASELINE C4 (3-D 7-point, z-slab partition, k = 200) and C5 (2-D 5-point, k = 500) at reduced size
             ("c4_slab_k200", lambda lo, hi: synthetic.laplacian_3d_7pt(24, 24, 32, rows=(lo, hi)), 24 * 24 * 32, "halo", 200),
             ("c5_k500", lambda lo, hi: synthetic.laplacian_2d_5pt(160, 120, rows=(lo, hi)), 160 * 120, "halo", 500)]
    for name, build, M, mode, n in cases:
        b = partition.row_bounds(M, boot.world)
        lo, hi = b[boot.rank], b[boot.rank + 1]
        opts = 64 if "_partial" in name else 0  # LZ_FLAG_REORTH_PARTIAL: every rank must take the same sweep decisions
        s = distributed.DistributedLanczos(build(lo, hi), M, boot, device_id=0, backend="host", mode=mode, fused_norm=(name != "lap3d"), options=opts,
                                           one_reduce=name.endswith("_onereduce"), tuning={14: 2} if name == "graph_twophase" else None)
        if name == "graph_twophase":
            assert s.h.spmv_plan() == "two-phase"
        if name == "dense_poison":
            s.h.set_tuning(_capi.TUNE_POISON_BASIS, 1)
        a, bta = s.execute_Lanczos(n)
        sweeps, misses, engine = s.h.last_sweeps(), s.h.last_sweep_misses(), s.h.last_engine()
        device_built_equal = None
        if name == "c4_slab_k200":  # the same slab assembled on the device from the boundary-row plan: identical matrix, identical run
            s2 = distributed.DistributedLanczos.from_stencil((24, 24, 32), 7, boot, device_id=0, backend="host")
            a2, b2 = s2.execute_Lanczos(n)
            device_built_equal = bool(np.array_equal(a2, a) and np.array_equal(b2, bta) and np.array_equal(s2.V_local, s.V_local))
        comm_launches = s.timings()["comm"]["launches"]  # of the run (the diagnostics below add their own)
        theta = s.get_H_eigs()
        V = s.V_local
        Y = s.H_eigvecs_local
        qual = s.ritz_quality() if name in ("lap2d", "graph", "dense", "lap3d") else None  # collective; restores basis row 0
        if qual is not None:
            assert np.array_equal(s.V_local, V)
        chunked = None
        if name in ("lap2d", "graph"):
            # the CHUNKED Ritz mode on a row-block partition (what a rank does when a second rows x n array does not fit beside its
            # basis): Y re-formed in 512-row chunks, Gram matrix accumulated per chunk and all-reduced, quality sums from column batches
            G0 = s.h.ritz_gram()
            s.h.set_tuning(_capi.TUNE_RITZ_CHUNK_ROWS, 512)
            s.get_H_eigs()
            Yc, Gc, qc = s.H_eigvecs_local, s.h.ritz_gram(), s.ritz_quality()
            chunked = dict(rows=s.h.ritz_info()["chunk_rows"], dY=float(np.abs(Yc - Y).max()), dG=float(np.abs(Gc - G0).max()),
                           dq=float(np.abs(qc - qual).max()), basis_ok=bool(np.array_equal(s.V_local, V)))
            s.h.set_tuning(_capi.TUNE_RITZ_CHUNK_ROWS, 0)
            s.get_H_eigs()
        # single-rank oracle on the full matrix
        full = build(0, M)
        full = full.to_scipy() if hasattr(full, "to_scipy") else __import__("scipy.sparse").sparse.csr_matrix(full)
        ao, bo, Vo = oracle.execute_lanczos(full, n, economy=True)
        th_o = np.linalg.eigvalsh(oracle.build_h_eff(ao, bo))
        S = np.linalg.eigh(s.H_eff)[1]
        dq = None
        if qual is not None:  # against NumPy on the assembled Ritz vectors
            Yfull = np.concatenate(boot.allgather_obj(Y), axis=0)
            Z = full @ Yfull
            qref = np.einsum("ri,ri->i", Z, Yfull) ** 2 / np.einsum("ri,ri->i", Z, Z)
            dq = float(np.abs(qual - qref).max() / np.abs(qref).max())
        # long runs outlive the prefix the reference arithmetic itself determines (converged Ritz values make the late
        # coefficients rounding noise, see oracle.stable_masks): compare coefficients on that prefix, Ritz values on the mask
        prefix, mask = (n, np.ones(n, bool)) if n < 100 or "_partial" in name else oracle.stable_masks(full, n, ao, bo)
        out[name] = dict(mode=s.plan.mode, da=float(np.abs(a - ao)[:prefix].max()), db=float(np.abs(bta - bo)[:max(prefix - 1, 1)].max()),
                         dth=float(np.abs(theta - th_o)[mask].max() / np.abs(th_o).max()), prefix=int(prefix), nmask=int(mask.sum()),
                         scale=float(max(np.abs(ao).max(), np.abs(bo).max())),
                         dV=float(np.abs(V[:, :8] - Vo[:8, lo:hi].T).max() / np.abs(Vo[:8]).max()), dY=float(np.abs(Y - V @ S).max()),
                         orth=float(np.abs(boot.allreduce_sum(V.T @ V) - np.eye(n)).max()),
                         comm_launches=comm_launches, sweeps=sweeps, misses=misses, engine=engine, n=n, device_built_equal=device_built_equal, dq=dq,
                         chunked=chunked)
    res = boot.allgather_obj(out)
    if boot.rank == 0:
        import json
        print("RESULT", json.dumps(res))
''')


@pytest.mark.parametrize("world", [2, 3])
def test_partitioned_run_on_one_gpu(tmp_path, world):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, LZ_ROOT=ROOT, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0", LZ_BOOT="socket" if world == 2 else "torch")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    import json

    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT")][0][len("RESULT"):])
    assert len(res) == world
    for per_rank in res:
        assert per_rank["lap2d"]["mode"] == "halo" and per_rank["lap3d"]["mode"] == "halo"
        assert per_rank["graph"]["mode"] == "allgather" and per_rank["graph_halo"]["mode"] == "halo"
        assert per_rank["dense"]["mode"] == "allgather" and per_rank["dense_poison"]["mode"] == "allgather"
        assert per_rank["c4_slab_k200"]["mode"] == "halo" and per_rank["c5_k500"]["mode"] == "halo"
        assert per_rank["c4_slab_k200"]["device_built_equal"] is True
        for name in ("lap2d", "graph", "dense", "lap3d"):  # lz_ritz_quality on the row-block partition (halo, all-gather, dense)
            assert per_rank[name]["dq"] is not None and per_rank[name]["dq"] < 1e-12, (name, per_rank[name])
        for name in ("lap2d", "graph"):  # the chunked Ritz mode on the partition (halo and all-gather exchange of the column batches)
            c = per_rank[name]["chunked"]
            assert c["rows"] == 512 and c["dY"] < 1e-13 and c["dG"] < 1e-13 and c["dq"] < 1e-12 and c["basis_ok"], (name, c)
        # (graph_twophase: same y bits as "graph", but alpha's partial sums are grouped by the kernel's own row blocks, so the
        # coefficients agree to rounding, not bit for bit: held to the same tolerances below)
        assert per_rank["c4_slab_k200"]["prefix"] >= 100 and per_rank["c5_k500"]["prefix"] >= 300, per_rank
        for name, r in per_rank.items():
            if "_partial" in name:
                # converging run with sweeps: Ritz values of the partial mode vs the oracle's full sweep
                assert 1 <= r["sweeps"] < r["n"] and r["dth"] < 1e-10 and r["dY"] < 1e-12, (name, r)
                assert r["sweeps"] == res[0][name]["sweeps"]
                if name.endswith("_onereduce"):
                    # ONE all-reduce + one exchange per iteration, sweep or no sweep: (n + 1) exchanges + n
                    # combined all-reduces + the last alpha - against exchange + three all-reduces per step in the device loop
                    assert r["engine"] == "partial-one-reduce" and r["misses"] == 0, (name, r)
                    assert r["comm_launches"] == 2 * r["n"] + 2, (name, r)
                    assert r["orth"] < 1e-6, (name, r)
                else:
                    assert r["engine"] == "partial-device" and r["comm_launches"] >= 4 * r["n"], (name, r)
                continue
            # short runs: 1e-11 absolute on every coefficient; the k = 200 / 500 runs: the north-star bar (1e-10 of the
            # spectral scale) on the stable prefix, whose end is by definition where coefficients start to move at 1e-12
            ctol = 1e-11 if r["n"] < 100 else 1e-10 * r["scale"]
            # dV: the first 8 basis vectors relative to their largest entry (1e-10, the same bar as the coefficients)
            assert r["da"] < ctol and r["db"] < ctol and r["dth"] < 1e-10 and r["dV"] < 1e-10 and r["dY"] < 1e-12, (name, r)
            assert r["comm_launches"] > 0 and r["sweeps"] == r["n"] and r["orth"] < 1e-12, (name, r)
        # collectives per run: default (fused norm) = (n + 1) exchanges + (n + 1) alpha all-reduces + n coefficient
        # all-reduces; one-reduce = (n + 1) exchanges + n combined all-reduces + the last alpha
        assert per_rank["lap2d"]["comm_launches"] == 3 * 40 + 2 and per_rank["lap2d_onereduce"]["comm_launches"] == 2 * 40 + 2, per_rank


def test_rccl_single_rank_communicator():
    """dlopen(librccl), ncclCommInitRank, ncclAllReduce / ncclAllGather on the compute stream with world = 1
    (forced through the tuning knob): the result must equal the plain single-rank run bit for bit."""
    A = synthetic.laplacian_2d_5pt(64, 48)
    M = A.shape[0]
    v0 = np.random.RandomState(99).uniform(-1, 1, M)
    v0 /= np.linalg.norm(v0)
    h0 = _capi.Handle(0)
    h0.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
    a0, b0 = h0.run(20, v0)
    h = _capi.Handle(0)
    h.comm_init_rccl(1, 0, h.unique_id())
    h.set_tuning(_capi.TUNE_FORCE_COLLECTIVES, 1)  # issue the collectives although world == 1
    h.set_csr(M, 0, A.rowptr, A.colidx, A.vals, ncols_ext=h.padded_rows(M))
    h.set_allgather(h.padded_rows(M))
    a1, b1 = h.run(20, v0)
    assert np.array_equal(a0, a1) and np.array_equal(b0, b1)
    assert h.timings()["comm"]["launches"] >= 60
    # lz_ritz_quality: the row-block form (every Ritz vector exchanged over RCCL and multiplied like a Lanczos vector, one
    # all-reduce of the 2 n sums) against the single-rank fused kernel; the basis row it borrows comes back unchanged
    S = np.linalg.eigh(np.diag(a0) + np.diag(b0, 1) + np.diag(b0, -1))[1]
    h0.ritz_vectors(S, fetch=False)
    h.ritz_vectors(S, fetch=False)
    V_before = h.get_basis()
    q0, q1 = h0.ritz_quality(), h.ritz_quality()
    assert np.abs(q1 - q0).max() <= 1e-13 * np.abs(q0).max()
    assert np.array_equal(h.get_basis(), V_before)
    h.close()
    h0.close()


def test_partial_loop_over_a_one_rank_rccl_communicator():
    """The device-decided partial re-orthogonalisation loop on a partition issues its collectives EVERY step (the host cannot
    skip a collective the device may need): alpha, ||r||^2 and - used or not - the coefficient vector.  Run with real RCCL calls
    (1-rank communicator, forced through the tuning knob) on a matrix whose Ritz values converge (several sweeps): coefficients,
    basis and sweep count equal to the plain single-rank run bit for bit, still no host synchronisation inside lz_run."""
    A = synthetic.laplacian_3d_7pt(20, 18, 16)
    M = A.shape[0]
    n = 120
    v0 = np.random.RandomState(99).uniform(-1, 1, M)
    v0 /= np.linalg.norm(v0)
    h0 = _capi.Handle(0)
    h0.set_options(_capi.FLAG_REORTH_PARTIAL)
    h0.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
    a0, b0 = h0.run(n, v0)
    V0, s0 = h0.get_basis(), h0.last_sweeps()
    h = _capi.Handle(0)
    h.comm_init_rccl(1, 0, h.unique_id())
    h.set_tuning(_capi.TUNE_FORCE_COLLECTIVES, 1)
    h.set_options(_capi.FLAG_REORTH_PARTIAL)
    h.set_csr(M, 0, A.rowptr, A.colidx, A.vals, ncols_ext=h.padded_rows(M))
    h.set_allgather(h.padded_rows(M))
    a1, b1 = h.run(n, v0)
    assert h.last_engine() == "partial-device" and h.last_host_syncs() == 0
    assert 1 < s0 < n and h.last_sweeps() == s0
    assert np.array_equal(a0, a1) and np.array_equal(b0, b1) and np.array_equal(V0, h.get_basis())
    assert h.timings()["comm"]["launches"] >= 4 * n  # all-gather + three all-reduces per step
    h.close()
    h0.close()


def test_partial_one_reduce_loop_over_a_one_rank_rccl_communicator():
    """The one-reduce partial loop (LZ_FLAG_REORTH_PARTIAL | LZ_FLAG_ONE_REDUCE) with real RCCL calls (1-rank communicator, forced
    through the tuning knob), on a matrix whose Ritz values converge (several sweeps): ONE ncclAllReduce + one exchange per step -
    2 n + 2 collectives per run against more than 4 n of the three-collective loop -, no host synchronisation, and results equal to
    the same loop without a communicator bit for bit."""
    A = synthetic.laplacian_3d_7pt(20, 18, 16)
    M = A.shape[0]
    n = 120
    v0 = np.random.RandomState(99).uniform(-1, 1, M)
    v0 /= np.linalg.norm(v0)
    flags = _capi.FLAG_REORTH_PARTIAL | _capi.FLAG_ONE_REDUCE
    h0 = _capi.Handle(0)
    h0.set_options(flags)
    h0.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
    a0, b0 = h0.run(n, v0)
    V0, s0 = h0.get_basis(), h0.last_sweeps()
    h = _capi.Handle(0)
    h.comm_init_rccl(1, 0, h.unique_id())
    h.set_tuning(_capi.TUNE_FORCE_COLLECTIVES, 1)
    h.set_options(flags)
    h.set_csr(M, 0, A.rowptr, A.colidx, A.vals, ncols_ext=h.padded_rows(M))
    h.set_allgather(h.padded_rows(M))
    a1, b1 = h.run(n, v0)
    assert h.last_engine() == "partial-one-reduce" and h.last_host_syncs() == 0 and h.last_sweep_misses() == 0
    assert 1 < s0 < n and h.last_sweeps() == s0
    assert np.array_equal(a0, a1) and np.array_equal(b0, b1) and np.array_equal(V0, h.get_basis())
    assert h.timings()["comm"]["launches"] == 2 * n + 2  # (n + 1) all-gathers + n combined all-reduces + the last alpha
    h.close()
    h0.close()


@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("faces", ["both", "one"])
def test_rccl_self_send_recv_halo(faces, overlap):
    """ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on the compute stream, with a 1-rank communicator
    exchanging with itself: the vertical wrap-around neighbours of a periodic 2-D stencil are routed through the
    ghost tail (packed by k_gather, sent and received by RCCL) instead of being read in place.  Must equal the
    plain run bit for bit."""
    nx, ny = 64, 48
    A = synthetic.laplacian_2d_5pt(nx, ny)
    M = A.shape[0]
    v0 = np.random.RandomState(99).uniform(-1, 1, M)
    v0 /= np.linalg.norm(v0)
    h0 = _capi.Handle(0)
    h0.set_csr(M, 0, A.rowptr, A.colidx, A.vals)
    a0, b0 = h0.run(25, v0)
    V0 = h0.get_basis()

    rows_pad = (M + 31) // 32 * 32
    row_of = np.repeat(np.arange(M), np.diff(A.rowptr))
    wrap = np.abs(A.colidx.astype(np.int64) - row_of) > nx  # entries that cross the periodic seam in y
    if faces == "one":  # a single contiguous face: sent straight out of V[j] without the pack kernel
        wrap &= A.colidx < nx
    ghost_cols = np.unique(A.colidx[wrap])
    col = A.colidx.astype(np.int64).copy()
    col[wrap] = rows_pad + np.searchsorted(ghost_cols, A.colidx[wrap])
    h = _capi.Handle(0)
    h.comm_init_rccl(1, 0, h.unique_id())
    h.set_tuning(_capi.TUNE_FORCE_COLLECTIVES, 1)
    if overlap:  # LZ_FLAG_OVERLAP_HALO: faces updated first, exchanged on a second stream behind the interior update
        h.set_options(_capi.FLAG_OVERLAP_HALO)
    h.set_csr(M, 0, A.rowptr, col.astype(np.int32), A.vals, ncols_ext=rows_pad + len(ghost_cols))
    if faces == "both":  # two contiguous faces, like the two slab neighbours of a rank: two (self) peer entries
        h.set_halo([0, 0], [nx, nx], ghost_cols.astype(np.int32), [nx, nx])
    else:
        h.set_halo([0], [len(ghost_cols)], ghost_cols.astype(np.int32), [len(ghost_cols)])
    a1, b1 = h.run(25, v0)
    assert len(ghost_cols) == (2 * nx if faces == "both" else nx)
    assert np.array_equal(a0, a1) and np.array_equal(b0, b1)
    assert np.array_equal(V0, h.get_basis())
    assert h.timings()["comm"]["launches"] >= 3 * 25
    h.close()
    h0.close()
