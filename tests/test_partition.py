"""Row-block partition planning (pure NumPy) and the N > 1 path over a real 2-rank gloo
process group on CPU: each rank runs the oracle's partitioned recurrence with its SpMV input
assembled exactly as the device exchange would (halo lists / all-gather) and its sums reduced
with torch.distributed; the result must equal the single-rank oracle."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from lanczos_amd import partition, synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_row_bounds():
    assert partition.row_bounds(10_000_000, 8) == [min(10_000_000, p * 1250016) for p in range(9)]
    assert partition.row_bounds(100, 1) == [0, 100]
    b = partition.row_bounds(1000, 3)
    assert b[0] == 0 and b[-1] == 1000 and all(x % 32 == 0 for x in b[:-1])


@pytest.mark.parametrize("world", [2, 3, 4])
@pytest.mark.parametrize("build,mode", [
    (lambda: synthetic.laplacian_2d_5pt(40, 30), "halo"),
    (lambda: synthetic.laplacian_3d_7pt(12, 11, 10), "halo"),
    (lambda: synthetic.random_graph_laplacian(3000, 9000, seed=5), "auto"),
    (lambda: synthetic.random_graph_laplacian(3000, 9000, seed=5), "halo"),
])
def test_plans_reproduce_the_global_spmv(world, build, mode):
    A = build()
    S = A.to_scipy()
    M = S.shape[0]
    x = np.random.default_rng(0).standard_normal(M)
    y = S * x
    bounds = partition.row_bounds(M, world)
    plans = []
    for r in range(world):
        loc = A.row_slice(bounds[r], bounds[r + 1])
        plans.append((loc, partition.plan_exchange(loc.rowptr, loc.colidx, M, world, r, mode)))
    gathered = [(p.peers, p.send_counts, p.recv_counts) for _, p in plans]
    for r, (loc, p) in enumerate(plans):
        partition.check_plans(p, r, gathered)
        lo, hi = bounds[r], bounds[r + 1]
        if p.mode == "halo":
            # what the peers send (their send_idx, peer-major) must be exactly my ghost tail, in order
            recv = []
            for q in p.peers:
                _, pq = plans[int(q)]
                off = np.concatenate([[0], np.cumsum(pq.send_counts)])
                k = int(np.searchsorted(pq.peers, r))
                recv.append(x[bounds[int(q)] + pq.send_idx[off[k]: off[k + 1]]])
            ghost = np.concatenate(recv) if recv else np.zeros(0)
            assert np.array_equal(ghost, x[p.ghost_cols])
            xe = np.zeros(p.ncols_ext)
            xe[: hi - lo] = x[lo:hi]
            xe[p.rows_pad:] = ghost
        else:
            assert p.mode == "allgather"
            xe = np.zeros(p.ncols_ext)
            for q in range(world):
                xe[q * p.chunk: q * p.chunk + bounds[q + 1] - bounds[q]] = x[bounds[q]: bounds[q + 1]]
        import scipy.sparse

        yl = scipy.sparse.csr_matrix((loc.vals, p.colidx, loc.rowptr), shape=(hi - lo, p.ncols_ext)) * xe
        assert np.array_equal(yl, y[lo:hi])
    if mode == "halo" and "lap" in repr(build.__code__.co_consts):
        pass


def test_stencil_halo_is_two_faces():
    A = synthetic.laplacian_2d_5pt(64, 64)
    loc = A.row_slice(*partition.row_bounds(4096, 4)[1:3])
    p = partition.plan_exchange(loc.rowptr, loc.colidx, 4096, 4, 1)
    assert p.mode == "halo" and list(p.peers) == [0, 2] and list(p.recv_counts) == [64, 64] and list(p.send_counts) == [64, 64]


def test_asymmetric_structure_is_detected():
    import scipy.sparse

    S = synthetic.laplacian_2d_5pt(8, 8).to_scipy().tolil()
    S[0, 63] = 0.5  # entry without its transpose partner... (0,63) already exists via wrap; use a fresh one
    S[3, 40] = 0.5
    S = S.tocsr()
    bounds = partition.row_bounds(64, 2)
    plans = []
    for r in range(2):
        blk = S[bounds[r]: bounds[r + 1]]
        plans.append(partition.plan_exchange(blk.indptr, blk.indices, 64, 2, r, "halo"))
    gathered = [(p.peers, p.send_counts, p.recv_counts) for p in plans]
    with pytest.raises(ValueError, match="not structurally symmetric"):
        for r, p in enumerate(plans):
            partition.check_plans(p, r, gathered)


WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    sys.path.insert(0, os.environ["LZ_ROOT"]); sys.path.insert(0, os.path.join(os.environ["LZ_ROOT"], "tests"))
    from lanczos_amd import partition, synthetic, distributed
    from oracle import lanczos_ref as oracle
    import scipy.sparse
    boot = distributed.TorchBootstrap()
    rank, world = boot.rank, boot.world
    out = {}
    for name, A, mode, n in [("lap2d", synthetic.laplacian_2d_5pt(48, 40), "halo", 30),
                             ("graph", synthetic.random_graph_laplacian(2500, 8000, seed=9), "auto", 25)]:
        S = A.to_scipy(); M = S.shape[0]
        bounds = partition.row_bounds(M, world); lo, hi = bounds[rank], bounds[rank + 1]
        loc = A.row_slice(lo, hi)
        plan = partition.plan_exchange(loc.rowptr, loc.colidx, M, world, rank, mode)
        if plan.mode == "halo":
            partition.check_plans(plan, rank, boot.allgather_obj((plan.peers, plan.send_counts, plan.recv_counts)))
        Bl = scipy.sparse.csr_matrix((loc.vals, plan.colidx, loc.rowptr), shape=(hi - lo, plan.ncols_ext))
        soff = np.concatenate([[0], np.cumsum(plan.send_counts)]).astype(int)
        def spmv_local(x):
            if plan.mode == "halo":
                segs = [x[plan.send_idx[soff[i]:soff[i + 1]]] for i in range(len(plan.peers))]
                recv = boot.exchange(plan.peers, segs, plan.recv_counts)
                xe = np.zeros(plan.ncols_ext); xe[:hi - lo] = x
                if recv: xe[plan.rows_pad:] = np.concatenate(recv)
            else:
                pad = np.zeros(plan.chunk); pad[:hi - lo] = x
                xe = boot.allgather_array(pad)
            return Bl * xe
        a, b, V = oracle.execute_lanczos_partitioned(S, n, (lo, hi), allreduce=lambda v: boot.allreduce_sum(np.array(v, dtype=np.float64)), spmv_local=spmv_local)
        a1, b1, V1 = oracle.execute_lanczos(S, n, economy=True)
        out[name] = (float(np.abs(a - a1).max()), float(np.abs(b - b1).max()), float(np.abs(V[:8] - V1[:8, lo:hi]).max()), plan.mode)
        # the ONE-collective-per-step loops (round 5: LZ_FLAG_ONE_REDUCE, and with LZ_FLAG_REORTH_PARTIAL the look-ahead gate) on the same
        # partition: every rank must take the same sweep decisions (they come from reduced sums) and issue n + 1 all-reduces
        for partial in (False, True):
            ao, bo, Vo, gates, calls = oracle.execute_lanczos_one_reduce(S, n, (lo, hi), allreduce=lambda v: boot.allreduce_sum(np.array(v, dtype=np.float64)),
                                                                         spmv_local=spmv_local, partial=partial)
            a2, b2, V2, g2, c2 = oracle.execute_lanczos_one_reduce(S, n, [0, M], partial=partial)  # the same loop on one rank
            out[name + ("_partial_onereduce" if partial else "_onereduce")] = (
                float(np.abs(ao - a1).max()), float(np.abs(bo - b1).max()), float(np.abs(ao - a2).max()), [int(x) for x in gates], [int(x) for x in g2], calls, n)
    res = boot.allgather_obj(out)
    if rank == 0:
        import json
        print("RESULT", json.dumps(res))
''')


def test_two_rank_gloo_partitioned_lanczos(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, LZ_ROOT=ROOT, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")][0]
    import json

    res = json.loads(line[len("RESULT"):])
    for per_rank in res:
        assert per_rank["lap2d"][3] == "halo" and per_rank["graph"][3] == "allgather"
        for name in ("lap2d", "graph"):
            da, db, dv, _ = per_rank[name]
            assert da < 1e-11 and db < 1e-11 and dv < 1e-9, (name, per_rank[name])
            for suffix in ("_onereduce", "_partial_onereduce"):
                da, db, d1, gates, gates1, calls, n = per_rank[name + suffix]
                assert da < 1e-11 and db < 1e-11 and d1 < 1e-11, (name + suffix, da, db, d1)
                assert calls == n + 1  # ONE all-reduce per step + the last alpha
                assert gates == gates1 == res[0][name + suffix][3]  # the same decisions on every rank, and as on one rank
                assert gates[0] == 1 and (sum(gates) == n if suffix == "_onereduce" else sum(gates) < n)


def _sock_worker(rank, world, key, q):
    import numpy as np

    from lanczos_amd import distributed

    b = distributed.SocketBootstrap(rank=rank, world=world, key=key, timeout=60)
    out = {}
    out["gather"] = b.allgather_obj(("r", rank))
    out["bcast"] = b.broadcast_bytes(b"uid" if rank == 0 else None)
    out["sum"] = b.allreduce_sum(np.arange(4, dtype=np.float64) + rank).tolist()
    peers = [p for p in range(world) if p != rank]
    got = b.exchange(peers, [np.full(2, 10.0 * rank + p) for p in peers], [2] * len(peers))
    out["xchg"] = [g.tolist() for g in got]
    out["ag"] = b.allgather_array(np.full(2, float(rank))).tolist()
    b.barrier()
    q.put((rank, out))


def test_socket_bootstrap_collectives():
    import multiprocessing as mp
    import uuid

    world, key = 3, uuid.uuid4().hex[:12]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_sock_worker, args=(r, world, key, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for r in range(world):
        o = res[r]
        assert o["gather"] == [("r", i) for i in range(world)] and o["bcast"] == b"uid"
        assert o["sum"] == [3.0 + 3 * i for i in range(4)]
        peers = [p for p in range(world) if p != r]
        assert o["xchg"] == [[10.0 * p + r] * 2 for p in peers]
        assert o["ag"] == [0.0, 0.0, 1.0, 1.0, 2.0, 2.0]


@pytest.mark.parametrize("world", [2, 4, 8])
def test_headline_partition_plan(world):
    """The M = 1e7 headline grid (4000 x 2500) split over 2/4/8 ranks: every rank exchanges exactly one grid row
    (4000 doubles = 32 KB) with each of its two slab neighbours (periodic wrap makes rank 0 and P-1 neighbours)."""
    nx, ny = 4000, 2500
    M = nx * ny
    b = partition.row_bounds(M, world)
    assert b[-1] == M and all((b[i + 1] - b[i]) > 0 for i in range(world))
    for r in (0, world // 2, world - 1):
        loc = synthetic.laplacian_2d_5pt(nx, ny, rows=(b[r], b[r + 1]))
        p = partition.plan_exchange(loc.rowptr, loc.colidx, M, world, r)
        assert p.mode == "halo"
        expect_peers = sorted({(r - 1) % world, (r + 1) % world})
        assert list(p.peers) == expect_peers
        total = 2 * nx
        assert int(p.recv_counts.sum()) == total and int(p.send_counts.sum()) == total
        assert p.ncols_ext == p.rows_pad + total
        assert p.colidx.max() < p.ncols_ext and p.colidx.min() >= 0


@pytest.mark.parametrize("points", [7, 27])
@pytest.mark.parametrize("dims,world", [((12, 11, 10), 2), ((12, 11, 10), 3), ((9, 8, 16), 8), ((20, 18, 16), 5)])
def test_stencil_slab_plan_equals_the_matrix_based_plan(dims, world, points):
    """partition.plan_stencil_slab plans a slab's halo from its boundary rows alone (the device then assembles the slab's
    CSR itself, lz_build_stencil3d_block): same peers, counts, send lists and ghost tail as plan_exchange on the host-built
    row block, and the ghost tail is a handful of contiguous global ranges."""
    M = int(np.prod(dims))
    cols = np.sort(partition.stencil3d_columns(dims, points, np.arange(M)), axis=1)
    if points == 7:
        A = synthetic.laplacian_3d_7pt(*dims)
        assert np.array_equal(cols.reshape(-1), A.colidx)
    b = partition.row_bounds(M, world)
    for r in range(world):
        lo, hi = b[r], b[r + 1]
        p0 = partition.plan_exchange(np.arange(hi - lo + 1) * points, cols[lo:hi].reshape(-1), M, world, r, "halo")
        p1, ranges = partition.plan_stencil_slab(dims, points, world, r)
        for k in ("peers", "send_counts", "send_idx", "recv_counts", "ghost_cols"):
            assert np.array_equal(getattr(p0, k), getattr(p1, k)), (k, r)
        assert p0.ncols_ext == p1.ncols_ext and p1.colidx is None
        assert np.array_equal(np.concatenate([np.arange(s, s + n) for s, n in ranges]), p0.ghost_cols) and len(ranges) <= 16
