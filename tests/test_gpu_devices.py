"""The multi-GPU path behind the DROP-IN class surface: ``Lanczos(H).execute_Lanczos(n)`` with ``Lanczos.devices`` set
(/root/reference/Python/Regular/Lanczos.py:19-26,75-141 is single-process; SURVEY.md section 5 "Config / flags" names the
``devices=`` hook).  The test box has ONE GPU and RCCL refuses two ranks per device, so the workers share GPU 0 and the
collectives are staged through the host (``comm_backend = "host"``); every kernel and every line of the partition logic is
the one the 8-GPU run uses.  Held to: the golden fixtures (1e-10 bar), and ``np.array_equal`` against
``distributed.DistributedLanczos`` driven directly with the same number of ranks."""
import json
import os
import subprocess
import sys
import textwrap
import uuid

import numpy as np
import pytest

from conftest import load_golden
from lanczos_amd import Hamiltonian, IrrLanczos, Lanczos, StencilOperator, synthetic
from oracle import lanczos_ref as oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# one rank of the direct DistributedLanczos run the pool is compared with (same world, same backend)
DIRECT = textwrap.dedent('''
    import os, sys
    import numpy as np
    sys.path.insert(0, os.environ["LZ_ROOT"])
    from lanczos_amd import distributed, partition, synthetic
    boot = distributed.SocketBootstrap()
    cases = {
        "lap2d": (lambda lo, hi: synthetic.laplacian_2d_5pt(96, 80, rows=(lo, hi)), 96 * 80, 40),
        "c4_slab": (lambda lo, hi: synthetic.laplacian_3d_7pt(24, 24, 32, rows=(lo, hi)), 24 * 24 * 32, 200),
        "c5": (lambda lo, hi: synthetic.laplacian_2d_5pt(160, 120, rows=(lo, hi)), 160 * 120, 500),
        "graph": (lambda lo, hi: synthetic.random_graph_laplacian(6000, 20000, seed=4).row_slice(lo, hi), 6000, 30),
    }
    out = {}
    for name, (build, M, n) in cases.items():
        b = partition.row_bounds(M, boot.world)
        lo, hi = b[boot.rank], b[boot.rank + 1]
        s = distributed.DistributedLanczos(build(lo, hi), M, boot, device_id=0, backend="host")
        a, bt = s.execute_Lanczos(n)
        s.get_H_eigs()
        V = np.concatenate(boot.allgather_obj(s.V_local), axis=0)
        Y = np.concatenate(boot.allgather_obj(s.H_eigvecs_local), axis=0)
        if boot.rank == 0:
            np.savez(os.path.join(os.environ["LZ_OUT"], name + ".npz"), alpha=a, beta=bt, V=V, Y=Y, theta=s.H_eigvals)
        s.h.close()
    boot.barrier()
''')


def _direct(tmp_path, world):
    script = tmp_path / "direct.py"
    script.write_text(DIRECT)
    key = uuid.uuid4().hex[:12]
    procs = []
    for r in range(world):
        env = dict(os.environ, LZ_ROOT=ROOT, LZ_OUT=str(tmp_path), RANK=str(r), WORLD_SIZE=str(world), LZ_RDZV_KEY=key, OMP_NUM_THREADS="2",
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        out, err = p.communicate(timeout=900)
        assert p.returncode == 0, out[-2000:] + err[-3000:]
    return {n: dict(np.load(tmp_path / (n + ".npz"))) for n in ("lap2d", "c4_slab", "c5", "graph")}


@pytest.mark.parametrize("world", [2, 3])
def test_class_surface_with_devices_equals_distributed_lanczos(tmp_path, world):
    ref = _direct(tmp_path, world)
    Lanczos.verbose = IrrLanczos.verbose = False
    inputs = {
        "lap2d": (synthetic.laplacian_2d_5pt(96, 80).to_scipy(), 40),
        "c4_slab": (synthetic.laplacian_3d_7pt(24, 24, 32).to_scipy(), 200),  # BASELINE C4 (z-slab partition, k = 200) at reduced size
        "c5": (synthetic.laplacian_2d_5pt(160, 120).to_scipy(), 500),         # BASELINE C5 (k = 500) at reduced size
        "graph": (synthetic.random_graph_laplacian(6000, 20000, seed=4).to_scipy(), 30),
    }
    for name, (H, n) in inputs.items():
        cls = IrrLanczos if name == "graph" else Lanczos  # the Irregular facade goes through the same pool
        s = cls(H)
        s.devices = [0] * world
        s.comm_backend = "host"
        (s.execute_LanczosOld if name == "graph" else s.execute_Lanczos)(n)
        d = ref[name]
        assert s._handle.world == world and s._handle.exchange_mode() == ("allgather" if name == "graph" else "halo")
        assert np.array_equal(np.diag(s.H_eff), d["alpha"]) and np.array_equal(np.diag(s.H_eff, 1), d["beta"]), name
        assert np.array_equal(s.H_eff, s.H_eff.T) and s.H_eff.shape == (n, n)
        assert np.array_equal(s.H_eigvals, d["theta"]), name
        assert s.V.shape == (H.shape[0], n) and np.array_equal(s.V, d["V"]), name
        # H_eigvecs = V S with S = eigh(H_eff) taken in THIS process (LAPACK's S is not bit-reproducible across processes with
        # different thread counts: signs and last bits of S differ, the device GEMM does not)
        S = np.linalg.eigh(s.H_eff)[1]
        Y = s.H_eigvecs
        assert Y.shape == (H.shape[0], n) and np.abs(Y - d["V"] @ S).max() <= 1e-13, name
        sign = np.sign(np.einsum("ri,ri->i", Y, d["Y"]))
        gaps = np.minimum(np.diff(d["theta"], prepend=-np.inf), np.diff(d["theta"], append=np.inf))
        iso = gaps > 1e-6 * np.abs(d["theta"]).max()  # (a Ritz vector is only determined where its Ritz value is isolated)
        assert np.abs(Y * sign - d["Y"])[:, iso].max() <= 1e-8, name
        # windows of V / H_eigvecs that straddle a rank boundary
        lo, hi = H.shape[0] // world - 7, H.shape[0] // world + 41
        assert np.array_equal(s.V_rows(lo, hi), d["V"][lo:hi]) and np.array_equal(s.H_eigvecs_rows(lo, hi), Y[lo:hi])
        if name == "lap2d":  # print_good_eigs' quality sums as a collective, against NumPy on the gathered Ritz vectors
            q = s._eigvec_quality()
            Z = H @ Y
            qref = np.einsum("ri,ri->i", Z, Y) ** 2 / np.einsum("ri,ri->i", Z, Z)
            assert np.abs(q - qref).max() <= 1e-12 * np.abs(qref).max()
            uploads = s._handle.matrix_uploads
            s.execute_Lanczos(n)  # same H: the workers keep their row blocks
            assert s._handle.matrix_uploads == uploads and np.array_equal(np.diag(s.H_eff), d["alpha"])
        # against the CPU oracle on the whole matrix (north-star bar on the prefix the reference arithmetic determines)
        ao, bo, _ = oracle.execute_lanczos(H, n, economy=True)
        prefix, mask = (n, np.ones(n, bool)) if n < 100 else oracle.stable_masks(H, n, ao, bo)
        scale = max(np.abs(ao).max(), np.abs(bo).max())
        assert np.abs(d["alpha"] - ao)[:prefix].max() <= 1e-10 * scale and np.abs(d["beta"] - bo)[: prefix - 1].max() <= 1e-10 * scale, name
        th = np.linalg.eigvalsh(oracle.build_h_eff(ao, bo))
        assert np.abs(s.H_eigvals - th)[mask].max() <= 1e-10 * np.abs(th).max(), name
        s.close()


@pytest.mark.parametrize("name", ["lap2d_32x32_n30", "lap3d_8x8x8_n40", "graph_M2000_E7000_n40", "c1_dense512_n20", "deuteron3d_N12_27pt_n100",
                                  "ragged_M700_n25", "lap2d_8x8_n2"])
def test_golden_fixtures_through_two_workers(name):
    """the reference's own outputs (tests/golden, generated by importing the reference) through ``devices = [0, 0]``: the
    same comparison tests/test_gpu_lanczos.py::test_golden makes on one GPU"""
    d, H = load_golden(name)
    n, seed = int(d["n"]), int(d["seed"])
    v0 = d["v0"] if "v0" in d else None
    Lanczos.verbose = False
    s = Lanczos(H.toarray() if name.startswith("c1_dense") else H)
    s.devices = [0, 0]
    s.comm_backend = "host"
    s.execute_Lanczos(n, seed=seed, v0=v0)
    alpha, beta = np.diag(s.H_eff), np.diag(s.H_eff, 1)
    scale = max(np.abs(d["alpha"]).max(), np.abs(d["beta"]).max())
    prefix, mask = oracle.stable_masks(H, n, d["alpha"], d["beta"], seed=seed, v0=v0)
    assert np.abs(alpha - d["alpha"])[:prefix].max() <= 1e-10 * scale
    assert np.abs(beta - d["beta"])[: max(prefix - 1, 1)].max() <= 1e-10 * scale
    if prefix == n:
        assert np.abs(s.H_eigvals - d["H_eigvals"]).max() <= 1e-10 * np.abs(d["H_eigvals"]).max()
    else:
        conv = oracle.converged_ritz(d["alpha"], d["beta"])
        nearest = np.abs(s.H_eigvals[None, :] - conv[:, None]).min(axis=1)
        assert nearest.max() <= 1e-10 * np.abs(d["H_eigvals"]).max()
    assert s.V.shape == (int(d["M"]), n) and s.H_eigvecs.shape == (int(d["M"]), n)
    if "V" in d:
        vrows = oracle.stable_basis_rows(H, n, d["V"], seed=seed, v0=v0)
        ref = d["V"][:vrows].T
        assert np.abs(s.V[:, :vrows] - ref).max() <= 1e-10 * np.abs(ref).max()
    s.close()


def test_stencil_operator_on_one_and_on_three_workers():
    """``Hamiltonian.operator()``: the matrix is assembled on the device(s) - each worker its own slab - and never exists on
    the host; same entries as ``build_H`` (the reference builder's bits), same run."""
    Hamiltonian.verbose = Lanczos.verbose = False
    N, n = 14, 60
    os.makedirs("T_matrices", exist_ok=True)
    ham = Hamiltonian(N, 25, synthetic.deuteron_potential, 197.327**2 / (2 * 469.4592) / (25.0 / N) ** 2)
    H = ham.build_H("27")
    op = ham.operator("27")
    assert isinstance(op, StencilOperator) and op.shape == H.shape
    Hop = op.to_scipy()
    assert (Hop != H).nnz == 0 and np.array_equal(Hop.indices, H.indices)
    s0 = Lanczos(H)
    s0.execute_Lanczos(n, seed=78)
    s1 = Lanczos(op)
    s1.execute_Lanczos(n, seed=78)
    assert np.array_equal(s0.H_eff, s1.H_eff) and np.array_equal(s0.V, s1.V)
    s3 = Lanczos(op)
    s3.devices = [0, 0, 0]
    s3.comm_backend = "host"
    s3.execute_Lanczos(n, seed=78)
    H3, V3, th3 = s3.H_eff.copy(), s3.V.copy(), s3.H_eigvals.copy()
    s3.close()  # (the box allows six processes on its GPU: one pool of three workers at a time next to this process)
    sH = Lanczos(H)
    sH.devices = [0, 0, 0]
    sH.comm_backend = "host"
    sH.execute_Lanczos(n, seed=78)
    assert np.array_equal(H3, sH.H_eff) and np.array_equal(V3, sH.V)  # slab assembled on the device == slab cut from the host matrix
    scale = np.abs(s0.H_eff).max()
    assert np.abs(H3 - s0.H_eff).max() <= 1e-10 * scale
    assert np.abs(th3 - s0.H_eigvals)[:4].max() <= 1e-10 * scale
    for s in (s0, s1, sH):
        s.close()
    # the potential evaluated INSIDE the assembly kernel (no N^3 host array at all): one device vs two workers
    ham2 = Hamiltonian(N, 25, synthetic.DeuteronPotential(), 197.327**2 / (2 * 469.4592) / (25.0 / N) ** 2)
    ham2.device_potential = True
    op2 = ham2.operator("27")
    assert op2.potential is None and op2.potential_params is not None
    a = Lanczos(op2)
    a.execute_Lanczos(n, seed=78)
    b = Lanczos(op2)
    b.devices = [0, 0]
    b.comm_backend = "host"
    b.execute_Lanczos(n, seed=78)
    assert np.abs(a.H_eff - b.H_eff).max() <= 1e-10 * np.abs(a.H_eff).max()
    assert np.abs(a.H_eff - s0.H_eff).max() <= 1e-9 * scale  # (device exp/pow differ from NumPy's in the last bits: not the bit-exact path)
    a.close()
    b.close()


def test_checkpoint_and_resume_through_the_workers(tmp_path):
    """SURVEY.md section 5 hook through ``devices``: a checkpoint written by two workers is one host array per field (no trace of
    the partition), so three workers - or one GPU without workers - continue it; the result equals the uninterrupted run of the
    same world bit for bit, and the one-GPU continuation of the two-worker checkpoint equals ... the two-worker prefix it was
    given plus its own arithmetic (coefficients to the 1e-10 bar of the partition-vs-single comparison)."""
    H = synthetic.laplacian_2d_5pt(90, 70).to_scipy()
    n1, n2 = 12, 30
    Lanczos.verbose = False

    def solver(devs):
        s = Lanczos(H)
        s.devices = devs
        s.comm_backend = "host"
        return s

    whole = solver([0, 0])
    whole.execute_Lanczos(n2)
    first = solver([0, 0])
    first.execute_Lanczos(n1)
    path = str(tmp_path / "ck.npz")
    first.save_checkpoint(path)
    ck = first.checkpoint()
    assert ck["V"].shape == (n1, H.shape[0]) and ck["r"].shape == (H.shape[0],)
    assert np.array_equal(ck["V"], whole.V.T[:n1])
    first.close()
    second = solver([0, 0])
    second.resume_Lanczos(n2, path)
    assert np.array_equal(second.H_eff, whole.H_eff) and np.array_equal(second.V, whole.V)
    assert np.array_equal(second.H_eigvals, whole.H_eigvals)
    assert np.array_equal(second.checkpoint()["r"], whole.checkpoint()["r"])
    second.close()
    whole.close()
    # another partition, and no partition at all, continue the same file
    ref = Lanczos(H)
    ref.execute_Lanczos(n2)
    scale = np.abs(np.diag(ref.H_eff)).max()
    for devs in ([0, 0, 0], None):
        t = solver(devs)
        t.resume_Lanczos(n2, path)
        assert np.array_equal(np.diag(t.H_eff)[: n1 - 1], ck["alpha"][: n1 - 1])
        assert np.abs(t.H_eff - ref.H_eff).max() <= 1e-10 * scale
        assert np.abs(t.V - ref.V).max() <= 1e-10 * np.abs(ref.V).max()
        t.close()
    ref.close()


@pytest.mark.parametrize("world", [2, 3])
def test_one_collective_selective_loop_through_the_class_surface(world):
    """The north star's partition behind the drop-in surface with the selective re-orthogonalisation AND one all-reduce per step
    (round 5): `Lanczos.devices` + `reorth = "partial"` + `options = FLAG_ONE_REDUCE` - the workers run engine 8 (look-ahead gate from
    reduced sums).  Ritz values within 1e-10 of the single-GPU full sweep, the same (small) number of sweeps as the same loop on one
    GPU, an (only) semi-orthogonal basis."""
    from lanczos_amd import _capi

    H = synthetic.laplacian_3d_7pt(20, 18, 16).to_scipy()
    n = 120
    Lanczos.verbose = False
    full = Lanczos(H)
    full.execute_Lanczos(n)
    one = Lanczos(H)
    one.reorth = "partial"
    one.options = _capi.FLAG_ONE_REDUCE
    one.execute_Lanczos(n)
    assert one._get_handle().last_engine() == "partial-one-reduce" and 1 < one.sweeps < n // 4
    s = Lanczos(H)
    s.devices = [0] * world
    s.comm_backend = "host"
    s.reorth = "partial"
    s.options = _capi.FLAG_ONE_REDUCE
    s.execute_Lanczos(n)
    scale = np.abs(full.H_eigvals).max()
    assert np.abs(s.H_eigvals - full.H_eigvals).max() <= 1e-10 * scale
    assert np.abs(one.H_eigvals - full.H_eigvals).max() <= 1e-10 * scale
    assert s.sweeps == one.sweeps  # the decisions come from reduced sums: the partition does not change them (here: not even by rounding)
    V = s.V
    assert 1e-13 < np.abs(V.T @ V - np.eye(n)).max() < 1e-6
    for obj in (full, one, s):
        obj.close()


def test_bench_self_spawn_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` without a launcher: the parent (which never touches the GPU) spawns both ranks, rank 0
    prints ONE JSON line - the plumbing the driver's first multi-GPU run will go through."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "host", "--device", "0", "--workload", "lap2d_5pt_M1e6_k100",
           "--steps", "1", "--warmup", "1", "--prewarm-s", "0.1", "--arm-timeout", "600"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["comm"] == "host" and line["config"]["exchange"] == "halo"
    assert line["comm_calls_per_step"] == 3 * 100 + 2  # (k + 1) exchanges + (k + 1) alpha all-reduces + k coefficient all-reduces
    assert line["partial_reorth"] is not None and line["one_reduce_arm"] is not None  # both extra arms ran
    assert "stalled" not in line and line["value"] > 0 and line["roofline"]["kernel"] in ("qtw", "update", "spmv")
    # collectives per Lanczos iteration, by kind: the default loops - full sweep 2 all-reduces + 1 exchange, the
    # device-decided partial loop 3 + 1; the one-reduce arm - ONE all-reduce + one exchange for BOTH re-orthogonalisation modes
    cpi = line["comm_calls_per_iteration"]
    assert abs(cpi["allreduce"] - 2.0) < 0.02 and abs(cpi["exchange"] - 1.0) < 0.02, cpi
    cpp = line["partial_reorth"]["comm_calls_per_iteration"]
    assert line["partial_reorth"]["engine"] == "partial-device" and abs(cpp["allreduce"] - 3.0) < 0.03 and abs(cpp["exchange"] - 1.0) < 0.02, cpp
    arm = line["one_reduce_arm"]
    assert abs(arm["comm_calls_per_iteration"] - 2.0) < 0.03, arm
    pa = arm["partial"]
    assert pa["engine"] == "partial-one-reduce" and pa["lookahead_misses"] == 0 and pa["max_rel_ritz_diff_vs_full"] < 1e-10, pa
    assert abs(pa["comm_calls_per_iteration"]["allreduce"] - 1.0) < 0.02 and abs(pa["comm_calls_per_iteration"]["exchange"] - 1.0) < 0.02, pa


def test_a_dying_worker_ends_the_pool_and_is_reported():
    """A rank that dies mid-life (here: killed) must surface as an exception in the caller - its peers would otherwise sit in a
    collective forever - with every other worker terminated; a fresh `execute_Lanczos` on the object starts a new pool."""
    Lanczos.verbose = False
    H = synthetic.laplacian_2d_5pt(96, 80).to_scipy()
    s = Lanczos(H)
    s.devices = [0, 0]
    s.comm_backend = "host"
    s.execute_Lanczos(20)
    H_eff = s.H_eff.copy()
    pool = s._handle.pool
    pool.procs[1].kill()
    pool.procs[1].wait()
    import lanczos_amd

    with pytest.raises(lanczos_amd.LanczosHipError, match="rank 1"):
        s.execute_Lanczos(20)
    assert pool.closed and all(p.poll() is not None for p in pool.procs)
    s.close()
    s.execute_Lanczos(20)  # a new pool
    assert s._handle.pool is not pool and np.array_equal(s.H_eff, H_eff)
    s.close()


def test_stalled_worker_ends_the_call_within_its_deadline():
    """Round 4 (SURVEY section 5 "failure detection"): a worker that stalls - stopped with SIGSTOP here, which is what a rank
    stuck in a collective looks like from the caller - no longer hangs ``execute_Lanczos``: the command's deadline
    (``Lanczos.worker_timeout``, otherwise derived from the work) expires, the pool ends every child it started (by pid,
    SIGKILL for the silent rank), /dev/shm is left clean and the caller gets ``LanczosHipError`` naming the silent rank."""
    import glob
    import signal
    import time

    from lanczos_amd import LanczosHipError

    before = set(glob.glob("/dev/shm/lz_*"))
    Lanczos.verbose = False
    s = Lanczos(synthetic.laplacian_2d_5pt(64, 48).to_scipy())
    s.devices = [0, 0]
    s.comm_backend = "host"
    s.execute_Lanczos(20)  # the pool works
    ref = s.H_eff.copy()
    pool = s._handle.pool
    os.kill(pool.procs[1].pid, signal.SIGSTOP)
    s.worker_timeout = 5.0
    t = time.time()
    # (rank 0 sits in the first all-reduce waiting for its stopped peer: it is silent too)
    with pytest.raises(LanczosHipError, match=r"rank\(s\) \[(0, )?1\].*did not answer within 5 s"):
        s.execute_Lanczos(20)
    assert time.time() - t < 40.0
    assert pool.closed and all(p.poll() is not None for p in pool.procs)
    assert set(glob.glob("/dev/shm/lz_*")) == before
    # the object recovers: a fresh pool is started by the next call
    s.close()
    s.worker_timeout = None
    s.execute_Lanczos(20)
    assert np.array_equal(s.H_eff, ref)
    s.close()
