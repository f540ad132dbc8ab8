"""Multi-GPU execution behind the drop-in class surface: ``Lanczos.devices = [0, 1, ..., 7]``.

The reference is single-process, single-device (Python/Regular/Lanczos.py:19-26,75: ``Lanczos(H).execute_Lanczos(n)``;
SURVEY.md section 2 #12/#13: no parallelism, no communication layer).  To keep exactly that call surface while the Krylov
basis is row-block partitioned over the GPUs of a node, the calling process spawns one FRESH child process per device
(``python -m lanczos_amd._worker``; never a fork/exec of GPU state - the parent need not have touched a GPU at all),
hands each its row block through /dev/shm, and lets them run ``distributed.DistributedLanczos`` over the torch-free
``SocketBootstrap`` + RCCL.  ``PoolHandle`` below offers the subset of ``_capi.Handle`` that ``_solver.LanczosBase``
uses, so the class mirror is the same code for one GPU and for eight: ``H_eff`` / ``H_eigvals`` come back replicated,
``V`` / ``H_eigvecs`` are gathered lazily, the ``get_H_eigs`` checks and ``print_good_eigs`` run as collectives on the
device-resident Ritz vectors.
"""
from __future__ import annotations

import atexit
import json
import os
import select
import subprocess
import sys
import time
import uuid
import weakref

import numpy as np

from . import _capi, partition
from .distributed import _dec, _enc

_SHM_DIR = "/dev/shm"
_live_pools = weakref.WeakSet()


def _close_live_pools():
    for p in list(_live_pools):
        try:
            p.close()
        except Exception:
            pass


atexit.register(_close_live_pools)


def _copy_parallel(pairs, piece=16 << 20):
    """dst[...] = src for every pair, in pieces on a few threads (NumPy's copy loops release the GIL)"""
    jobs = []
    for dst, src in pairs:
        d, s_ = dst.reshape(-1), np.asarray(src).reshape(-1)
        step = max(1, piece // max(d.itemsize, 1))
        jobs += [(d, s_, o, min(o + step, d.size)) for o in range(0, d.size, step)]
    if len(jobs) <= 2:
        for d, s_, lo, hi in jobs:
            d[lo:hi] = s_[lo:hi]
        return
    from concurrent.futures import ThreadPoolExecutor

    def one(j):
        d, s_, lo, hi = j
        d[lo:hi] = s_[lo:hi]

    with ThreadPoolExecutor(max_workers=max(1, min(6, (os.cpu_count() or 2) // 2))) as ex:
        list(ex.map(one, jobs))


class _Shm:
    """A parent-owned file in /dev/shm holding one array; ``spec`` is what a worker needs to map it."""

    def __init__(self, key, name, shape, dtype):
        self.path = os.path.join(_SHM_DIR, f"lz_{key}_{name}_{uuid.uuid4().hex[:8]}")
        self.shape, self.dtype = tuple(int(x) for x in shape), np.dtype(dtype)
        nbytes = max(int(np.prod(self.shape)), 1) * self.dtype.itemsize
        fd = os.open(self.path, os.O_CREAT | os.O_EXCL | os.O_RDWR, 0o600)  # this user only: the matrix / the basis are in there
        try:
            os.ftruncate(fd, nbytes)
        finally:
            os.close(fd)
        self.arr = np.memmap(self.path, dtype=self.dtype, mode="r+", shape=self.shape if int(np.prod(self.shape)) else (1,))
        if not int(np.prod(self.shape)):
            self.arr = self.arr[:0].reshape(self.shape)

    @property
    def spec(self):
        return (self.path, self.dtype.str, list(self.shape))

    def unlink(self):
        try:
            os.unlink(self.path)
        except OSError:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.arr = None
        self.unlink()


class WorkerPool:
    """``len(devices)`` worker processes, rank r on GPU ``devices[r]``; ``request`` sends one command to every rank and
    returns the per-rank replies.  A dead or failing rank ends the whole pool (the others may sit in a collective)."""

    def __init__(self, devices, backend="rccl", start_timeout=300.0):
        self.devices = [int(d) for d in devices]
        self.world = len(self.devices)
        self.backend = backend
        self.key = uuid.uuid4().hex[:16]
        self.procs = []
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        for r, dev in enumerate(self.devices):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(self.world), LZ_RDZV_KEY=self.key, LZ_DEVICE=str(dev),
                       LZ_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                       PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
            self.procs.append(subprocess.Popen([sys.executable, "-m", "lanczos_amd._worker"], env=env, stdin=subprocess.PIPE,
                                               stdout=subprocess.PIPE, bufsize=0))
        self._buf = [b""] * self.world
        self.closed = False
        _live_pools.add(self)
        try:
            self._collect(start_timeout)  # the "ready" lines (rendezvous done, library loaded)
        except BaseException:
            self.kill()
            raise

    # ---- protocol ---------------------------------------------------------------------------------------------------------
    def _collect(self, timeout):
        """one reply line from every rank; raises (and ends the pool) on an error reply, a dead rank or a timeout"""
        replies = [None] * self.world
        deadline = None if timeout is None else time.time() + timeout
        fds = {p.stdout.fileno(): r for r, p in enumerate(self.procs)}
        while any(x is None for x in replies):
            for r, p in enumerate(self.procs):
                if replies[r] is None and p.poll() is not None and b"\n" not in self._buf[r]:
                    rest = p.stdout.read() or b""
                    self._buf[r] += rest
                    if b"\n" not in self._buf[r]:
                        self.kill()
                        raise _capi.LanczosHipError(-3, f"worker rank {r} (GPU {self.devices[r]}) exited with status {p.returncode}")
            wait = 0.5 if deadline is None else max(0.0, min(0.5, deadline - time.time()))
            ready, _, _ = select.select([fd for fd, r in fds.items() if replies[r] is None], [], [], wait)
            for fd in ready:
                r = fds[fd]
                chunk = os.read(fd, 1 << 20)
                self._buf[r] += chunk
            for r in range(self.world):
                if replies[r] is None and b"\n" in self._buf[r]:
                    line, self._buf[r] = self._buf[r].split(b"\n", 1)
                    replies[r] = _dec(json.loads(line.decode("utf-8")))
                    if "error" in replies[r]:
                        self.kill()
                        raise _capi.LanczosHipError(-3, f"worker rank {r} (GPU {self.devices[r]}): {replies[r]['error']}\n{replies[r].get('traceback', '')}")
            if deadline is not None and time.time() > deadline and any(x is None for x in replies):
                silent = [r for r, x in enumerate(replies) if x is None]
                self.kill(hard=silent)  # a stalled rank (stuck collective, stopped process) does not react to SIGTERM: SIGKILL, by pid
                raise _capi.LanczosHipError(-3, f"worker rank(s) {silent} (GPU {[self.devices[r] for r in silent]}) did not answer within "
                                                f"{timeout:.0f} s; the pool was terminated (Lanczos.worker_timeout overrides the deadline)")
        return replies

    def request(self, msg, timeout=60.0):
        """``timeout``: seconds every rank has to answer (``PoolHandle`` derives it from the work of the command; ``None`` waits
        for ever and only notices DEAD ranks).  On expiry every child this pool started is ended by pid - the silent ones with
        SIGKILL - and ``LanczosHipError`` names the silent ranks."""
        if self.closed:
            raise _capi.LanczosHipError(-4, "the worker pool is closed")
        data = (json.dumps(_enc(msg)) + "\n").encode("utf-8")
        for r, p in enumerate(self.procs):
            try:
                p.stdin.write(data)
                p.stdin.flush()
            except (BrokenPipeError, OSError):
                self.kill()
                raise _capi.LanczosHipError(-3, f"worker rank {r} (GPU {self.devices[r]}) is gone (exit status {p.poll()})")
        return self._collect(timeout)

    def close(self):
        if self.closed:
            return
        self.closed = True
        for p in self.procs:
            try:
                p.stdin.write((json.dumps(_enc({"cmd": "close"})) + "\n").encode("utf-8"))
                p.stdin.flush()
                p.stdin.close()
            except Exception:
                pass
        t_end = time.time() + 30.0
        for p in self.procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
            p.stdout.close()

    def kill(self, hard=()):
        """end every rank now (exact pids of the children this pool started); ranks in ``hard`` get SIGKILL at once"""
        self.closed = True
        for r, p in enumerate(self.procs):
            if p.poll() is None:
                if r in hard:
                    p.kill()
                else:
                    p.terminate()
        t_end = time.time() + 10.0
        for p in self.procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
            for f in (p.stdin, p.stdout):
                try:
                    f.close()
                except Exception:
                    pass


class StencilOperator:
    """Closed-form description of the periodic ``Nx x Ny x Nz`` 7-/27-point operator ``(+-) T_factor * stencil + diag`` that
    ``lanczos_amd.Hamiltonian`` assembles (Python/Regular/Hamiltonian.py:45-128 + ``H = -T + V``, 3Ddeuteron.py:80): accepted
    by ``Lanczos(H)`` in place of a SciPy matrix, so the matrix is assembled ON THE DEVICE(S) - each rank its own slab - and
    never exists on the host (BASELINE config C4: 8.4 GB of CSR).  ``to_scipy()`` materialises it (diagnostics, small sizes)."""

    def __init__(self, dims, points, T_factor=1.0, weights4=(-6.0, 1.0, 0.0, 0.0), negate_T=True, potential=None, potential_params=None):
        self.dims = tuple(int(d) for d in dims)
        self.points = int(points)
        self.T_factor = float(T_factor)
        self.weights4 = tuple(float(w) for w in weights4)
        self.negate_T = bool(negate_T)
        self.potential = None if potential is None else np.ascontiguousarray(potential, dtype=np.float64).reshape(-1)
        self.potential_params = None if potential_params is None else np.ascontiguousarray(potential_params, dtype=np.float64)
        M = int(np.prod(self.dims))
        self.shape = (M, M)
        if self.potential is not None and self.potential.shape != (M,):
            raise ValueError("potential must have one entry per grid point")

    def key(self):
        """What decides whether the device copy is current.  The potential enters by CONTENT (8 M bytes through the threaded
        hash of the matrix cache: cheap next to a run), not by address: ``np.ascontiguousarray`` aliases the caller's array, so
        a parameter scan that changes it in place - or a freed array whose address is reused - must not be mistaken for the
        matrix already on the device."""
        from ._solver import _fingerprint

        return ("stencil", self.dims, self.points, self.T_factor, self.weights4, self.negate_T,
                None if self.potential is None else _fingerprint(self.potential),
                None if self.potential_params is None else tuple(float(x) for x in self.potential_params))

    def to_scipy(self, device_id=0):
        import scipy.sparse

        h = _capi.Handle(device_id)
        try:
            h.build_stencil3d_block(self.dims, self.points, self.T_factor, self.weights4, 0, self.shape[0], (), potential=self.potential,
                                    potential_params=self.potential_params, negate_T=self.negate_T)
            rowptr, colidx, vals = h.get_csr()
        finally:
            h.close()
        return scipy.sparse.csr_matrix((vals, colidx, rowptr), shape=self.shape)


class PoolHandle:
    """What ``_solver.LanczosBase`` needs from a ``_capi.Handle``, served by a ``WorkerPool``."""

    # Deadlines (round 4; SURVEY section 5 "failure detection").  A dead rank was always noticed (poll()); a STALLED one - a
    # collective that never completes, a stopped process - was not: the caller hung.  Every command now carries a deadline
    # derived from its work with pessimistic rates (a tenth or less of what the hardware does), never below `floor_s`;
    # `timeout_override` (Lanczos.worker_timeout) replaces it.  On expiry the pool is terminated and the /dev/shm segments
    # of the command are unlinked by their context managers.
    floor_s = 60.0
    rate_host_copy = 0.5e9     # bytes/s through /dev/shm + a worker's validation + upload
    rate_device_stream = 2e11  # bytes/s of algorithmic traffic per rank (HBM does 6e12)
    rate_flops = 2e12          # FP64 flop/s per rank in the Ritz GEMMs (the kernels do 5e13)
    per_step_s = 0.05          # per Lanczos step: collectives, host-staged backend included

    def __init__(self, devices, backend="rccl"):
        self.timeout_override = None
        self.pool = WorkerPool(devices, backend)
        self.world = self.pool.world
        self.breakdown = False
        self.matrix_uploads = 0
        self._flags = 0
        self._info = None
        self.rows = self.n = None
        self._last = {}

    def close(self):
        self.pool.close()

    def _seg(self, name, shape, dtype):
        return _Shm(self.pool.key, name, shape, dtype)

    def deadline(self, host_bytes=0.0, device_bytes=0.0, flops=0.0, steps=0):
        """seconds a command of that much work may take before its ranks count as stalled"""
        if self.timeout_override is not None:
            return float(self.timeout_override)
        w = max(self.world, 1)
        return self.floor_s + host_bytes / self.rate_host_copy + device_bytes / w / self.rate_device_stream + flops / w / self.rate_flops + steps * self.per_step_s

    def _matrix_bytes(self):
        return float(getattr(self, "_mat_bytes", 0.0))

    def _run_deadline(self, n):
        M = float(self.rows or 0)
        return self.deadline(host_bytes=8.0 * M, device_bytes=n * (self._matrix_bytes() + 64.0 * M) + 8.0 * n * n * M, steps=n)

    # ---- matrix ---------------------------------------------------------------------------------------------------------------
    def set_options(self, flags):
        self._flags = int(flags)

    def _matrix(self, msg, host_bytes=0.0):
        msg = dict(msg, cmd="matrix", options=self._flags, fused_norm=bool(self._flags & _capi.FLAG_FUSED_NORM),
                   one_reduce=bool(self._flags & _capi.FLAG_ONE_REDUCE))
        self._info = self.pool.request(msg, timeout=self.deadline(host_bytes=3.0 * host_bytes, device_bytes=10.0 * host_bytes))
        self._mat_bytes = float(host_bytes) if host_bytes else 12.0 * 27 * float(msg["M"])
        self.matrix_uploads += 1
        self.rows = int(msg["M"])
        bounds = partition.row_bounds(self.rows, self.world)
        assert [i["lo"] for i in self._info] == bounds[:-1]

    def set_csr(self, M_global, row0, rowptr, colidx, vals, ncols_ext=None, mode="auto"):
        rowptr, colidx, vals = np.asarray(rowptr), np.asarray(colidx), np.asarray(vals, dtype=np.float64)
        with self._seg("rowptr", rowptr.shape, np.int64) as a, self._seg("colidx", colidx.shape, np.int32) as b, self._seg("vals", vals.shape, np.float64) as c:
            _copy_parallel([(a.arr, rowptr), (b.arr, colidx), (c.arr, vals)])  # (640 MB at the headline: a few threads, not one)
            self._matrix({"kind": "csr", "M": int(M_global), "rowptr": a.spec, "colidx": b.spec, "vals": c.spec, "mode": mode},
                         host_bytes=float(rowptr.nbytes + colidx.nbytes + vals.nbytes))

    def set_dense(self, A):
        A = np.asarray(A, dtype=np.float64)
        with self._seg("dense", A.shape, np.float64) as a:
            a.arr[...] = A
            self._matrix({"kind": "dense", "M": A.shape[0], "A": a.spec}, host_bytes=float(A.nbytes))

    def set_stencil(self, op):
        msg = {"kind": "stencil", "M": op.shape[0], "dims": list(op.dims), "points": op.points, "T_factor": op.T_factor, "weights4": list(op.weights4),
               "negate_T": op.negate_T, "potential": None, "potential_params": op.potential_params}
        if op.potential is not None:
            with self._seg("pot", op.potential.shape, np.float64) as p:
                p.arr[...] = op.potential
                self._matrix(dict(msg, potential=p.spec), host_bytes=float(op.potential.nbytes))
        else:
            self._matrix(msg)

    def spmv_plan(self):
        return self._info[0]["spmv"]

    def exchange_mode(self):
        return self._info[0]["exchange"]

    def device_name(self):
        return f"{self.world} x {self._info[0]['device']}" if self._info else f"{self.world} workers"

    # ---- run --------------------------------------------------------------------------------------------------------------------
    def run(self, n, v0):
        v0 = np.asarray(v0, dtype=np.float64)
        if v0.shape != (self.rows,):
            raise ValueError("v0 has the wrong length")
        with self._seg("v0", v0.shape, np.float64) as s:
            s.arr[...] = v0
            rep = self.pool.request({"cmd": "run", "n": int(n), "v0": s.spec, "options": self._flags}, timeout=self._run_deadline(int(n)))
        return self._coefficients(rep, n)

    def _coefficients(self, rep, n):
        for r in rep[1:]:  # alpha, beta are all-reduced sums: every rank must hold the same bits
            if not (np.array_equal(r["alpha"], rep[0]["alpha"], equal_nan=True) and np.array_equal(r["beta"], rep[0]["beta"], equal_nan=True)):
                raise _capi.LanczosHipError(-3, "the ranks disagree on the recurrence coefficients")
        self._last = rep[0]
        self._all_timings = [r["timings"] for r in rep]
        self.breakdown = any(r["breakdown"] for r in rep)
        self.n = int(n)
        return rep[0]["alpha"], rep[0]["beta"]

    def get_residual(self):
        """(M,): r entering step n of the last run, every rank writing its rows"""
        with self._seg("r", (self.rows,), np.float64) as s:
            self.pool.request({"cmd": "residual", "r": s.spec}, timeout=self.deadline(host_bytes=8.0 * self.rows))
            return np.array(s.arr)

    def run_resume(self, n, V_rows, r, alpha, beta):
        """continue a run of j0 = len(V_rows) completed steps to n in total: every rank takes its rows of the checkpoint out of
        the shared mapping (the partition need not be the one the checkpoint was written with)"""
        V_rows, r = np.asarray(V_rows, dtype=np.float64), np.asarray(r, dtype=np.float64)
        alpha, beta = np.ascontiguousarray(alpha, dtype=np.float64), np.ascontiguousarray(beta, dtype=np.float64)
        j0 = V_rows.shape[0]
        if V_rows.shape != (j0, self.rows) or r.shape != (self.rows,) or alpha.shape != (j0,) or beta.shape != (max(j0 - 1, 0),):
            raise ValueError("checkpoint arrays have the wrong shapes")
        with self._seg("V", V_rows.shape, np.float64) as sv, self._seg("r", r.shape, np.float64) as sr:
            sv.arr[...] = V_rows
            sr.arr[...] = r
            rep = self.pool.request({"cmd": "resume", "n": int(n), "V": sv.spec, "r": sr.spec, "alpha": alpha, "beta": beta, "options": self._flags},
                                    timeout=self._run_deadline(int(n)) + float(V_rows.nbytes) / self.rate_host_copy)
        return self._coefficients(rep, n)

    def timings(self):
        return self._last.get("timings")

    def last_sweeps(self):
        return self._last.get("sweeps")

    def last_engine(self):
        return self._last.get("engine")

    def get_basis(self):
        """(n, M): every rank writes its columns of every basis row straight into the shared mapping"""
        with self._seg("V", (self.n, self.rows), np.float64) as s:
            self.pool.request({"cmd": "fetch_basis", "V": s.spec}, timeout=self.deadline(host_bytes=8.0 * self.n * self.rows))
            return s.arr  # the mapping itself (no second copy of a 16 GB basis): it outlives the unlinked file until the array is dropped

    def get_basis_block(self, r0, r1):
        r0, r1 = int(r0), int(r1)
        with self._seg("Vb", (self.n, r1 - r0), np.float64) as s:
            self.pool.request({"cmd": "fetch_basis_block", "V": s.spec, "rows": (r0, r1)}, timeout=self.deadline(host_bytes=8.0 * self.n * (r1 - r0)))
            return np.array(s.arr)

    def ritz_vectors(self, S, fetch=True):
        S = np.ascontiguousarray(S, dtype=np.float64)
        with self._seg("S", S.shape, np.float64) as s:
            s.arr[...] = S
            self.pool.request({"cmd": "ritz", "S": s.spec}, timeout=self.deadline(flops=2.0 * self.rows * self.n * self.n, device_bytes=16.0 * self.rows * self.n))
        return self.ritz_fetch() if fetch else None

    def ritz_fetch(self):
        return self.ritz_fetch_rows(0, self.rows)

    def ritz_fetch_rows(self, r0, r1):
        r0, r1 = int(r0), int(r1)
        with self._seg("Y", (r1 - r0, self.n), np.float64) as s:
            self.pool.request({"cmd": "fetch_ritz", "Y": s.spec, "rows": (r0, r1)},
                              timeout=self.deadline(host_bytes=8.0 * self.n * (r1 - r0), flops=2.0 * (r1 - r0) * self.n * self.n))
            return s.arr  # (see get_basis)

    def ritz_gram(self):
        return self.pool.request({"cmd": "gram"}, timeout=self.deadline(flops=4.0 * self.rows * self.n * self.n, device_bytes=16.0 * self.rows * self.n))[0]["G"]

    def ritz_quality(self):
        return self.pool.request({"cmd": "quality"}, timeout=self.deadline(flops=2.0 * self.rows * self.n * self.n,
                                                                           device_bytes=self.n * (self._matrix_bytes() + 32.0 * self.rows), steps=self.n))[0]["q"]
