"""Multi-GPU driver: one process per GPU, rows block-partitioned, RCCL collectives.

Data path: liblanczos_hip.so calls RCCL directly (all-reduce of alpha, ||r||^2
and the re-orthogonalisation coefficients; neighbour send/recv or all-gather of
the SpMV input) on the compute stream - nothing passes through Python during
the run.  Python only bootstraps: it plans the partition (partition.py) and
hands the 128-byte RCCL unique id from rank 0 to the other ranks through a
``Bootstrap`` (``TorchBootstrap`` = torch.distributed/gloo, the rendezvous the
launcher already provides; PyTorch is plumbing here, never on the data path).

``backend="host"`` stages the same collectives through host memory and the
bootstrap; it exists so the partitioned path can be tested with several ranks
on a single GPU (RCCL refuses two ranks on one device).
"""
from __future__ import annotations

import os

import numpy as np

from . import _capi, partition


class Bootstrap:
    """Minimal control-plane interface (setup only)."""

    rank = 0
    world = 1

    def allgather_obj(self, obj):
        return [obj]

    def broadcast_bytes(self, data, root=0):
        return data

    def barrier(self):
        pass

    # host-staged data path (backend="host")
    def allreduce_sum(self, a):
        return a

    def exchange(self, peers, send_segments, recv_counts):
        return []

    def allgather_array(self, a):
        return a


def _enc(o):
    """JSON-able form of the control-plane messages (None, bool, int, float, str, bytes, tuples, lists, dicts with any
    such keys, NumPy scalars and arrays).  No pickle: a message can never execute code on the receiving rank."""
    import base64

    if isinstance(o, np.ndarray):
        return {"__nd__": [o.dtype.str, list(o.shape)], "d": base64.b64encode(np.ascontiguousarray(o).tobytes()).decode("ascii")}
    if isinstance(o, (bytes, bytearray)):
        return {"__b__": base64.b64encode(bytes(o)).decode("ascii")}
    if isinstance(o, tuple):
        return {"__t__": [_enc(x) for x in o]}
    if isinstance(o, list):
        return [_enc(x) for x in o]
    if isinstance(o, dict):
        return {"__d__": [[_enc(k), _enc(v)] for k, v in o.items()]}
    if isinstance(o, np.bool_):
        return bool(o)
    if isinstance(o, np.integer):
        return int(o)
    if isinstance(o, np.floating):
        return float(o)
    if o is None or isinstance(o, (bool, int, float, str)):
        return o
    raise TypeError(f"SocketBootstrap cannot send a {type(o).__name__}")


def _dec(o):
    import base64

    if isinstance(o, list):
        return [_dec(x) for x in o]
    if isinstance(o, dict):
        if "__nd__" in o:
            dt, shape = o["__nd__"]
            return np.frombuffer(base64.b64decode(o["d"]), dtype=np.dtype(dt)).reshape(shape).copy()
        if "__b__" in o:
            return base64.b64decode(o["__b__"])
        if "__t__" in o:
            return tuple(_dec(x) for x in o["__t__"])
        if "__d__" in o:
            return {_dec(k): _dec(v) for k, v in o["__d__"]}
        raise ValueError("malformed rendezvous message")
    return o


class SocketBootstrap(Bootstrap):
    """Pure-Python single-node rendezvous over a Unix-domain socket (star topology, rank 0 is the hub).

    Needs only RANK / WORLD_SIZE (as exported by ``torch.distributed.run`` or any other launcher) and a
    key shared by the ranks of one launch - ``LZ_RDZV_KEY``, else ``MASTER_PORT`` plus the launcher's pid - so no
    TCP port is taken and no PyTorch (with its bundled copies of the ROCm runtime and RCCL) is mapped
    into the process.  Control plane only: unique-id broadcast, plan checks, barriers, the max-over-ranks of timings
    (and, for ``backend="host"``, the host-staged test collectives).

    The socket lives in a directory only this user can enter (``LZ_RDZV_DIR`` or ``/tmp/lz_rdzv_<uid>``, mode 0700,
    ownership verified), both ends check the peer's uid with SO_PEERCRED, and messages are JSON (``_enc``), never pickle.
    """

    def __init__(self, rank=None, world=None, key=None, timeout=600.0):
        import atexit
        import socket
        import time

        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        if key is None:
            key = os.environ.get("LZ_RDZV_KEY") or f"{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"
        self.path = os.path.join(self._private_dir(), f"{key}.sock")
        self._conns = {}
        self._sock = None
        if self.world == 1:
            return
        if self.rank == 0:
            srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            try:
                os.unlink(self.path)
            except FileNotFoundError:
                pass
            srv.bind(self.path)
            srv.listen(self.world)
            srv.settimeout(timeout)
            atexit.register(self._cleanup)
            self._srv = srv
            while len(self._conns) < self.world - 1:
                c, _ = srv.accept()
                self._check_peer(c)
                c.settimeout(timeout)
                r = self._recv(c)
                self._conns[int(r)] = c
        else:
            deadline = time.time() + timeout
            while True:
                s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                try:
                    s.connect(self.path)
                    break
                except (FileNotFoundError, ConnectionRefusedError):
                    s.close()
                    if time.time() > deadline:
                        raise TimeoutError(f"rank {self.rank}: no rendezvous socket at {self.path}")
                    time.sleep(0.05)
            self._check_peer(s)
            s.settimeout(timeout)
            self._sock = s
            self._send(s, self.rank)

    @staticmethod
    def _private_dir():
        import stat

        d = os.environ.get("LZ_RDZV_DIR") or os.path.join("/tmp", f"lz_rdzv_{os.getuid()}")
        os.makedirs(d, mode=0o700, exist_ok=True)
        st = os.lstat(d)
        if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
            raise PermissionError(f"rendezvous directory {d} must be a directory owned by uid {os.getuid()} with mode 0700")
        return d

    @staticmethod
    def _check_peer(sock):
        import socket
        import struct

        cred = sock.getsockopt(socket.SOL_SOCKET, socket.SO_PEERCRED, struct.calcsize("3i"))
        _pid, uid, _gid = struct.unpack("3i", cred)
        if uid != os.getuid():
            sock.close()
            raise PermissionError(f"rendezvous peer runs as uid {uid}, expected {os.getuid()}")

    def _cleanup(self):
        try:
            os.unlink(self.path)
        except OSError:
            pass

    @staticmethod
    def _send(sock, obj):
        import json
        import struct

        data = json.dumps(_enc(obj)).encode("utf-8")
        sock.sendall(struct.pack("<Q", len(data)) + data)

    @staticmethod
    def _recv(sock):
        import json
        import struct

        def read(n):
            buf = bytearray()
            while len(buf) < n:
                chunk = sock.recv(min(n - len(buf), 1 << 20))
                if not chunk:
                    raise ConnectionError("rendezvous peer closed the connection")
                buf += chunk
            return bytes(buf)

        (n,) = struct.unpack("<Q", read(8))
        return _dec(json.loads(read(n).decode("utf-8")))

    def _hub(self, obj, combine):
        """rank 0 collects one object per rank, ``combine(list) -> per-rank replies``; everyone gets its reply."""
        if self.world == 1:
            return combine([obj])[0]
        if self.rank == 0:
            items = [obj] + [None] * (self.world - 1)
            for r in range(1, self.world):
                items[r] = self._recv(self._conns[r])
            replies = combine(items)
            for r in range(1, self.world):
                self._send(self._conns[r], replies[r])
            return replies[0]
        self._send(self._sock, obj)
        return self._recv(self._sock)

    def allgather_obj(self, obj):
        return self._hub(obj, lambda items: [items] * len(items))

    def broadcast_bytes(self, data, root=0):
        return self.allgather_obj(data)[root]

    def barrier(self):
        self.allgather_obj(None)

    def allreduce_sum(self, a):
        def combine(items):
            total = items[0].copy()
            for x in items[1:]:
                total = total + x
            return [total] * len(items)

        a[...] = self._hub(np.array(a, dtype=np.float64), combine)
        return a

    def exchange(self, peers, send_segments, recv_counts):
        def combine(items):  # items[r] = {dst: segment}
            return [{src: items[src][dst] for src in range(len(items)) if dst in items[src]} for dst in range(len(items))]

        got = self._hub({int(p): np.ascontiguousarray(s) for p, s in zip(peers, send_segments)}, combine)
        return [got.get(int(p), np.zeros(0)) for p in peers]

    def allgather_array(self, a):
        return np.concatenate(self.allgather_obj(np.ascontiguousarray(a)))


class TorchBootstrap(Bootstrap):
    """torch.distributed (gloo, CPU) rendezvous from the RANK/WORLD_SIZE/MASTER_* environment.

    Construct it BEFORE ``lanczos_amd.load_library()`` / the first ``Handle``: the HIP library then binds to the one
    runtime torch has mapped (and takes RCCL from the same tree).  The opposite order is refused."""

    def __init__(self, init=True):
        import torch
        import torch.distributed as dist

        # torch brings its own ROCm runtime (torch/lib): if liblanczos_hip.so was loaded first there are now TWO HIP
        # runtimes in the process - the round-1 teardown abort.  Fail here, loudly.
        _capi.check_single_runtime()
        self._torch, self._dist = torch, dist
        if init and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def allgather_obj(self, obj):
        out = [None] * self.world
        self._dist.all_gather_object(out, obj)
        return out

    def broadcast_bytes(self, data, root=0):
        box = [data if self.rank == root else None]
        self._dist.broadcast_object_list(box, src=root)
        return box[0]

    def barrier(self):
        self._dist.barrier()

    def allreduce_sum(self, a):
        t = self._torch.from_numpy(a)
        self._dist.all_reduce(t)
        return a

    def exchange(self, peers, send_segments, recv_counts):
        torch, dist = self._torch, self._dist
        recv = [torch.empty(int(c), dtype=torch.float64) for c in recv_counts]
        reqs = []
        for p, seg, r in zip(peers, send_segments, recv):
            if len(seg):
                reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(seg)), dst=int(p)))
            if r.numel():
                reqs.append(dist.irecv(r, src=int(p)))
        for q in reqs:
            q.wait()
        return [r.numpy() for r in recv]

    def allgather_array(self, a):
        torch, dist = self._torch, self._dist
        out = [torch.empty(len(a), dtype=torch.float64) for _ in range(self.world)]
        dist.all_gather(out, torch.from_numpy(np.ascontiguousarray(a)))
        return np.concatenate([o.numpy() for o in out])


class DistributedLanczos:
    """Row-partitioned Lanczos over ``boot.world`` ranks (this object = one rank).

    ``local`` is this rank's row block as a ``synthetic.CSR``-like object (``rowptr``,
    ``colidx`` with GLOBAL column ids, ``vals``) for the rows ``partition.row_bounds(M, world)``
    assigns to ``boot.rank`` - or, for a dense ``H``, the ``(rows_local, M)`` ndarray block of those rows (the SpMV input is
    then all-gathered).  Results: ``alpha``, ``beta``, ``H_eff`` replicated; ``V_local``
    / ``H_eigvecs_local`` hold this rank's rows.
    """

    def __init__(self, local, M, boot=None, device_id=0, backend="rccl", mode="auto", options=0, fused_norm=True, one_reduce=False,
                 tuning=None):
        self._tuning = dict(tuning or {})  # lz_set_tuning knobs applied BEFORE the matrix is uploaded (some are read there)
        if one_reduce:  # opt-in: alpha, |r|^2 and the re-orthogonalisation coefficients travel in ONE all-reduce per iteration
            options |= _capi.FLAG_ONE_REDUCE
            fused_norm = True
        self.boot = boot or Bootstrap()
        self.M = int(M)
        self.rank, self.world = self.boot.rank, self.boot.world
        bounds = partition.row_bounds(self.M, self.world)
        self.lo, self.hi = bounds[self.rank], bounds[self.rank + 1]
        rows = self.hi - self.lo
        self.dense = isinstance(local, np.ndarray)
        if self.dense:
            if local.shape != (rows, self.M):
                raise ValueError(f"rank {self.rank} must own the dense block of rows [{self.lo}, {self.hi}) x {self.M} columns")
            self._init_dense(local, device_id, backend, options, fused_norm)
            return
        if len(local.rowptr) != rows + 1:
            raise ValueError(f"rank {self.rank} must own rows [{self.lo}, {self.hi})")
        modes = self.boot.allgather_obj(partition.plan_exchange(local.rowptr, local.colidx, self.M, self.world, self.rank, mode).mode)
        if "allgather" in modes:
            mode = "allgather"  # all ranks must agree
        self.plan = partition.plan_exchange(local.rowptr, local.colidx, self.M, self.world, self.rank, mode)
        if self.plan.mode == "halo":
            gathered = self.boot.allgather_obj((self.plan.peers, self.plan.send_counts, self.plan.recv_counts))
            partition.check_plans(self.plan, self.rank, gathered)
        if fused_norm:
            # ||r||^2 rides with the Q^T r partials: one all-reduce per re-orthogonalisation carries [Q^T r, r.r] (saves a
            # latency-bound collective per iteration at N > 1, and one pass over V[j] + one launch at any N)
            options |= _capi.FLAG_FUSED_NORM
        self.options = options
        self.h = _capi.Handle(device_id)
        self.h.set_options(options)
        for k, v in self._tuning.items():
            self.h.set_tuning(k, v)
        self.backend = backend
        self._init_comm(backend)
        self.h.set_csr(self.M, self.lo, local.rowptr, self.plan.colidx, local.vals, ncols_ext=self.plan.ncols_ext)
        if self.plan.mode == "halo":
            self.h.set_halo(self.plan.peers, self.plan.send_counts, self.plan.send_idx, self.plan.recv_counts)
        elif self.plan.mode == "allgather":
            self.h.set_allgather(self.plan.chunk)
        self.executed = False

    @classmethod
    def from_stencil(cls, dims, points, boot=None, device_id=0, backend="rccl", options=0, fused_norm=True, one_reduce=False,
                     tuning=None, T_factor=1.0, weights4=(-6.0, 1.0, 0.0, 0.0), negate_T=True, potential_params=None, potential_local=None):
        """This rank's slab of the periodic ``Nx x Ny x Nz`` 7-/27-point stencil operator assembled DIRECTLY ON THE DEVICE
        (no host matrix at any size: BASELINE config C4's 1e8-row Laplacian is 8.4 GB of CSR): the halo plan comes from
        the slab's boundary rows alone (``partition.plan_stencil_slab``), the kernel renumbers the columns itself.
        Defaults give ``synthetic.laplacian_3d_7pt`` (6 on the diagonal, -1 on the six neighbours), bit for bit."""
        self = cls.__new__(cls)
        self.boot = boot or Bootstrap()
        Nx, Ny, Nz = (int(d) for d in dims)
        self.M = Nx * Ny * Nz
        self.rank, self.world = self.boot.rank, self.boot.world
        bounds = partition.row_bounds(self.M, self.world)
        self.lo, self.hi = bounds[self.rank], bounds[self.rank + 1]
        self.dense = False
        self._tuning = dict(tuning or {})
        self.plan, ranges = partition.plan_stencil_slab(dims, points, self.world, self.rank)
        if self.plan.mode == "halo":
            gathered = self.boot.allgather_obj((self.plan.peers, self.plan.send_counts, self.plan.recv_counts))
            partition.check_plans(self.plan, self.rank, gathered)
        if one_reduce:
            options |= _capi.FLAG_ONE_REDUCE
            fused_norm = True
        if fused_norm:
            options |= _capi.FLAG_FUSED_NORM
        self.options = options
        self.h = _capi.Handle(device_id)
        self.h.set_options(options)
        for k, v in self._tuning.items():
            self.h.set_tuning(k, v)
        self.backend = backend
        self._init_comm(backend)
        self.h.build_stencil3d_block(dims, points, T_factor, weights4, self.lo, self.hi - self.lo, ranges, potential=potential_local,
                                     potential_params=potential_params, negate_T=negate_T)
        if self.plan.mode == "halo":
            self.h.set_halo(self.plan.peers, self.plan.send_counts, self.plan.send_idx, self.plan.recv_counts)
        self.executed = False
        return self

    def _init_dense(self, block, device_id, backend, options, fused_norm):
        """Dense row block: columns keep their global index (every rank starts at rank * chunk, so the all-gathered
        padded vector is the global numbering followed by zero padding up to world * chunk)."""
        from types import SimpleNamespace

        if fused_norm:
            options |= _capi.FLAG_FUSED_NORM
        self.options = options
        self.h = _capi.Handle(device_id)
        self.h.set_options(options)
        for k, v in self._tuning.items():
            self.h.set_tuning(k, v)
        self.backend = backend
        self._init_comm(backend)
        chunk = partition.chunk_size(self.M, self.world)
        if self.world == 1:
            self.h.set_dense(block)
            self.plan = SimpleNamespace(mode="none", chunk=chunk, ncols_ext=self.M)
        else:
            ext = np.zeros((block.shape[0], chunk * self.world))
            ext[:, : self.M] = block
            self.h.set_dense_block(self.M, self.lo, ext)
            self.h.set_allgather(chunk)
            self.plan = SimpleNamespace(mode="allgather", chunk=chunk, ncols_ext=chunk * self.world)
        self.executed = False

    def _init_comm(self, backend):
        if self.world > 1:
            if backend == "rccl":
                uid = None
                if self.rank == 0:
                    try:
                        uid = self.h.unique_id()
                    except _capi.LanczosHipError as e:  # keep the ranks in step: everyone learns about the failure
                        uid = ("error", str(e))
                uid = self.boot.broadcast_bytes(uid, root=0)
                if isinstance(uid, tuple):
                    raise _capi.LanczosHipError(-3, f"rank 0 could not create the RCCL unique id: {uid[1]}")
                self.h.comm_init_rccl(self.world, self.rank, uid)
            elif backend == "host":
                self.h.comm_init_host(self.world, self.rank, self.boot.allreduce_sum, self.boot.exchange, self.boot.allgather_array)
            else:
                raise ValueError(f"unknown backend {backend!r}")

    def start_vector(self, seed=99, v0=None):
        """The reference's start vector (Lanczos.py:93-100), generated identically on every rank."""
        np.random.seed(seed)
        v = np.random.uniform(-1, 1, size=(self.M)) if v0 is None else np.array(v0)
        return v / np.linalg.norm(v)

    def execute_Lanczos(self, n, seed=99, v0=None, v0_normalized_local=None):
        if n > self.M:
            raise ValueError("n cannot be larger than M!")
        if v0_normalized_local is None:
            v0_normalized_local = self.start_vector(seed, v0)[self.lo : self.hi]
        self.n = n
        self.alpha, self.beta = self.h.run(n, v0_normalized_local)
        self.breakdown = bool(self.h.breakdown)
        if self.breakdown:  # lz_run returned LZ_WARN_BREAKDOWN: the same signal the single-GPU class gives (_solver.py)
            import warnings

            warnings.warn("Lanczos breakdown: a residual norm beta reached zero (invariant subspace); H_eff contains rounding "
                          "noise or non-finite entries, exactly as the reference's would", RuntimeWarning, stacklevel=2)
        idx = np.arange(n)
        H_eff = np.zeros((n, n))
        H_eff[idx, idx] = self.alpha
        H_eff[idx[:-1], idx[1:]] = self.beta
        H_eff[idx[1:], idx[:-1]] = self.beta
        self.H_eff = H_eff
        self.executed = True
        return self.alpha, self.beta

    def checkpoint_local(self):
        """this rank's share of the state a finished run leaves behind (see Lanczos.checkpoint): ``V`` (n, rows_local), ``r``
        (rows_local) + the coefficients every rank holds"""
        return {"alpha": self.alpha.copy(), "beta": self.beta.copy(), "V": self.h.get_basis(), "r": self.h.get_residual(), "M": self.M}

    def resume_Lanczos(self, n, V_rows_local, r_local, alpha, beta):
        """continue a run of ``len(alpha)`` completed steps to ``n`` in total from this rank's rows of the checkpoint; collective
        (every rank calls it with its own rows and the same coefficients); bit-identical to one run of ``n`` steps"""
        if n > self.M:
            raise ValueError("n cannot be larger than M!")
        self.n = n
        self.alpha, self.beta = self.h.run_resume(n, V_rows_local, r_local, alpha, beta)
        self.breakdown = bool(self.h.breakdown)
        idx = np.arange(n)
        H_eff = np.zeros((n, n))
        H_eff[idx, idx] = self.alpha
        H_eff[idx[:-1], idx[1:]] = self.beta
        H_eff[idx[1:], idx[:-1]] = self.beta
        self.H_eff = H_eff
        self.executed = True
        return self.alpha, self.beta

    @property
    def V_local(self):
        """(rows_local, n) block of the basis."""
        return self.h.get_basis().T

    def get_H_eigs(self, fetch=True):
        self.H_eigvals, S = np.linalg.eigh(self.H_eff)
        self.H_eigvecs_local = self.h.ritz_vectors(S, fetch=fetch)
        return self.H_eigvals

    def ritz_quality(self):
        """print_good_eigs' figure of merit (A y_i . y_i)^2 / ||A y_i||^2 for every Ritz vector of the last get_H_eigs
        (Lanczos.py:166-185), on the partitioned matrix: collective, every rank gets the same n values."""
        return self.h.ritz_quality()

    def timings(self):
        return self.h.timings()
