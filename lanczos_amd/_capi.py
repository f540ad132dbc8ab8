"""ctypes binding of liblanczos_hip.so (include/lanczos_hip.h).

The library is the only compute backend of this package: if it cannot be
loaded, or no GPU is present, every compute call raises - there is no CPU
fallback (the NumPy path is the reference's own and lives, as a checker, under
oracle/ for the tests only).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblanczos_hip.so")

LZ_OK = 0
LZ_WARN_BREAKDOWN = 1
STATUS_NAMES = {0: "LZ_OK", 1: "LZ_WARN_BREAKDOWN", -1: "LZ_ERR_ARG", -2: "LZ_ERR_HIP", -3: "LZ_ERR_COMM", -4: "LZ_ERR_STATE", -5: "LZ_ERR_NOMEM", -6: "LZ_ERR_NODEVICE"}

FLAG_PROFILE = 1
FLAG_QTW_MFMA = 2
FLAG_QTW_VALU = 4
FLAG_SPMV_SCALAR = 8
FLAG_FUSED_NORM = 16
FLAG_SPMV_STREAM = 32
FLAG_REORTH_PARTIAL = 64
FLAG_OVERLAP_HALO = 128
FLAG_ONE_REDUCE = 256

# lz_set_tuning knob indices (the legend lives in include/lanczos_hip.h)
TUNE_QTW_SLICE = 0          # Q^T w slice length per block
TUNE_QTW_VARIANT = 1        # Q^T w kernel variant (unroll / rows per tile; >= 20: timing-only ablations, kernel-bench build)
TUNE_STREAM_ROWS = 2        # CSR-stream rows per block
TUNE_SPMV_ABLATION = 3      # (removed in round 5: the timing-only SpMV ablation arms; any non-zero value is refused)
TUNE_STREAM_ENTRIES = 4     # CSR-stream entries per block
TUNE_FIXED_ROWS = 5         # fixed-K rows per block (CSR-order kernel)
TUNE_FORCE_COLLECTIVES = 6  # issue the collectives even when world == 1
TUNE_PROFILE_STRIDE = 7     # bracket only every value-th iteration of lz_run with events
TUNE_UPDATE_VARIANT = 8     # pass-2 kernel variant
TUNE_RITZ_KERNEL = 9        # Ritz back-transform kernel (0 auto, 1 the one-workgroup-per-128-rows kernel always)
TUNE_PB_ENTRIES = 10        # two-phase SpMV: entries per row block
TUNE_BI_LINKS = 11          # two-sided Gram-Schmidt links
TUNE_NO_SKEW = 12           # 1 = no row-stride skew
TUNE_POISON_BASIS = 13      # 1 = NaN-poison a fresh basis allocation (test knob)
TUNE_SPMV_PLAN = 14         # irregular SpMV plan (0 auto, 1 never two-phase, 2 always)
TUNE_LOOP = 15              # loop structure (0 auto, 1 six launches per step always)
TUNE_RITZ_CHUNK_ROWS = 16   # rows per chunk of the chunked Ritz mode (> 0 forces it)
TUNE_FIXED_LAYOUT = 17      # fixed-K SpMV layout (0 auto: row-class coded ELL where the rows fall into classes, else CSR order and ELL only in the partial loop;
                            # 1 never ELL; 2 / 3 uncoded ELL always, one / two rows per lane; 4 offsets-only coding)
TUNE_GRAM_KERNEL = 19       # Gram matrix of the Ritz vectors (0 auto: symmetric accumulator-stationary kernel with LDS-staged operands, 1 split-K TN GEMM, 2 the register-ring form)
TUNE_CLS_GROUP = 23         # fully coded SpMV: 0 two adjacent rows per lane; 1 / 3 one row per lane, one / two units per workgroup (A/B)
TUNE_PB_GROUPS = 22         # two-phase SpMV: phases interleaved over this many row-block groups (A/B arm; 0 = off)
TUNE_GRAM_SLICES = 21       # Gram matrix: K slices of the symmetric kernel (0 auto)
TUNE_PARTIAL_LOOKAHEAD = 20 # one-reduce partial loop: safety factor of the look-ahead sweep decision (0 = default 4)
TUNE_PARTIAL_LOOP = 18      # partial re-orthogonalisation loop (0 device-resident, 1 host-decided, 2 device-resident without the fused scale,
                            # 3 device-resident with a separate second-stage kernel behind pass 1)

KERNEL_CLASSES = ("spmv", "qtw", "update", "three_term", "final", "comm", "ritz")
K_COUNT = len(KERNEL_CLASSES)


class LzTimings(C.Structure):
    _fields_ = [
        ("ms", C.c_double * K_COUNT),
        ("timed_bytes", C.c_double * K_COUNT),
        ("timed_launches", C.c_int64 * K_COUNT),
        ("bytes", C.c_double * K_COUNT),
        ("flops", C.c_double * K_COUNT),
        ("launches", C.c_int64 * K_COUNT),
        ("total_ms", C.c_double),
    ]


HOST_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int64)
HOST_EXCHANGE_FN = C.CFUNCTYPE(
    C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int64)
)
HOST_ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64)

_P = C.c_void_p
_D = C.POINTER(C.c_double)
_I32 = C.POINTER(C.c_int32)
_I64 = C.POINTER(C.c_int64)

# name -> (restype, argtypes): every symbol include/lanczos_hip.h declares
SIGNATURES = {
    "lz_version": (C.c_int, []),
    "lz_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "lz_create": (C.c_int, [C.POINTER(_P), C.c_int]),
    "lz_destroy": (C.c_int, [_P]),
    "lz_last_error": (C.c_char_p, [_P]),
    "lz_set_options": (C.c_int, [_P, C.c_int]),
    "lz_set_tuning": (C.c_int, [_P, C.c_int, C.c_int]),
    "lz_runtime_info": (C.c_int, [C.c_char_p, C.c_size_t]),
    "lz_device_synchronize": (C.c_int, [_P]),
    "lz_device_memory": (C.c_int, [_P, _I64, _I64]),
    "lz_device_name": (C.c_int, [_P, C.c_char_p, C.c_size_t]),
    "lz_padded_rows": (C.c_int64, [C.c_int64]),
    "lz_comm_load": (C.c_int, []),
    "lz_comm_unique_id": (C.c_int, [_P, C.c_size_t]),
    "lz_comm_init_rccl": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_size_t]),
    "lz_comm_init_host": (C.c_int, [_P, C.c_int, C.c_int, HOST_ALLREDUCE_FN, HOST_EXCHANGE_FN, HOST_ALLGATHER_FN, _P]),
    "lz_set_csr": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _I32, _I32, _D]),
    "lz_set_dense": (C.c_int, [_P, C.c_int64, _D]),
    "lz_set_dense_block": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _D]),
    "lz_build_stencil3d": (C.c_int, [_P, C.c_int, C.c_int, C.c_double, _D, _D, C.c_int]),
    "lz_build_stencil3d_block": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, _D, C.c_int, _D, C.c_int, C.c_int64, C.c_int64,
                                           C.c_int, _I64, _I64]),
    "lz_csr_info": (C.c_int, [_P, _I64, _I64]),
    "lz_spmv_plan": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "lz_spmv_coding": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "lz_get_csr": (C.c_int, [_P, _I32, _I32, _D]),
    "lz_set_halo": (C.c_int, [_P, C.c_int, _I32, _I64, _I32, _I64]),
    "lz_set_allgather": (C.c_int, [_P, C.c_int64]),
    "lz_reserve": (C.c_int, [_P, C.c_int64, C.c_int, C.c_int]),
    "lz_run": (C.c_int, [_P, C.c_int, _D, _D, _D]),
    "lz_get_residual": (C.c_int, [_P, _D]),
    "lz_run_resume": (C.c_int, [_P, C.c_int, C.c_int, _D, C.c_int64, _D, _D, _D, _D, _D]),
    "lz_get_basis": (C.c_int, [_P, _D, C.c_int64]),
    "lz_get_basis_block": (C.c_int, [_P, C.c_int64, C.c_int64, _D, C.c_int64]),
    "lz_ritz_vectors": (C.c_int, [_P, _D, _D]),
    "lz_get_ritz_vectors": (C.c_int, [_P, _D]),
    "lz_get_ritz_rows": (C.c_int, [_P, C.c_int64, C.c_int64, _D]),
    "lz_ritz_info": (C.c_int, [_P, _I64, _D]),
    "lz_ritz_gram": (C.c_int, [_P, _D]),
    "lz_gram_info": (C.c_int, [_P, _D]),
    "lz_ritz_quality": (C.c_int, [_P, _D]),
    "lz_get_timings": (C.c_int, [_P, C.POINTER(LzTimings)]),
    "lz_last_sweeps": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "lz_last_sweep_misses": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "lz_get_omega_state": (C.c_int, [_P, _D, C.c_int64]),
    "lz_run_resume_partial": (C.c_int, [_P, C.c_int, C.c_int, _D, C.c_int64, _D, _D, _D, _D, _D, _D]),
    "lz_last_sweep_log": (C.c_int, [_P, C.POINTER(C.c_int), C.c_int]),
    "lz_comm_counts": (C.c_int, [_P, _I64, _I64]),
    "lz_last_engine": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "lz_last_host_syncs": (C.c_int, [_P, _I64]),
    "lz_basis_alloc": (C.c_int, [_P, C.c_int]),
    "lz_basis_set_row": (C.c_int, [_P, C.c_int, _D]),
    "lz_basis_set_rows": (C.c_int, [_P, C.c_int, C.c_int, _D, C.c_int64]),
    "lz_basis_get_row": (C.c_int, [_P, C.c_int, _D]),
    "lz_r_set": (C.c_int, [_P, _D]),
    "lz_r_get": (C.c_int, [_P, _D]),
    "lz_step_spmv": (C.c_int, [_P, C.c_int, _D]),
    "lz_step_reorth": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _D, _D]),
    "lz_step_three_term": (C.c_int, [_P, C.c_int, C.c_int, C.c_double, C.c_double, _D]),
    "lz_spmv_host": (C.c_int, [_P, _D, _D]),
    "lz_set_csr_transpose": (C.c_int, [_P, C.c_int64, _I32, _I32, _D]),
    "lz_run_two_sided": (C.c_int, [_P, C.c_int, _D, _D, _D, _D, _D]),
    "lz_bi_alloc": (C.c_int, [_P, C.c_int]),
    "lz_bi_set_row": (C.c_int, [_P, C.c_int, C.c_int, _D]),
    "lz_bi_get_row": (C.c_int, [_P, C.c_int, C.c_int, _D]),
    "lz_step_bireorth": (C.c_int, [_P, C.c_int]),
    "lz_step_bireorth_mem_safe": (C.c_int, [_P, C.c_int]),
}


class LanczosHipError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


KBENCH_LIB_PATH = os.path.join(_HERE, "liblanczos_kbench.so")  # kernel-bench build: retired A/B arms + timing-only ablations (tools/, tests of those arms)

_lib = None
_live_handles = None  # weakref.WeakSet of open Handle objects, closed by an atexit hook (see Handle)


def mapped_runtimes():
    """{"amdhip64": [paths], "hsa-runtime64": [...], "rccl": [...]} of the ROCm runtime copies mapped into THIS process
    (from /proc/self/maps).  More than one libamdhip64 means two HIP runtimes share the process - what happens when
    PyTorch (which bundles its own ROCm libraries under torch/lib, found through unversioned NEEDED names that never
    match the system sonames) is imported AFTER this library was loaded; DESIGN.md section 5 has the round-1 abort it caused."""
    import re

    found = {"amdhip64": set(), "hsa-runtime64": set(), "rccl": set()}
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                m = re.search(r"(/\S*lib(amdhip64|hsa-runtime64|rccl)\.so\S*)", line)
                if m:
                    found[m.group(2)].add(m.group(1))
    except OSError:
        pass
    return {k: sorted(v) for k, v in found.items()}


def check_single_runtime():
    """Raise if two HIP runtimes are mapped (see ``mapped_runtimes``)."""
    hip = mapped_runtimes()["amdhip64"]
    if len(hip) > 1:
        raise LanczosHipError(-4, "two HIP runtimes are mapped into this process (" + ", ".join(hip) + "): torch must be imported BEFORE "
                              "lanczos_amd.load_library() so that liblanczos_hip.so binds to the runtime torch brings, "
                              "or keep torch out of the process (SocketBootstrap needs none)")


def load_library(path=None):
    """dlopen the HIP extension and bind every declared symbol.  Raises
    ``LanczosHipError`` (never falls back) when the library is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    # The host driver of this pool only supports dmabuf IPC: without this RCCL's cross-process buffer sharing fails with
    # "hipIpcGetMemHandle: invalid argument".  Must be in the environment before the HIP runtime initialises.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not os.path.isfile(p):
        raise LanczosHipError(-6, f"HIP extension not built: {p} is missing (run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C lanczos_amd/csrc`)")
    try:
        # default (RTLD_LOCAL) scope: nothing in this library is meant to interpose on, or be interposed by, another
        # ROCm copy in the process
        lib = C.CDLL(p)
    except OSError as e:  # missing libamdhip64 etc.
        raise LanczosHipError(-6, f"cannot load {p}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def runtime_info():
    """{"hip": path of the libamdhip64 the library is bound to, "rccl": path of the librccl it opened ('' if none yet)}."""
    buf = C.create_string_buffer(1024)
    load_library().lz_runtime_info(buf, 1024)
    return dict(kv.split("=", 1) for kv in buf.value.decode().split(";"))


def _close_live_handles():
    # Runs from atexit, i.e. BEFORE the interpreter tears modules down and long before the HIP runtime's own exit
    # handlers / static destructors: device memory, streams and the RCCL communicator are released while the runtime
    # that owns them is still alive (Handle.__del__ at interpreter shutdown may run after it is gone).
    if _live_handles is not None:
        for h in list(_live_handles):
            try:
                h.close()
            except Exception:
                pass


def preload_rccl():
    """dlopen the system RCCL now (before any other library maps a different copy of librccl.so.1)."""
    lib = load_library()
    st = lib.lz_comm_load()
    if st != LZ_OK:
        raise LanczosHipError(st, lib.lz_last_error(None).decode())


def dptr(a):
    return a.ctypes.data_as(_D)


def i32ptr(a):
    return a.ctypes.data_as(_I32)


def i64ptr(a):
    return a.ctypes.data_as(_I64)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Handle:
    """Thin RAII wrapper over ``lz_handle``: raises on every non-zero status."""

    def __init__(self, device_id=0, lib=None):
        """``lib``: another build of the library (``load_library(KBENCH_LIB_PATH)``: the kernel-bench build); default the product."""
        self.lib = lib if lib is not None else load_library()
        self.matrix_uploads = 0  # lz_set_csr / lz_set_dense / stencil assemblies issued through this handle
        self._h = _P()
        st = self.lib.lz_create(C.byref(self._h), int(device_id))
        if st != LZ_OK:
            msg = self.lib.lz_last_error(None).decode()
            self._h = None
            raise LanczosHipError(st, msg)
        self._keep = []  # keeps ctypes callbacks alive
        self._reserve_lock = threading.Lock()  # lz_reserve may run on a helper thread: it must not meet lz_destroy
        self.breakdown = False
        global _live_handles
        if _live_handles is None:
            import atexit
            import weakref

            _live_handles = weakref.WeakSet()
            atexit.register(_close_live_handles)
        _live_handles.add(self)

    def check(self, st):
        if st != LZ_OK:
            raise LanczosHipError(st, self.lib.lz_last_error(self._h).decode())

    def close(self):
        lock = getattr(self, "_reserve_lock", None)
        if lock is not None:
            lock.acquire()
        try:
            if getattr(self, "_h", None):
                self.lib.lz_destroy(self._h)
                self._h = None
        finally:
            if lock is not None:
                lock.release()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- info / options
    def device_name(self):
        buf = C.create_string_buffer(256)
        self.check(self.lib.lz_device_name(self._h, buf, 256))
        return buf.value.decode()

    def set_options(self, flags):
        self.check(self.lib.lz_set_options(self._h, int(flags)))

    def set_tuning(self, index, value):
        self.check(self.lib.lz_set_tuning(self._h, int(index), int(value)))

    def synchronize(self):
        self.check(self.lib.lz_device_synchronize(self._h))

    def padded_rows(self, rows):
        return int(self.lib.lz_padded_rows(int(rows)))

    # -- communication
    def unique_id(self):
        buf = C.create_string_buffer(128)
        st = self.lib.lz_comm_unique_id(buf, 128)
        if st != LZ_OK:
            raise LanczosHipError(st, self.lib.lz_last_error(None).decode())
        return buf.raw

    def comm_init_rccl(self, world, rank, uid):
        buf = C.create_string_buffer(bytes(uid), 128)
        self.check(self.lib.lz_comm_init_rccl(self._h, int(world), int(rank), buf, 128))

    def comm_init_host(self, world, rank, allreduce, exchange=None, allgather=None):
        """``allreduce(np.ndarray)`` sums in place over ranks; ``exchange(peers, send_segments) -> recv_segments``;
        ``allgather(np.ndarray) -> np.ndarray (world*count)``."""

        def _ar(user, buf, count):
            try:
                a = np.ctypeslib.as_array(buf, shape=(count,))
                allreduce(a)
                return 0
            except Exception:  # pragma: no cover - surfaced as LZ_ERR_COMM
                import traceback

                traceback.print_exc()
                return 1

        def _ex(user, npeers, peers, sendbuf, scount, recvbuf, rcount):
            try:
                pr = [int(peers[i]) for i in range(npeers)]
                sc = [int(scount[i]) for i in range(npeers)]
                rc = [int(rcount[i]) for i in range(npeers)]
                s = np.ctypeslib.as_array(sendbuf, shape=(max(sum(sc), 1),))
                r = np.ctypeslib.as_array(recvbuf, shape=(max(sum(rc), 1),))
                so = np.concatenate([[0], np.cumsum(sc)]).astype(int)
                ro = np.concatenate([[0], np.cumsum(rc)]).astype(int)
                out = exchange(pr, [s[so[i] : so[i + 1]].copy() for i in range(npeers)], rc)
                for i in range(npeers):
                    r[ro[i] : ro[i + 1]] = out[i]
                return 0
            except Exception:  # pragma: no cover
                import traceback

                traceback.print_exc()
                return 1

        def _ag(user, sendbuf, recvbuf, count):
            try:
                s = np.ctypeslib.as_array(sendbuf, shape=(count,))
                r = np.ctypeslib.as_array(recvbuf, shape=(count * world,))
                r[:] = allgather(s.copy())
                return 0
            except Exception:  # pragma: no cover
                import traceback

                traceback.print_exc()
                return 1

        cbs = (HOST_ALLREDUCE_FN(_ar), HOST_EXCHANGE_FN(_ex), HOST_ALLGATHER_FN(_ag))
        self._keep.append(cbs)
        self.check(self.lib.lz_comm_init_host(self._h, int(world), int(rank), cbs[0], cbs[1], cbs[2], None))

    # -- matrix
    def set_csr(self, M_global, row0, rowptr, colidx, vals, ncols_ext=None):
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        colidx = np.ascontiguousarray(colidx, dtype=np.int32)
        vals = f64(vals)
        rows = len(rowptr) - 1
        nnz = int(rowptr[-1])
        if ncols_ext is None:
            ncols_ext = M_global
        self.check(self.lib.lz_set_csr(self._h, int(M_global), int(row0), rows, int(ncols_ext), nnz, i32ptr(rowptr), i32ptr(colidx), dptr(vals)))
        self.rows = rows
        self.matrix_uploads += 1

    def set_dense(self, A):
        A = f64(A)
        if A.ndim != 2 or A.shape[0] != A.shape[1]:
            raise ValueError("dense H must be square")
        self.check(self.lib.lz_set_dense(self._h, A.shape[0], dptr(A)))
        self.rows = A.shape[0]
        self.matrix_uploads += 1

    def set_dense_block(self, M_global, row0, A_block):
        """Row block of a dense matrix split over ranks; columns already in the all-gather layout (see the header)."""
        A_block = f64(A_block)
        rows, ncols_ext = A_block.shape
        self.check(self.lib.lz_set_dense_block(self._h, int(M_global), int(row0), rows, ncols_ext, dptr(A_block)))
        self.rows = rows
        self.matrix_uploads += 1

    def build_stencil3d(self, N, points, T_factor, weights4, potential=None, negate_T=False):
        w = f64(weights4)
        assert w.shape == (4,)
        pot = None if potential is None else f64(potential).reshape(-1)
        if pot is not None and pot.shape != (N**3,):
            raise ValueError("potential must have N^3 entries")
        self.check(self.lib.lz_build_stencil3d(self._h, int(N), int(points), float(T_factor), dptr(w), None if pot is None else dptr(pot), int(bool(negate_T))))
        self.rows = N**3

    def build_stencil3d_block(self, dims, points, T_factor, weights4, row0, rows_local, ghost_ranges=(), potential=None,
                              potential_params=None, negate_T=False):
        """Row block ``[row0, row0 + rows_local)`` of the periodic ``Nx x Ny x Nz`` stencil operator, assembled on the device.
        ``ghost_ranges``: ``[(global_start, length), ...]`` in ghost-tail order (``partition.plan_stencil_slab``).
        ``potential``: this rank's diagonal (host array) or ``potential_params`` = the 8 device-potential parameters."""
        Nx, Ny, Nz = (int(d) for d in dims)
        w = f64(weights4)
        assert w.shape == (4,)
        kind, pot = 0, None
        if potential_params is not None:
            kind, pot = 2, f64(potential_params)
            assert pot.shape == (8,)
        elif potential is not None:
            kind, pot = 1, f64(potential).reshape(-1)
            if pot.shape != (rows_local,):
                raise ValueError("potential must have one entry per local row")
        gs = np.ascontiguousarray([g[0] for g in ghost_ranges], dtype=np.int64)
        gl = np.ascontiguousarray([g[1] for g in ghost_ranges], dtype=np.int64)
        self.check(self.lib.lz_build_stencil3d_block(self._h, Nx, Ny, Nz, int(points), float(T_factor), dptr(w), kind,
                                                     None if pot is None else dptr(pot), int(bool(negate_T)), int(row0), int(rows_local),
                                                     len(gs), i64ptr(gs) if len(gs) else None, i64ptr(gl) if len(gl) else None))
        self.rows = int(rows_local)
        self.matrix_uploads += 1

    SPMV_PLANS = ("scalar", "csr-stream", "fixed-k", "two-phase", "dense")

    def spmv_plan(self):
        k = C.c_int()
        self.check(self.lib.lz_spmv_plan(self._h, C.byref(k)))
        return self.SPMV_PLANS[k.value]

    SPMV_CODINGS = ("none", "offsets", "offsets+values", "offsets+values, diagonal streamed")

    def spmv_coding(self):
        """row-class coding of a stencil matrix: ("none" | "offsets" | "offsets+values", number of classes)"""
        c, k = C.c_int(), C.c_int()
        self.check(self.lib.lz_spmv_coding(self._h, C.byref(c), C.byref(k)))
        return self.SPMV_CODINGS[c.value], k.value

    def get_csr(self):
        rows, nnz = C.c_int64(), C.c_int64()
        self.check(self.lib.lz_csr_info(self._h, C.byref(rows), C.byref(nnz)))
        rowptr = np.empty(rows.value + 1, dtype=np.int32)
        colidx = np.empty(nnz.value, dtype=np.int32)
        vals = np.empty(nnz.value)
        self.check(self.lib.lz_get_csr(self._h, i32ptr(rowptr), i32ptr(colidx), dptr(vals)))
        return rowptr, colidx, vals

    def set_halo(self, peers, send_counts, send_idx, recv_counts):
        peers = np.ascontiguousarray(peers, dtype=np.int32)
        sc = np.ascontiguousarray(send_counts, dtype=np.int64)
        rc = np.ascontiguousarray(recv_counts, dtype=np.int64)
        si = np.ascontiguousarray(send_idx, dtype=np.int32)
        self.check(self.lib.lz_set_halo(self._h, len(peers), i32ptr(peers), i64ptr(sc), i32ptr(si), i64ptr(rc)))

    def set_allgather(self, chunk):
        self.check(self.lib.lz_set_allgather(self._h, int(chunk)))

    # -- run
    def reserve(self, rows_local, n, with_ritz=True):
        """allocate the basis (and the Ritz vectors) of the coming run now; safe to call from a helper thread (see lz_reserve).
        ``with_ritz``: False the basis only, True both, 2 the Ritz vectors only."""
        with self._reserve_lock:
            if self._h:
                self.check(self.lib.lz_reserve(self._h, int(rows_local), int(n), 2 if with_ritz == 2 else (1 if with_ritz else 0)))

    def run(self, n, v0_local):
        v0 = f64(v0_local)
        if v0.shape != (self.rows,):
            raise ValueError("v0 has the wrong length")
        alpha = np.zeros(n)
        beta = np.zeros(max(n - 1, 1))
        st = self.lib.lz_run(self._h, int(n), dptr(v0), dptr(alpha), dptr(beta))
        self.breakdown = st == LZ_WARN_BREAKDOWN  # positive status: the run completed, coefficients as the reference's (inf/NaN)
        if not self.breakdown:
            self.check(st)
        self.n = n
        return alpha, beta[: n - 1]

    def get_residual(self):
        """r entering step n of the last run (this rank's rows): with the basis and alpha / beta, a checkpoint"""
        r = np.empty(self.rows)
        self.check(self.lib.lz_get_residual(self._h, dptr(r)))
        return r

    def get_omega_state(self):
        """the omega-recurrence state the device-decided partial loop (engine 7) left behind: part of its checkpoint"""
        n = int(self.n)
        out = np.empty(2 + (n + 2) + 3 * (n + 1))
        self.check(self.lib.lz_get_omega_state(self._h, dptr(out), out.size))
        return out

    def run_resume(self, n, V_rows, r, alpha, beta, omega_state=None):
        """continue a run of j0 = len(V_rows) completed steps to n steps in total; returns the full alpha (n), beta (n - 1).
        ``omega_state`` (from ``get_omega_state`` of the j0-step run): continue the device-decided partial loop."""
        V_rows, r, alpha, beta = f64(V_rows), f64(r), f64(alpha), f64(beta)
        j0 = V_rows.shape[0]
        if V_rows.shape != (j0, self.rows) or r.shape != (self.rows,) or alpha.shape != (j0,) or beta.shape != (max(j0 - 1, 0),):
            raise ValueError("checkpoint arrays have the wrong shapes")
        a_out, b_out = np.zeros(n), np.zeros(max(n - 1, 1))
        if beta.size == 0:
            beta = np.zeros(1)
        if omega_state is not None:
            om = f64(omega_state)
            if om.shape != (2 + (j0 + 2) + 3 * (j0 + 1),):
                raise ValueError("omega_state does not belong to a run of %d steps" % j0)
            st = self.lib.lz_run_resume_partial(self._h, int(n), int(j0), dptr(V_rows), self.rows, dptr(r), dptr(alpha), dptr(beta), dptr(om), dptr(a_out),
                                                dptr(b_out))
        else:
            st = self.lib.lz_run_resume(self._h, int(n), int(j0), dptr(V_rows), self.rows, dptr(r), dptr(alpha), dptr(beta), dptr(a_out), dptr(b_out))
        self.breakdown = st == LZ_WARN_BREAKDOWN
        if not self.breakdown:
            self.check(st)
        self.n = n
        return a_out, b_out[: n - 1]

    def get_basis(self):
        V = np.empty((self.n, self.rows))
        self.check(self.lib.lz_get_basis(self._h, dptr(V), self.rows))
        return V

    def get_basis_block(self, r0, r1):
        """(n, r1 - r0): the entries [r0, r1) of every basis vector"""
        r0, r1 = int(r0), int(r1)
        V = np.empty((self.n, r1 - r0))
        self.check(self.lib.lz_get_basis_block(self._h, r0, r1 - r0, dptr(V), r1 - r0))
        return V

    def ritz_vectors(self, S, fetch=True):
        S = f64(S)
        Y = np.empty((self.rows, self.n)) if fetch else None
        self.check(self.lib.lz_ritz_vectors(self._h, dptr(S), dptr(Y) if fetch else None))
        return Y

    def ritz_fetch(self):
        Y = np.empty((self.rows, self.n))
        self.check(self.lib.lz_get_ritz_vectors(self._h, dptr(Y)))
        return Y

    def ritz_fetch_rows(self, r0, r1):
        """rows [r0, r1) of the Ritz vectors, (r1 - r0, n): served from the resident Y or re-formed from the basis (chunked mode)"""
        r0, r1 = int(r0), int(r1)
        Y = np.empty((r1 - r0, self.n))
        self.check(self.lib.lz_get_ritz_rows(self._h, r0, r1 - r0, dptr(Y)))
        return Y

    def ritz_info(self):
        """{"chunk_rows": 0 (Y resident) or the rows per chunk of the chunked mode, "clock_mhz", "cycles_per_tile",
        "mfma_floor_cycles_per_tile", "tiles"}: the last four from the S-stationary kernel's own clock record (0 if another kernel ran)"""
        ch = C.c_int64()
        clk = np.zeros(4)
        self.check(self.lib.lz_ritz_info(self._h, C.byref(ch), dptr(clk)))
        return {"chunk_rows": int(ch.value), "clock_mhz": float(clk[0]), "cycles_per_tile": float(clk[1]), "mfma_floor_cycles_per_tile": float(clk[2]),
                "tiles": int(clk[3])}

    def ritz_gram(self):
        G = np.empty((self.n, self.n))
        self.check(self.lib.lz_ritz_gram(self._h, dptr(G)))
        return G

    def gram_info(self):
        """in-kernel clock record of the last ritz_gram (symmetric kernel only): held clock, cycles per k-step, MFMA issue floor"""
        c = (C.c_double * 4)()
        self.check(self.lib.lz_gram_info(self._h, c))
        return {"shader_clock_mhz": c[0], "cycles_per_kstep": c[1], "mfma_issue_floor_cycles_per_kstep": c[2], "ksteps": int(c[3])}

    def ritz_quality(self):
        q = np.empty(self.n)
        self.check(self.lib.lz_ritz_quality(self._h, dptr(q)))
        return q

    def last_sweeps(self):
        k = C.c_int()
        self.check(self.lib.lz_last_sweeps(self._h, C.byref(k)))
        return k.value

    def device_memory(self):
        """(free, total) bytes of the handle's GPU"""
        f, t = C.c_int64(), C.c_int64()
        self.check(self.lib.lz_device_memory(self._h, C.byref(f), C.byref(t)))
        return f.value, t.value

    def last_sweep_log(self, n=None):
        """per step of the last run: True where the re-orthogonalisation sweep ran (the device's own record for the partial loops)"""
        n = int(self.n if n is None else n)
        buf = (C.c_int * n)()
        self.check(self.lib.lz_last_sweep_log(self._h, buf, n))
        return np.array(buf[:], dtype=bool)

    def last_sweep_misses(self):
        """one-reduce partial loop: vectors the look-ahead gate should have swept and did not (swept one step late)"""
        k = C.c_int()
        self.check(self.lib.lz_last_sweep_misses(self._h, C.byref(k)))
        return k.value

    def last_engine(self):
        """"kernels": six launches per step; "fused": the three-launch path of small problems (second-stage reductions
        and the three-term recurrence folded into their consumer kernels); "three-term-fused": five launches per step (the
        three-term recurrence folded into pass 1; the default between the small problems and 4e6 rows per rank);
        "one-reduce": LZ_FLAG_ONE_REDUCE; "one-reduce-repeated": such a run whose cancellation guard fired and that was
        repeated on the default loop; "partial-device": LZ_FLAG_REORTH_PARTIAL with the omega-recurrence and the sweep decision on
        the device (no host synchronisation inside the run); "partial-one-reduce": the same with ONE all-reduce per step
        (LZ_FLAG_REORTH_PARTIAL | LZ_FLAG_ONE_REDUCE, look-ahead sweep decision); "step" / "small": the retired one-launch-per-step / one-kernel
        engines (kernel-bench build)."""
        k = C.c_int()
        self.check(self.lib.lz_last_engine(self._h, C.byref(k)))
        return ("kernels", "small", "fused", "three-term-fused", "step", "one-reduce-repeated", "one-reduce", "partial-device", "partial-one-reduce")[k.value]

    def last_host_syncs(self):
        """host <-> device synchronisations inside the last lz_run (between its first and its last launch)"""
        k = C.c_int64()
        self.check(self.lib.lz_last_host_syncs(self._h, C.byref(k)))
        return k.value

    def timings(self):
        t = LzTimings()
        self.check(self.lib.lz_get_timings(self._h, C.byref(t)))
        out = {"total_ms": t.total_ms}
        ar, ex = C.c_int64(), C.c_int64()
        self.check(self.lib.lz_comm_counts(self._h, C.byref(ar), C.byref(ex)))
        out["allreduces"], out["exchanges"] = ar.value, ex.value  # the collectives of this interval by kind
        for i, k in enumerate(KERNEL_CLASSES):
            out[k] = {"ms": t.ms[i], "timed_bytes": t.timed_bytes[i], "timed_launches": int(t.timed_launches[i]),
                      "bytes": t.bytes[i], "flops": t.flops[i], "launches": int(t.launches[i])}
        return out

    # -- single steps
    def basis_alloc(self, n):
        self.check(self.lib.lz_basis_alloc(self._h, int(n)))
        self.n = n

    def basis_set_row(self, j, row):
        row = f64(row)
        assert row.shape == (self.rows,)
        self.check(self.lib.lz_basis_set_row(self._h, int(j), dptr(row)))

    def basis_set_rows(self, j0, rows):
        """rows j0 .. j0 + len(rows) - 1 of the basis from a C-ordered (count, rows_local) array, in one strided copy"""
        rows = f64(rows)
        assert rows.ndim == 2 and rows.shape[1] == self.rows
        self.check(self.lib.lz_basis_set_rows(self._h, int(j0), int(rows.shape[0]), dptr(rows), int(rows.shape[1])))

    def basis_get_row(self, j):
        out = np.empty(self.rows)
        self.check(self.lib.lz_basis_get_row(self._h, int(j), dptr(out)))
        return out

    def r_set(self, r):
        r = f64(r)
        assert r.shape == (self.rows,)
        self.check(self.lib.lz_r_set(self._h, dptr(r)))

    def r_get(self):
        out = np.empty(self.rows)
        self.check(self.lib.lz_r_get(self._h, dptr(out)))
        return out

    def step_spmv(self, j):
        d = C.c_double()
        self.check(self.lib.lz_step_spmv(self._h, int(j), C.byref(d)))
        return d.value

    def step_reorth(self, j, nrows, scale=False):
        beta = C.c_double()
        c = np.empty(nrows)
        self.check(self.lib.lz_step_reorth(self._h, int(j), int(nrows), int(bool(scale)), C.byref(beta), dptr(c)))
        return (beta.value if scale else None), c

    def step_three_term(self, j, jm1, alpha, beta):
        d = C.c_double()
        self.check(self.lib.lz_step_three_term(self._h, int(j), int(jm1), float(alpha), float(beta), C.byref(d)))
        return d.value

    # ---- two-sided (bi-orthogonal) Lanczos ----
    def set_csr_transpose(self, rowptr=None, colidx=None, vals=None):
        """``None``: H is symmetric (H^T x runs on H)."""
        if rowptr is None:
            self.check(self.lib.lz_set_csr_transpose(self._h, 0, None, None, None))
            return
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        colidx = np.ascontiguousarray(colidx, dtype=np.int32)
        vals = f64(vals)
        self.check(self.lib.lz_set_csr_transpose(self._h, int(rowptr[-1]), i32ptr(rowptr), i32ptr(colidx), dptr(vals)))

    def run_two_sided(self, n, q0, p0):
        q0, p0 = f64(q0), f64(p0)
        if q0.shape != (self.rows,) or p0.shape != (self.rows,):
            raise ValueError("start pair has the wrong length")
        alpha, beta, gamma = np.zeros(n), np.zeros(max(n - 1, 1)), np.zeros(max(n - 1, 1))
        self.check(self.lib.lz_run_two_sided(self._h, int(n), dptr(q0), dptr(p0), dptr(alpha), dptr(beta), dptr(gamma)))
        self.n = n
        return alpha, beta[: n - 1], gamma[: n - 1]

    def bi_alloc(self, n):
        self.check(self.lib.lz_bi_alloc(self._h, int(n)))
        self.n = n

    def bi_set_row(self, which, j, row):
        row = f64(row)
        assert row.shape == (self.rows,)
        self.check(self.lib.lz_bi_set_row(self._h, int(which), int(j), dptr(row)))

    def bi_get_row(self, which, j):
        out = np.empty(self.rows)
        self.check(self.lib.lz_bi_get_row(self._h, int(which), int(j), dptr(out)))
        return out

    def step_bireorth(self, j):
        self.check(self.lib.lz_step_bireorth(self._h, int(j)))

    def step_bireorth_mem_safe(self, j):
        self.check(self.lib.lz_step_bireorth_mem_safe(self._h, int(j)))

    def spmv_host(self, x, ncols=None):
        x = f64(x)
        y = np.empty(self.rows)
        self.check(self.lib.lz_spmv_host(self._h, dptr(x), dptr(y)))
        return y
