"""Host-side mirror of the reference's Lanczos class surface over the HIP library.

Mirrors Python/Regular/Lanczos.py:11-337 and the symmetric ("Old") half of
Python/Irregular/IrrLanczos.py:12-554 (paths relative to /root/reference):
same constructor, method names, keyword arguments, attributes, error types and
messages.  Everything numerical inside ``execute_Lanczos`` / ``get_H_eigs`` runs
in liblanczos_hip.so; this file only prepares inputs (start vector with NumPy's
legacy RNG exactly like the reference's CPU branch, CSR packing), assembles the
small tridiagonal ``H_eff``, calls ``numpy.linalg.eigh`` on it (as the reference
does, also in its GPU mode, Lanczos.py:151) and formats the diagnostics.

There is deliberately NO CPU implementation of the recurrence here: the only
compute path is the HIP library, and a missing extension or GPU raises
``LanczosHipError``.  ``use_cuda=False`` (the reference's NumPy branch, which its
``3Ddeuteron.py:95`` asks for) is accepted and ALSO runs the HIP path, with a
one-line notice - the two branches of the reference agree to rounding and this
build is held to its NumPy branch anyway; ``strict_use_cuda = True`` restores the
``NotImplementedError``.

Multi-GPU: ``Lanczos.devices = [0, 1, ..., 7]`` (class or instance attribute;
default ``None`` = one GPU, in this process) keeps the same calls and partitions
H and the Krylov basis row-block-wise over one worker process per GPU (see
``_pool.py``).
"""
from __future__ import annotations

import json

import numpy as np
import scipy.sparse
import scipy.sparse.linalg

from . import _capi

_NOT_EXECUTED = "Lanczos Algorithm has not been called."
_CPU_MSG = (
    "lanczos_amd implements the device path only (use_cuda=True runs on the MI355X through liblanczos_hip.so); "
    "use_cuda=False is the reference's own NumPy path and is not re-implemented here (strict_use_cuda is set)"
)
_CPU_NOTICE = "+++ use_cuda=False: lanczos_amd has no NumPy path; running the HIP path on the MI355X (same results to rounding)."


def _use_cuda_or_notice(obj, use_cuda):
    """``use_cuda=False`` handling shared by every entry point that takes the flag (Lanczos.py:75,89-91,233)."""
    if use_cuda:
        return
    if getattr(obj, "strict_use_cuda", LanczosBase.strict_use_cuda):
        raise NotImplementedError(_CPU_MSG)
    print(_CPU_NOTICE)


_HASH_CHUNK = 32 << 20
_hash_pool = None


def _fingerprint_start(*arrays):
    """Content hash of the matrix arrays: decides whether the device copy of H is still current.  Every byte is hashed (an
    in-place change of any entry is seen), in 32 MB chunks on a few threads (xxhash releases the GIL): ~0.02 s for the
    headline's 640 MB of CSR instead of the 0.1 s of one thread.  Returns a function that waits for the digest - the caller
    draws the start vector meanwhile (NumPy's legacy RNG releases the GIL too)."""
    views = []
    for a in arrays:
        a = np.ascontiguousarray(a)
        views.append(memoryview(a).cast("B") if a.size else memoryview(b""))
    try:
        import xxhash

        def one(mv):
            return xxhash.xxh3_64_intdigest(mv)
    except ImportError:  # pragma: no cover - xxhash is optional
        import zlib

        def one(mv):
            return zlib.crc32(mv)
    chunks = [mv[o:o + _HASH_CHUNK] for mv in views for o in range(0, max(len(mv), 1), _HASH_CHUNK)]
    sizes = tuple(len(mv) for mv in views)
    if len(chunks) <= 2:
        digests = tuple(one(c) for c in chunks)
        return lambda: (sizes, digests)
    global _hash_pool
    if _hash_pool is None:
        import os
        from concurrent.futures import ThreadPoolExecutor

        _hash_pool = ThreadPoolExecutor(max_workers=max(1, min(8, (os.cpu_count() or 2) // 2)), thread_name_prefix="lz-hash")
    futures = [_hash_pool.submit(one, c) for c in chunks]
    return lambda: (sizes, tuple(f.result() for f in futures))


def _fingerprint(*arrays):
    return _fingerprint_start(*arrays)()


def _canonical_key(H):
    """What a checkpoint says about the operator it was made with: one fixed hash (BLAKE2b, stdlib) over the CANONICAL form - the
    packed int32 / float64 CSR arrays - so that the same operator held as CSR, CSC, COO or dense (or the CSR copy `execute_Lanczos`
    leaves in ``self.H``, Lanczos.py:137), on a host with or without xxhash, resumes.  (The device-cache key `_matrix_key` is hashed
    from the arrays as the caller holds them and stays what it was: it only has to recognise the very same object again, fast.)"""
    import hashlib

    if hasattr(H, "dims") and hasattr(H, "points"):
        return "stencil:" + json.dumps(H.key(), default=str)
    packed = _pack_matrix(H)
    if packed[0] == "dense":
        A = scipy.sparse.csr_matrix(packed[1])
        packed = ("csr", A.indptr.astype(np.int32, copy=False), A.indices.astype(np.int32, copy=False), A.data)
    hsh = hashlib.blake2b(digest_size=16)
    hsh.update(np.asarray(np.shape(H), dtype=np.int64).tobytes())
    for a in packed[1:]:
        a = np.ascontiguousarray(a)
        hsh.update(np.int64(a.size).tobytes())
        if a.size:
            hsh.update(memoryview(a).cast("B"))
    return "csr-blake2b:" + hsh.hexdigest()


def _native_arrays(H):
    """the arrays that ARE the matrix as the caller holds it (no conversion, no copy): what the cache key is hashed from"""
    if scipy.sparse.issparse(H):
        if H.format in ("csr", "csc", "bsr"):
            return (H.format, H.indptr, H.indices, H.data)
        if H.format == "coo":
            return ("coo", H.row, H.col, H.data)
        return None
    if hasattr(H, "rowptr") and hasattr(H, "colidx"):
        return ("csr", H.rowptr, H.colidx, H.vals)
    if isinstance(H, np.ndarray):
        return ("dense", H)
    return None


def _pack_matrix(H):
    """-> ("csr", rowptr, colidx, vals) or ("dense", A).  Keeps the stored entry
    order of a CSR input (the per-row summation order then equals SciPy's)."""
    if scipy.sparse.issparse(H):
        A = H if H.format == "csr" else H.tocsr()
        if A.dtype != np.float64:
            A = A.astype(np.float64)
        if A.shape[0] != A.shape[1]:
            raise ValueError("H must be square")
        if A.nnz >= 2**31 or A.shape[0] >= 2**31:
            raise ValueError("matrix too large for int32 CSR indices")
        return ("csr", A.indptr.astype(np.int32, copy=False), A.indices.astype(np.int32, copy=False), A.data)
    if hasattr(H, "rowptr") and hasattr(H, "colidx"):  # lanczos_amd.synthetic.CSR
        return ("csr", H.rowptr, H.colidx, H.vals)
    A = np.asarray(H, dtype=np.float64)
    if A.ndim != 2 or A.shape[0] != A.shape[1]:
        raise ValueError("H must be a square matrix")
    return ("dense", np.ascontiguousarray(A))


class LanczosBase:
    """State, lazy properties and diagnostics shared by ``Lanczos`` and ``IrrLanczos``."""

    verbose = True  # the reference prints "+++ ..." progress lines; set False to silence
    device_id = 0
    options = 0     # lz_flags forwarded to lz_set_options (see include/lanczos_hip.h)
    fused_norm = True  # beta = ||r|| travels with the Q^T r partial sums (c_i = (V_i.r)/beta, V[j] = r/beta formed in the update
                       # kernel): one pass less over V[j] and one launch less per step, +2 % at the headline; False = scale first,
                       # then dot, exactly in the reference's order (the coefficients differ by one rounding of a division)
    reorth = "full"  # "full" = the reference's sweep at every step; "partial" = opt-in Simon partial
                     # re-orthogonalisation (same sweep kernels, run only when semi-orthogonality is about to be lost)
    devices = None   # None / one entry: this process drives that GPU.  Several entries, e.g. list(range(8)): H and the Krylov
                     # basis are row-block partitioned over one FRESH worker process per listed GPU (RCCL all-reduces of
                     # alpha / ||r||^2 / Q^T w, halo or all-gather exchange of the SpMV input); same calls, same results
    comm_backend = "rccl"  # "host": collectives staged through host memory (several workers on ONE GPU: tests)
    worker_timeout = None  # with ``devices``: seconds every worker has to answer a command; None = derived from the work of the
                           # command with pessimistic rates, never below 60 s (a stalled rank ends the pool with LanczosHipError)
    strict_use_cuda = False  # True: use_cuda=False raises NotImplementedError instead of running the HIP path with a notice
    cache_matrix = True  # keep H on the device across execute_Lanczos calls while its content hash is unchanged
    _check_eigs = ("normalized", "orthogonal")  # which asserts get_H_eigs runs (Lanczos.py:157-158)

    def __init__(self, H):
        self.H = H
        self.M = H.shape[0] if hasattr(H, "shape") else np.shape(H)[0]
        self.Lanczos_has_been_executed = False
        self.H_eigs_have_been_found = False
        self.H_exact_eigs_have_been_found = False
        self._handle = None
        self._matrix_key = self._matrix_key_alt = None
        self._V = None
        self._H_eigvecs_host = None
        self._timings = None

    def close(self):
        """Release the device memory (and, with ``devices``, end the worker processes) now rather than at exit."""
        self._join_bg()
        if self._handle is not None:
            self._handle.close()
            self._handle = None
            self._matrix_key = self._matrix_key_alt = None
            self._reserved = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _say(self, msg):
        if self.verbose:
            print(msg)

    def _join_bg(self):
        """wait for the helper thread that reserves device buffers (started by ``_execute``)"""
        bg = getattr(self, "_bg", None)
        if bg is not None:
            bg.join()
            self._bg = None

    def _device(self):
        """the live handle behind the lazily fetched results (V, H_eigvecs, windows of them)"""
        self._join_bg()
        if self._handle is None:
            raise _capi.LanczosHipError(-4, "the device state of this run was released (close()); run execute_Lanczos again")
        return self._handle

    # ------------------------------------------------------------------ properties
    @property
    def H_eff(self):
        if not self.Lanczos_has_been_executed:
            raise ValueError(_NOT_EXECUTED)
        return self._H_eff

    @property
    def V(self):
        """(M, n) Krylov basis, columns = Lanczos vectors; like the reference this is
        the transposed view of an (n, M) array.  Copied off the device on first use."""
        if not self.Lanczos_has_been_executed:
            raise ValueError(_NOT_EXECUTED)
        if self._V is None:
            self._V = self._device().get_basis().T
        return self._V

    @property
    def H_eigvecs(self):
        if not self.H_eigs_have_been_found:
            self.get_H_eigs()
        return self._H_eigvecs

    @property
    def H_eigvals(self):
        if not self.H_eigs_have_been_found:
            self.get_H_eigs()
        return self._H_eigvals

    @property
    def H_eigvals_actual(self):
        if not self.H_exact_eigs_have_been_found:
            self.find_exact_eigs()
            self.H_exact_eigs_have_been_found = True
        return self._H_eigvals_actual

    @property
    def H_eigvecs_actual(self):
        if not self.H_exact_eigs_have_been_found:
            self.find_exact_eigs()
            self.H_exact_eigs_have_been_found = True
        return self._H_eigvecs_actual

    @property
    def timings(self):
        """Per-kernel-class device timings of the last run (only with ``options |= FLAG_PROFILE``)."""
        return self._timings

    def find_exact_eigs(self, nr_vecs=20):
        self._say("+++ Calculating exact eigs using scipy.sparse.linalg.eigsh.")
        H = self.H.to_scipy() if hasattr(self.H, "to_scipy") else self.H  # (a StencilOperator / synthetic.CSR is materialised for SciPy)
        self._H_eigvals_actual, self._H_eigvecs_actual = scipy.sparse.linalg.eigsh(H, k=nr_vecs, which="SM")
        self._say("+++ Finished calculating exact eigs.")

    # ------------------------------------------------------------------ the hot path
    def _multi(self):
        return self.devices is not None and len(self.devices) > 1

    def _get_handle(self):
        """The object that owns the device state: a ``_capi.Handle`` (this process, one GPU) or a ``_pool.PoolHandle``
        (one worker process per entry of ``devices``).  Re-created when ``devices`` changes."""
        want = tuple(int(d) for d in self.devices) if self.devices is not None else (int(self.device_id),)
        if self._handle is not None and getattr(self, "_handle_devices", None) != (want, self.comm_backend):
            self.close()
        if self._handle is None:
            if len(want) > 1:
                from . import _pool

                self._handle = _pool.PoolHandle(want, self.comm_backend)
            else:
                self._handle = _capi.Handle(want[0])
            self._handle_devices = (want, self.comm_backend)
            self._matrix_key = self._matrix_key_alt = None
            self._reserved = None
        if hasattr(self._handle, "timeout_override"):
            self._handle.timeout_override = self.worker_timeout
        return self._handle

    def _matrix_key_start(self):
        """Start hashing H as the caller holds it (threads); returns a function that delivers the cache key (or None)."""
        H = self.H
        if not self.cache_matrix or (hasattr(H, "dims") and hasattr(H, "points")):
            return lambda: None
        native = _native_arrays(H)
        if native is None:
            return lambda: None
        wait = _fingerprint_start(*native[1:])
        shape = tuple(np.shape(H))
        return lambda: (native[0], shape, wait())

    def _upload_matrix(self, h, pending_key=None):
        """H -> device(s), skipped when the device copy is still current (same format, shape and content hash): a second
        ``execute_Lanczos`` on one object then costs no repack, no H2D and no SpMV-plan rebuild (C3: ~1 GB + 2.6 GB of layout)."""
        H = self.H
        if hasattr(H, "dims") and hasattr(H, "points"):  # _pool.StencilOperator: assembled on the device(s), never on the host
            key = H.key()
            if not (self.cache_matrix and key == self._matrix_key):
                if self._multi():
                    h.set_stencil(H)
                else:
                    h.build_stencil3d_block(H.dims, H.points, H.T_factor, H.weights4, 0, H.shape[0], (), potential=H.potential,
                                            potential_params=H.potential_params, negate_T=H.negate_T)
                self._matrix_key, self._matrix_key_alt = key, None
            return
        # The cache key is hashed from the arrays the caller holds (CSR, CSC, COO, dense ...), BEFORE any conversion: a second
        # call on an unchanged H costs one threaded hash, no tocsr(), no repack (round 3 packed first and hashed the result)
        if pending_key is not None:
            key = pending_key()
        else:
            key = self._matrix_key_start()()
        if key is not None and key in (self._matrix_key, self._matrix_key_alt):
            return
        packed = _pack_matrix(H)
        if key is None and self.cache_matrix:
            key = (packed[0], tuple(np.shape(H)), _fingerprint(*packed[1:]))
            if key in (self._matrix_key, self._matrix_key_alt):
                return
        if packed[0] == "csr":
            h.set_csr(self.M, 0, packed[1], packed[2], packed[3])
        else:
            h.set_dense(packed[1])
        self._matrix_key, self._matrix_key_alt = key, None

    def _execute(self, n, seed, use_cuda, v0):
        if n > self.M:
            raise ValueError("n cannot be larger than M!")
        self._say("+++ Executing Lanczos algorithm")
        self.n = n
        _use_cuda_or_notice(self, use_cuda)
        M = self.M

        # start vector: the reference's CPU-branch stream (global legacy RNG), Lanczos.py:93-100
        import time

        t_0 = time.perf_counter()
        self._join_bg()
        pending_key = self._matrix_key_start()  # hashes H on helper threads while this thread draws v0
        # ... and the 8nM-byte basis (+ the Ritz vectors) is allocated on another one: a first hipMalloc of 16 GB takes 0.1-0.5 s
        reserve = None
        if n >= 2 and self.reorth in ("full", "partial") and not self._multi() and getattr(self, "_reserved", None) != (M, n):
            import threading

            hres = self._get_handle()
            basis_ready = threading.Event()
            matrix_on_device = threading.Event()  # set when the upload thread is through (or was never started)
            run_started = threading.Event()       # set when this thread enters lz_run (or gives up)

            def _reserve():
                # Order matters where the device has to CLEAR the memory it hands out (a box whose free memory is "dirty": 16 GB take
                # ~0.7 s there, and the small allocations of the matrix upload queue up behind the big one - measured 1.4 s of
                # `device alloc` in lz_set_csr, tools/first_call_probe.py): first the matrix (0.6 GB, on the upload thread), then the
                # basis, and the Ritz vectors only once the solve is running - nothing waits for them before `get_H_eigs`.
                matrix_on_device.wait(60.0)
                try:
                    hres.reserve(M, n, with_ritz=False)  # the basis: lz_run needs it
                except Exception:  # never fatal: lz_run allocates (and reports) itself
                    pass
                basis_ready.set()
                run_started.wait(60.0)
                time.sleep(0.03)  # (lz_run enqueues its whole loop in the first 3 - 10 ms: keep the allocator out of the launches' way)
                try:
                    hres.reserve(M, n, with_ritz=2)  # the Ritz vectors ONLY: while the solve runs (lz_reserve touches only its own fields)
                except Exception:
                    pass

            reserve = threading.Thread(target=_reserve, name="lz-reserve", daemon=True)
            reserve.start()
            self._bg = reserve
        # ... and the matrix is packed, validated and uploaded on a third while this thread draws the start vector (round 5: the draw
        # and normalisation of 1e7 legacy-RNG doubles is 0.05 - 0.14 s, the upload 0.05 - 0.08 s; both release the GIL).  Argument errors
        # keep the reference's order - they are raised after the RNG has been seeded and drawn from - so the upload only starts when
        # none is coming.
        args_ok = n >= 2 and self.reorth in ("full", "partial")
        upload, upload_err, h = None, [], None
        if args_ok:
            import threading

            h = self._get_handle()
            h.set_options(self.options | (_capi.FLAG_REORTH_PARTIAL if self.reorth == "partial" else 0)
                          | (_capi.FLAG_FUSED_NORM if self.fused_norm else 0))

            def _upload():
                try:
                    self._upload_matrix(h, pending_key)
                except BaseException as e:  # re-raised in the calling thread
                    upload_err.append(e)
                finally:
                    if reserve is not None:
                        matrix_on_device.set()

            upload = threading.Thread(target=_upload, name="lz-upload", daemon=True)
            upload.start()
        try:
            cached = getattr(self, "_v0_cache", None)
            if v0 is None and cached is not None and cached[0] == (seed, M):
                # The default start vector is a pure function of (seed, M): a repeated call reuses the normalised vector of the last
                # one (drawing 1e7 legacy-RNG doubles is 0.05 s - the whole overhead of a second call) and leaves the GLOBAL RNG exactly
                # where the reference's `np.random.seed(seed); np.random.uniform(-1, 1, M)` would: the state saved right after the draw.
                np.random.set_state(cached[2])
                v0 = cached[1]
            else:
                np.random.seed(seed)
                if v0 is None:
                    v0 = np.random.uniform(-1, 1, size=(M))
                    state = np.random.get_state()
                    v0 /= np.linalg.norm(v0)
                    self._v0_cache = ((seed, M), v0, state)  # (never written to again: lz_run only reads it)
                else:
                    v0 = np.array(v0)
                    v0 = v0 / np.linalg.norm(v0)  # out of place, like the reference (Lanczos.py:100): an integer or list v0 becomes float64 here
            t_1 = time.perf_counter()
            if n < 2:
                # the reference allocates beta = zeros(n-1) and writes beta[-1] at j = 0 (Lanczos.py:107,112)
                raise IndexError("index -1 is out of bounds for axis 0 with size 0")
            if self.reorth not in ("full", "partial"):
                raise ValueError("reorth must be 'full' or 'partial'")
        finally:
            if reserve is not None and upload is None:
                matrix_on_device.set()  # (no upload thread: nothing to wait for)
            if upload is not None:
                upload.join()
            if reserve is not None:  # the basis must be reserved (or given up on) before lz_run looks for it
                basis_ready.wait()
                self._reserved = (M, n)
                run_started.set()  # (also on the error paths: the helper thread must not sit out its timeout)
        if upload_err:
            raise upload_err[0]
        t_2 = time.perf_counter()
        alpha, beta = h.run(n, v0)
        t_3 = time.perf_counter()
        # where the caller's wall time went, host side (seconds): drawing / normalising v0, deciding whether the device copy of
        # H is current (+ pack, validate and upload when it is not), lz_run (upload of v0, the solve, alpha / beta back)
        self.host_timings = {"start_vector_s": t_1 - t_0, "matrix_s": t_2 - t_1, "run_s": t_3 - t_2}
        if h.breakdown:  # lz_run returned LZ_WARN_BREAKDOWN
            # The reference divides by beta blindly (Lanczos.py:113): an exhausted Krylov space gives inf/NaN there too.
            import warnings

            warnings.warn("Lanczos breakdown: a residual norm beta reached zero (invariant subspace); H_eff contains "
                          "non-finite entries, exactly as the reference's would", RuntimeWarning, stacklevel=3)
        self._timings = h.timings()
        self.sweeps = h.last_sweeps()  # steps that ran the re-orthogonalisation sweep (== n for reorth="full")

        # H_eff (Lanczos.py:121-130): symmetric tridiagonal, assembled on the host from 2n-1 doubles
        H_eff = np.zeros((n, n))
        idx = np.arange(n)
        H_eff[idx, idx] = alpha
        H_eff[idx[:-1], idx[1:]] = beta
        H_eff[idx[1:], idx[:-1]] = beta
        self._alpha, self._beta = alpha, beta
        self._H_eff = H_eff
        self._V = None
        self._H_eigvecs_host = None
        # the reference's GPU branch leaves a SciPy CSR in self.H (Lanczos.py:137); a stencil descriptor stays what it is
        # (materialising it is exactly what it exists to avoid)
        if not (hasattr(self.H, "dims") and hasattr(self.H, "points")):
            newH = scipy.sparse.csr_matrix(self.H.to_scipy() if hasattr(self.H, "to_scipy") else self.H, dtype=np.float64)
            if self.cache_matrix and self._matrix_key is not None and self._matrix_key_alt is None:
                # what the caller handed over stays resident (a dense matrix keeps its dense GEMV); the CSR copy now in self.H
                # is the same operator: remember its key too, unless it is the very same arrays (a CSR input: no second hash)
                old = _native_arrays(self.H)
                same = old is not None and old[0] == "csr" and all(
                    getattr(a, "ctypes", None) is not None and a.ctypes.data == b.ctypes.data and a.nbytes == b.nbytes
                    for a, b in zip(old[1:], (newH.indptr, newH.indices, newH.data)))
                if not same:
                    self._matrix_key_alt = ("csr", tuple(newH.shape), _fingerprint(newH.indptr, newH.indices, newH.data))
            self.H = newH
        self.H_eigs_have_been_found = False
        self._say("+++ Lanczos executed successfully.")
        self.Lanczos_has_been_executed = True

    # ------------------------------------------------------------------ checkpoint / resume (extension; SURVEY.md section 5 hook)
    def checkpoint(self):
        """The state a finished run of n steps leaves behind - enough to continue it later with ``resume_Lanczos``:
        ``{"alpha" (n), "beta" (n - 1), "V" (n, M) row-major, "r" (M), "M", "fused_norm"}`` (+ ``"omega_state"`` after a
        ``reorth="partial"`` run on one device).  With ``devices`` the ranks write their rows into one host array, so a checkpoint does
        not depend on the partition that made it."""
        if not self.Lanczos_has_been_executed:
            raise ValueError(_NOT_EXECUTED)
        h = self._device()
        ck = {"alpha": self._alpha.copy(), "beta": self._beta.copy(), "V": np.array(h.get_basis()), "r": h.get_residual(), "M": self.M,
              "fused_norm": bool(self.fused_norm), "options": int(self.options), "reorth": str(self.reorth),
              # what the run was made WITH: resume refuses a different operator of the same size (hash of its canonical CSR form)
              "matrix_key": _canonical_key(self.H)}
        if self.reorth == "partial" and h.last_engine() == "partial-device" and hasattr(h, "get_omega_state"):
            # the omega-recurrence of the device-decided selective loop: with it the continued run takes the same sweep decisions
            ck["omega_state"] = h.get_omega_state()
        return ck

    def save_checkpoint(self, path):
        np.savez(path, **self.checkpoint())

    def resume_Lanczos(self, n, checkpoint):
        """Continue the run stored in ``checkpoint`` (a dict from ``checkpoint()`` or the path of a ``save_checkpoint`` file) to
        ``n`` steps in total.  The result - ``H_eff``, ``V``, Ritz pairs - is bit-identical to ``execute_Lanczos(n)`` run in one go
        with the same start vector (tests/test_gpu_lanczos.py)."""
        ck = dict(np.load(checkpoint, allow_pickle=False)) if isinstance(checkpoint, (str, bytes)) or hasattr(checkpoint, "__fspath__") else checkpoint
        j0 = len(ck["alpha"])
        if int(ck["M"]) != self.M:
            raise ValueError("the checkpoint belongs to a matrix of another size")
        if n > self.M:
            raise ValueError("n cannot be larger than M!")
        if n <= j0:
            raise ValueError("resume_Lanczos: n must exceed the %d steps already in the checkpoint" % j0)
        if self.reorth not in ("full", "partial"):
            raise NotImplementedError("resume needs reorth='full' or 'partial'")
        if "reorth" in ck and str(ck["reorth"]) != self.reorth:
            raise ValueError("the checkpoint was written by a reorth='%s' run, this object has reorth='%s'" % (str(ck["reorth"]), self.reorth))
        omega_state = None
        if self.reorth == "partial":
            if "omega_state" not in ck:
                raise NotImplementedError("this checkpoint of a reorth='partial' run carries no omega-recurrence state (written by the "
                                          "one-reduce arm, on several devices, or before round 5): it cannot be continued")
            if self.devices is not None and len(self.devices) > 1:
                raise NotImplementedError("resume of a reorth='partial' run: one device only")
            omega_state = ck["omega_state"]
        if "options" in ck and int(ck["options"]) & ~_capi.FLAG_PROFILE != int(self.options) & ~_capi.FLAG_PROFILE:
            raise ValueError("the checkpoint was written with options=%d, this object has options=%d" % (int(ck["options"]), int(self.options)))
        self._say("+++ Executing Lanczos algorithm")
        self.n = n
        fused = bool(ck["fused_norm"])  # the norm order of the run being continued (this object's own setting is left alone)
        h = self._get_handle()
        h.set_options(self.options | (_capi.FLAG_REORTH_PARTIAL if self.reorth == "partial" else 0) | (_capi.FLAG_FUSED_NORM if fused else 0))
        self._upload_matrix(h)
        stored = str(ck["matrix_key"]) if "matrix_key" in ck else ""
        if stored.startswith(("csr-blake2b:", "stencil:")):
            # format-independent: CSR / CSC / COO / dense holders of one operator (and `Lanczos(old.H)`) all resume
            if stored != _canonical_key(self.H):
                raise ValueError("the checkpoint belongs to a different matrix (same size, different content)")
        elif stored and self._matrix_key is not None and stored != json.dumps(self._matrix_key, default=str):
            # a checkpoint of an earlier version (key of the arrays as the caller held them): a mismatch may be a change of container
            # format or of the hash function only, so it is reported, not refused
            import warnings

            warnings.warn("the checkpoint's matrix key is in the pre-round-5 format and does not match this object's H as held now; "
                          "resuming on the caller's word that it is the same operator", RuntimeWarning, stacklevel=2)
        if omega_state is not None:
            alpha, beta = h.run_resume(n, ck["V"], ck["r"], ck["alpha"], ck["beta"], omega_state=omega_state)
        else:
            alpha, beta = h.run_resume(n, ck["V"], ck["r"], ck["alpha"], ck["beta"])
        if h.breakdown:
            import warnings

            warnings.warn("Lanczos breakdown: a residual norm beta reached zero (invariant subspace)", RuntimeWarning, stacklevel=2)
        self._timings = h.timings()
        self.sweeps = h.last_sweeps()
        H_eff = np.zeros((n, n))
        idx = np.arange(n)
        H_eff[idx, idx] = alpha
        H_eff[idx[:-1], idx[1:]] = beta
        H_eff[idx[1:], idx[:-1]] = beta
        self._alpha, self._beta, self._H_eff = alpha, beta, H_eff
        self._V = None
        self._H_eigvecs_host = None
        if not (hasattr(self.H, "dims") and hasattr(self.H, "points")):
            self.H = scipy.sparse.csr_matrix(self.H.to_scipy() if hasattr(self.H, "to_scipy") else self.H, dtype=np.float64)
        self.H_eigs_have_been_found = False
        self._say("+++ Lanczos executed successfully.")
        self.Lanczos_has_been_executed = True

    def _ritz(self, checks):
        if not self.Lanczos_has_been_executed:
            raise ValueError(_NOT_EXECUTED)
        self._say("+++ Converting eigenvectors from H_eff to H basis.")
        H_eff_eigvals, H_eff_eigvecs = np.linalg.eigh(self.H_eff)
        # Y = V S on the device (FP64 MFMA GEMM), Lanczos.py:153-156.  Y stays on the device; the two checks of
        # Lanczos.py:157-158 only need its n x n Gram matrix, which the device forms too.  When a second M x n array does
        # not fit beside the basis (BASELINE C4 on one GPU) the library keeps S and re-forms Y in row chunks for the Gram
        # matrix and for every later fetch (lz_ritz_info): nothing here depends on free device memory.
        self._device().ritz_vectors(H_eff_eigvecs, fetch=False)
        self._H_eff_eigvecs = H_eff_eigvecs
        self._H_eigvecs_host = None
        if checks:
            G = self._handle.ritz_gram()
            norms = np.sqrt(np.diag(G))
            if "normalized" in checks:
                pick = norms[np.argmin(np.abs(norms - 1))]  # like test_is_normalized: the norm CLOSEST to 1
                assert np.abs(pick - 1) < 0.001, "VECTOR HAS NORM %.4f. IS NOT NORMALIZED." % pick
            if "orthogonal" in checks:
                off = np.abs(G - np.diag(np.diag(G)))
                a, b = np.unravel_index(np.argmax(off), off.shape)
                worst = np.sqrt(off[a, b])
                assert worst < 0.01, "VECTORS %d AND %d NOT ORTHOGONAL! INNER PRODUCT %.4f" % (a, b, worst)
        self._H_eigvals = H_eff_eigvals
        self._say("+++ Finished Converting.")
        self.H_eigs_have_been_found = True

    @property
    def _H_eigvecs(self):
        """(M, n) Ritz vectors, C-order; copied off the device on first use."""
        if self._H_eigvecs_host is None:
            self._H_eigvecs_host = self._device().ritz_fetch()
        return self._H_eigvecs_host

    def V_rows(self, lo, hi):
        """Rows ``[lo, hi)`` of ``V`` (an (hi - lo, n) block) without moving the whole basis off the device (extension)."""
        if not self.Lanczos_has_been_executed:
            raise ValueError(_NOT_EXECUTED)
        return self._device().get_basis_block(lo, hi).T

    def H_eigvecs_rows(self, lo, hi):
        """Rows ``[lo, hi)`` of ``H_eigvecs`` without materialising the whole (M, n) array on the host (extension: at
        BASELINE C4 size ``H_eigvecs`` alone is 160 GB)."""
        if not self.H_eigs_have_been_found:
            self.get_H_eigs()
        return self._device().ritz_fetch_rows(lo, hi)

    def get_H_eigs(self):
        self._ritz(self._check_eigs)

    # ------------------------------------------------------------------ diagnostics (host, NumPy)
    def _eigvec_quality(self):
        """cos^2 between H x / |H x| and x for every Ritz vector (Lanczos.py:169-175), computed on the device-resident Ritz
        vectors (lz_ritz_quality)."""
        if not self.H_eigs_have_been_found:
            self.get_H_eigs()
        return self._device().ritz_quality()  # device kernels for CSR and dense, one rank or a partition: no host fallback

    def print_good_eigs(self, tol=0.01, print_nr=20, print_bad=True):
        """Print the first ``print_nr`` Ritz values with the eigenvector quality
        ``((Hx/|Hx|) . x)^2``; lines failing ``|1 - q| < tol`` are tagged BAD."""
        eigvals, q = self.H_eigvals, self._eigvec_quality()
        print("__________EIGENVALUE AND EIGVENVECTOR COMPARISON__________")
        print("%12s %12s" % ("Eigval", "Eigvec InnerProd"))
        for i in range(print_nr):
            tag = "" if abs(1 - q[i]) < tol else " --- BAD"
            print("%12.4f %20.14f%s" % (eigvals[i], q[i], tag))

    def _match_to_exact(self):
        va, la = self.H_eigvecs_actual, self.H_eigvals_actual
        ve, le = self.H_eigvecs, self.H_eigvals
        nr = len(la)
        overlap = (ve.T @ va) ** 2  # (n, nr)
        est = np.full(nr, np.nan)
        best = np.full(nr, np.nan)
        who = np.full(nr, np.nan)
        for i in range(self.n):
            k = int(overlap[i].argmax())
            if np.isnan(best[k]) or overlap[i, k] > best[k]:
                est[k], best[k], who[k] = le[i], overlap[i, k], i
        return la, est, best, who

    def compare_eigs(self):
        """Table of exact (``eigsh``) vs Lanczos eigenpairs matched by largest eigenvector overlap."""
        if not self.Lanczos_has_been_executed:
            raise ValueError(_NOT_EXECUTED)
        print("+++ Comparing to exact eigs.")
        la, est, best, who = self._match_to_exact()
        pct = np.abs((la - est) / est) * 100
        print("__________EIGENVALUE AND EIGVENVECTOR COMPARISON__________")
        print("%6s %6s %20s %20s %14s %14s" % ("Idx1", "Idx2", "Actual", "Lanczos", "% Diff", "Eigvec Prod"))
        for i in range(len(la)):
            print("%6.0d %6.0f %20.10f %20.10f %14.4f %14.4f" % (i, who[i], la[i], est[i], pct[i], best[i]))

    @staticmethod
    def reorthogonalize(V, j, use_cuda=True):
        """In-place single-pass Gram-Schmidt of row ``j`` of the (n, M) array ``V`` against
        all rows, with the reference's arithmetic ``V[j] = 2 V[j] - (V V[j])^T V``
        (Lanczos.py:247-249), executed by the HIP kernels."""
        _use_cuda_or_notice(LanczosBase, use_cuda)
        V = np.asarray(V)
        n, M = V.shape
        h = _capi.Handle(LanczosBase.device_id)
        try:
            eye_ptr = np.arange(M + 1, dtype=np.int32)
            h.set_csr(M, 0, eye_ptr, eye_ptr[:-1], np.ones(M))  # length carrier only; no matvec is run
            h.basis_alloc(n)
            h.basis_set_rows(0, V)  # one strided copy (row by row until round 5: n synchronous copies per call)
            h.step_reorth(j, n, scale=False)
            V[j] = h.basis_get_row(j)
        finally:
            h.close()

    @staticmethod
    def get_matched_eigs(v, vL, l, lL):
        """Order estimated eigenpairs (vL, lL) by their best overlap with exact ones (v, l);
        returns ``(v_sorted, vL_sorted, l_sorted, lL_sorted)``, best match first."""
        overlap = (vL.T @ v) ** 2
        partner = overlap.argmax(axis=1)
        quality = overlap[np.arange(len(lL)), partner]
        order = quality.argsort()[::-1]
        return v[:, partner[order]], vL[:, order], l[partner[order]], lL[order]

    @staticmethod
    def test_is_Hermitian(A):
        """Assert A equals its transpose."""
        At = A.T
        same = (A != At).nnz == 0 if scipy.sparse.issparse(A) else bool((np.asarray(A) == np.asarray(At)).all())
        assert same, "A IS NOT HERMITIAN!"

    @staticmethod
    def test_is_normalized(V, tol=0.001, no_assert=False):
        """Column norms of the (M, m) matrix V; like the reference (Lanczos.py:288-305) only the
        column whose norm is CLOSEST to 1 is reported/asserted."""
        norms = np.linalg.norm(V, axis=0)
        pick = norms[np.argmin(np.abs(norms - 1))]
        if no_assert:
            return pick
        assert np.abs(pick - 1) < tol, "VECTOR HAS NORM %.4f. IS NOT NORMALIZED." % pick

    @staticmethod
    def test_is_orthogonal(V, tol=0.01, no_assert=False):
        """sqrt of the largest off-diagonal |V^T V| entry (Lanczos.py:307-323)."""
        G = V.T @ V
        G[np.diag_indices_from(G)] -= np.linalg.norm(V, axis=0) ** 2
        G = np.abs(G)
        a, b = np.unravel_index(np.argmax(G), G.shape)
        worst = np.sqrt(G[a, b])
        if no_assert:
            return worst
        assert worst < tol, "VECTORS %d AND %d NOT ORTHOGONAL! INNER PRODUCT %.4f" % (a, b, worst)

    @staticmethod
    def test_is_eigvecs(A, V, tol=0.01, no_assert=False):
        """Spread of (A v)/v per column; the reference asserts ``max > tol`` (sic, Lanczos.py:326-337)."""
        R = (np.asarray(A @ V)) / V
        worst = np.max(R.max(axis=0) - R.min(axis=0))
        if no_assert:
            return worst
        assert worst > tol, "VECTOR NOT EIGENVECTOR."
