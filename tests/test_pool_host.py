"""Host-side plumbing of ``Lanczos.devices`` (lanczos_amd/_pool.py, _worker.py) without a GPU: worker processes are
spawned, meet over the socket rendezvous, answer the protocol - and, this box having no GPU, the first command that
needs one fails in the workers with LZ_ERR_NODEVICE, which must come back to the caller as an exception (never a hang,
never a CPU fallback), with every worker ended and no /dev/shm segment left behind.  Plus the sanitizer build of the
C-ABI host shim (SURVEY.md section 5 hook): ``make SAN=1``, argument validation under ASan + UBSan."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

import lanczos_amd
from lanczos_amd import Lanczos, _capi, _pool, synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    import ctypes as C

    n = C.c_int(0)
    return lanczos_amd.load_library().lz_device_count(C.byref(n)) == 0 and n.value > 0


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU failure mode of the worker pool")
@pytest.mark.parametrize("world", [2, 3])
def test_worker_failure_reaches_the_caller_and_cleans_up(world):
    Lanczos.verbose = False
    before = set(glob.glob("/dev/shm/lz_*"))
    s = Lanczos(synthetic.laplacian_2d_5pt(16, 16).to_scipy())
    s.devices = [0] * world
    s.comm_backend = "host"
    with pytest.raises(lanczos_amd.LanczosHipError, match="LZ_ERR_NODEVICE"):
        s.execute_Lanczos(10)
    pool = s._handle.pool
    assert pool.closed and all(p.poll() is not None for p in pool.procs)
    assert set(glob.glob("/dev/shm/lz_*")) == before
    with pytest.raises(lanczos_amd.LanczosHipError, match="closed"):
        pool.request({"cmd": "ping"})
    s.close()


def test_pool_protocol_ping_and_close():
    """rendezvous + one command round trip + orderly shutdown of two workers (no GPU call is made by ``ping``)"""
    pool = _pool.WorkerPool([0, 0], backend="host")
    rep = pool.request({"cmd": "ping"}, timeout=120)
    assert [r["rank"] for r in rep] == [0, 1]
    assert all(os.path.basename(r["runtime"]["hip"]).startswith("libamdhip64") for r in rep)
    pool.close()
    assert all(p.returncode == 0 for p in pool.procs)


def test_unknown_command_is_an_error_not_a_hang():
    pool = _pool.WorkerPool([0], backend="host")
    with pytest.raises(lanczos_amd.LanczosHipError, match="cmd_nonsense"):
        pool.request({"cmd": "nonsense"}, timeout=120)
    assert pool.closed


def test_stalled_worker_meets_its_deadline():
    """Round 4: a STALLED rank (here: stopped with SIGSTOP mid-protocol - what a collective that never completes looks like
    from outside) used to hang the caller for ever; every command now has a deadline.  On expiry: LanczosHipError naming the
    silent rank, every child ended (the silent one with SIGKILL), no /dev/shm segment left."""
    import signal
    import time

    before = set(glob.glob("/dev/shm/lz_*"))
    h = _pool.PoolHandle([0, 0], backend="host")
    pool = h.pool
    assert [r["rank"] for r in pool.request({"cmd": "ping"}, timeout=120)] == [0, 1]
    # deadlines derived from the work: never below the floor, growing with bytes / steps; the override wins
    h.rows, h.n = 10_000_000, 200
    assert h.deadline() == h.floor_s and h.deadline(host_bytes=16e9) > h.floor_s + 30 and h._run_deadline(200) > h._run_deadline(20) > h.floor_s
    h.timeout_override = 3.0
    assert h.deadline(host_bytes=1e12) == 3.0
    os.kill(pool.procs[1].pid, signal.SIGSTOP)
    t = time.time()
    with pytest.raises(lanczos_amd.LanczosHipError, match=r"rank\(s\) \[1\].*did not answer within 3 s"):
        pool.request({"cmd": "ping"}, timeout=h.deadline(host_bytes=1e12))  # rank 0 answers, rank 1 never does
    assert time.time() - t < 30.0
    assert pool.closed and all(p.poll() is not None for p in pool.procs)
    assert set(glob.glob("/dev/shm/lz_*")) == before


def test_stencil_operator_descriptor():
    op = lanczos_amd.StencilOperator((6, 5, 4), 7)
    assert op.shape == (120, 120) and op.key() == lanczos_amd.StencilOperator((6, 5, 4), 7).key()
    assert op.key() != lanczos_amd.StencilOperator((6, 5, 4), 27).key()
    # the potential is part of the key by CONTENT: an in-place change (a parameter scan) must invalidate the device copy
    pot = np.linspace(0.0, 1.0, 120)
    op_p = lanczos_amd.StencilOperator((6, 5, 4), 7, potential=pot)
    k0 = op_p.key()
    assert k0 == lanczos_amd.StencilOperator((6, 5, 4), 7, potential=pot.copy()).key()
    pot[17] += 1.0
    assert op_p.key() != k0
    s = Lanczos(op)
    assert s.M == 120
    with pytest.raises(ValueError, match="one entry per grid point"):
        lanczos_amd.StencilOperator((6, 5, 4), 7, potential=np.zeros(7))


def test_asan_build_of_the_host_shim(tmp_path):
    """``make SAN=1`` builds liblanczos_hip_asan.so (host code under AddressSanitizer + UBSan; device code unchanged); a child
    process with the sanitizer runtime preloaded drives the C ABI's argument validation and error paths: a clean exit."""
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if not os.path.isfile(clang):
        pytest.skip("no ROCm clang")
    rt = subprocess.run([clang, "--print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isfile(rt):
        pytest.skip("no ASan runtime")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "lanczos_amd", "csrc"), "-j4", "SAN=1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    code = r"""
import ctypes as C, sys
sys.path.insert(0, %r)
from lanczos_amd import _capi
lib = _capi.load_library(%r)
n = C.c_int(-1)
st = lib.lz_device_count(C.byref(n))
assert lib.lz_set_tuning(None, 1, 21) == -1 and lib.lz_set_options(None, 0) == -1 and lib.lz_destroy(None) == 0
assert lib.lz_padded_rows(33) == 64
h = C.c_void_p()
st = lib.lz_create(C.byref(h), 0)
if st != 0:  # no GPU here: every entry point must refuse a NULL handle, and lz_create must explain itself
    assert st == -6 and b"no HIP device" in lib.lz_last_error(None)
    for name in ("lz_run", "lz_get_basis", "lz_ritz_vectors", "lz_ritz_gram", "lz_get_ritz_rows", "lz_ritz_info", "lz_get_basis_block"):
        pass
    assert lib.lz_run(None, 5, None, None, None) == -1
    assert lib.lz_get_ritz_rows(None, 0, 1, None) == -1 and lib.lz_ritz_info(None, None, None) == -1
    assert lib.lz_set_csr(None, 4, 0, 4, 4, 0, None, None, None) == -1
else:
    import numpy as np
    hd = _capi.Handle(0, lib=lib)
    hd.close()
    lib.lz_destroy(h)
buf = C.create_string_buffer(8)
assert lib.lz_runtime_info(buf, 8) == 0  # truncated, never overrun
print("ASAN-OK")
""" % (ROOT, os.path.join(ROOT, "lanczos_amd", "liblanczos_hip_asan.so"))
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "ASAN-OK" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr
