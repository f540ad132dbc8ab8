"""CPU-side checks of the boundary: the shared library loads without a GPU, exports every
symbol include/lanczos_hip.h declares, fails loudly (no CPU fallback), and the Python mirror
reproduces the reference's call surface and error behaviour."""
import ctypes
import inspect
import os
import re

import ctypes as C

import numpy as np
import pytest

import lanczos_amd
from lanczos_amd import IrrLanczos, Lanczos, _capi, synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "lanczos_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lz_[a-z0-9_]+)\s*\(", text)) - {"lz_host_allreduce_fn", "lz_host_exchange_fn", "lz_host_allgather_fn"})


def test_library_exports_every_declared_symbol():
    lib = lanczos_amd.load_library()
    names = declared_symbols()
    assert len(names) >= 30
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/lanczos_hip.h but not exported"
        assert name in _capi.SIGNATURES, f"{name} has no ctypes signature"
    assert sorted(_capi.SIGNATURES) == names
    assert lib.lz_version() >= 100
    assert lib.lz_padded_rows(33) == 64 and lib.lz_padded_rows(64) == 64


def has_gpu():
    c = ctypes.c_int(0)
    return lanczos_amd.load_library().lz_device_count(ctypes.byref(c)) == 0 and c.value > 0


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly_not_silently():
    with pytest.raises(lanczos_amd.LanczosHipError):
        _capi.Handle(0)
    Lanczos.verbose = False
    s = Lanczos(synthetic.laplacian_2d_5pt(8, 8).to_scipy())
    with pytest.raises(lanczos_amd.LanczosHipError):
        s.execute_Lanczos(4)  # no CPU fallback
    assert not s.Lanczos_has_been_executed


def test_missing_extension_raises(tmp_path):
    with pytest.raises(lanczos_amd.LanczosHipError, match="not built"):
        _capi.load_library(str(tmp_path / "liblanczos_hip.so"))


def test_product_never_imports_the_oracle_or_torch():
    pkg = os.path.join(ROOT, "lanczos_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), fn
            if fn != "distributed.py":
                assert "import torch" not in src, fn
    # ... and nothing outside tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg touches oracle/ at all (tools/, examples/)
    for sub in ("tools", os.path.join("tools", "runs"), "examples"):
        d = os.path.join(ROOT, sub)
        for fn in os.listdir(d):
            if fn.endswith((".py", ".sh")):
                src = open(os.path.join(d, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) and "oracle/" not in src.replace("oracle/gen_golden", ""), (sub, fn)
    bench = open(os.path.join(ROOT, "bench.py")).read()
    hits = [m.start() for m in re.finditer(r"^\s*(?:from|import)\s+oracle\b", bench, flags=re.M)]
    assert len(hits) == 1 and bench[: hits[0]].rsplit("\ndef ", 1)[1].startswith("cpu_baseline(")


def test_call_surface_matches_reference():
    # signatures to keep
    sig = inspect.signature(Lanczos.execute_Lanczos)
    assert list(sig.parameters) == ["self", "n", "seed", "use_cuda", "v0"]
    assert sig.parameters["seed"].default == 99 and sig.parameters["use_cuda"].default is True and sig.parameters["v0"].default is None
    assert list(inspect.signature(IrrLanczos.execute_LanczosOld).parameters) == ["self", "n", "seed", "use_cuda", "v0"]
    assert list(inspect.signature(IrrLanczos.execute_Lanczos).parameters) == ["self", "n", "seed", "use_cuda", "v0", "dtype"]
    assert list(inspect.signature(Lanczos.print_good_eigs).parameters) == ["self", "tol", "print_nr", "print_bad"]
    assert list(inspect.signature(IrrLanczos.print_good_eigs).parameters) == ["self", "tol", "print_nr", "print_bad", "normal_eq"]
    assert list(inspect.signature(Lanczos.reorthogonalize).parameters) == ["V", "j", "use_cuda"]
    assert list(inspect.signature(IrrLanczos.bireorthogonalize).parameters) == ["V1", "V2", "q_basis", "p_basis", "j", "use_cuda", "mem_safe"]
    for name in ("get_H_eigs", "find_exact_eigs", "compare_eigs", "get_matched_eigs", "test_is_Hermitian", "test_is_normalized",
                 "test_is_orthogonal", "test_is_eigvecs"):
        assert hasattr(Lanczos, name) and hasattr(IrrLanczos, name)
    for name in ("get_H_eigsOld", "print_good_eigsOld"):
        assert hasattr(IrrLanczos, name)
    for prop in ("H_eff", "V", "H_eigvals", "H_eigvecs", "H_eigvals_actual", "H_eigvecs_actual"):
        assert isinstance(getattr(Lanczos, prop), property)


def test_error_behaviour_before_any_device_work(capsys):
    Lanczos.verbose = IrrLanczos.verbose = False
    H = synthetic.laplacian_2d_5pt(8, 8).to_scipy()
    s = Lanczos(H)
    assert s.M == 64 and s.H is H and not s.Lanczos_has_been_executed
    for prop in ("H_eff", "V"):
        with pytest.raises(ValueError, match="Lanczos Algorithm has not been called."):
            getattr(s, prop)
    with pytest.raises(ValueError, match="Lanczos Algorithm has not been called."):
        s.get_H_eigs()
    with pytest.raises(ValueError, match="Lanczos Algorithm has not been called."):
        s.compare_eigs()
    with pytest.raises(ValueError, match="n cannot be larger than M!"):
        s.execute_Lanczos(65)
    # use_cuda=False (3Ddeuteron.py:95): accepted - it announces itself and takes the ONLY compute path, the HIP one (so on
    # this GPU-less box it fails loudly there, never in a NumPy fallback); strict_use_cuda restores round 2's refusal
    s.strict_use_cuda = True
    with pytest.raises(NotImplementedError, match="device path only"):
        s.execute_Lanczos(10, use_cuda=False)
    s.strict_use_cuda = False
    if not has_gpu():
        with pytest.raises(lanczos_amd.LanczosHipError, match="LZ_ERR_NODEVICE"):
            s.execute_Lanczos(10, use_cuda=False)
        assert "use_cuda=False: lanczos_amd has no NumPy path" in capsys.readouterr().out
    with pytest.raises(IndexError):
        s.execute_Lanczos(1)
    t = IrrLanczos(H)
    # two-sided variant: argument errors come first (as in IrrLanczos.py:78-104), then - with no GPU here - the
    # device path fails loudly instead of falling back to a CPU implementation
    with pytest.raises(ValueError, match="n cannot be larger than M!"):
        t.execute_Lanczos(65)
    with pytest.raises(UnboundLocalError):
        t.execute_Lanczos(10, v0=np.ones(64))
    t.strict_use_cuda = True
    with pytest.raises(NotImplementedError, match="device path only"):
        t.execute_Lanczos(10, use_cuda=False)
    t.strict_use_cuda = False
    t.devices = [0, 1]  # the two-sided variant is single-GPU
    with pytest.raises(NotImplementedError, match="one GPU"):
        t.execute_Lanczos(10)
    t.devices = None
    import lanczos_amd._capi as capi
    ndev = C.c_int(-1)
    if capi.load_library().lz_device_count(C.byref(ndev)) != 0 or ndev.value == 0:
        with pytest.raises(capi.LanczosHipError, match="LZ_ERR_NODEVICE"):
            t.execute_Lanczos(10)
    with pytest.raises(ValueError, match="n cannot be larger than M!"):
        t.execute_LanczosOld(100)


def test_host_side_diagnostics_match_reference_semantics():
    rng = np.random.default_rng(0)
    Q, _ = np.linalg.qr(rng.standard_normal((50, 6)))
    assert abs(Lanczos.test_is_normalized(Q, no_assert=True) - 1) < 1e-14
    assert Lanczos.test_is_orthogonal(Q, no_assert=True) < 1e-7
    Q2 = Q.copy()
    Q2[:, 2] = Q2[:, 1]
    with pytest.raises(AssertionError, match="VECTORS 1 AND 2 NOT ORTHOGONAL"):
        Lanczos.test_is_orthogonal(Q2)
    # like the reference, only the column whose norm is CLOSEST to 1 is asserted
    Q3 = Q * np.array([1, 5, 1, 1, 1, 1.0])
    Lanczos.test_is_normalized(Q3)
    with pytest.raises(AssertionError, match="IS NOT NORMALIZED"):
        Lanczos.test_is_normalized(Q * 2)
    A = rng.standard_normal((50, 50))
    A = A + A.T
    l, v = np.linalg.eigh(A)
    vs, vLs, ls, lLs = Lanczos.get_matched_eigs(v, v[:, [3, 1]], l, l[[3, 1]])
    assert set(np.round(ls, 12)) == set(np.round(l[[3, 1]], 12)) and np.allclose(ls, lLs)
    Lanczos.test_is_Hermitian(A)
    with pytest.raises(AssertionError):
        Lanczos.test_is_Hermitian(np.triu(A))


def _build_c_example(tmp_path):
    import subprocess

    exe = str(tmp_path / "c_abi_example")
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_example.c"),
           "-L" + os.path.join(ROOT, "lanczos_amd"), "-llanczos_hip", "-Wl,-rpath," + os.path.join(ROOT, "lanczos_amd"), "-lm", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True)
    return exe


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_header_is_plain_c_and_client_links(tmp_path):
    """include/lanczos_hip.h compiles as C99 and a plain-C client links against the shared library; without a GPU the
    client gets LZ_ERR_NODEVICE from lz_create (exit code 3) - no CPU fallback."""
    import subprocess

    exe = _build_c_example(tmp_path)
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode == 3 and "no HIP device" in p.stderr


@pytest.mark.gpu
def test_plain_c_client_runs(tmp_path):
    import subprocess

    exe = _build_c_example(tmp_path)
    p = subprocess.run([exe, "64", "48", "24"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "sweeps=24" in p.stdout
    assert "ritz rows (resident, chunk_rows=0): 0 mismatches" in p.stdout and "ritz rows (chunked, chunk_rows=512): 0 mismatches" in p.stdout
    assert "resume 24 -> 48 steps vs one run of 48: 0 mismatches" in p.stdout


def test_product_library_has_no_ablation_kernels():
    """Round 5 deleted the timing-only ablation arms (kernel template parameter ABL != 0: wrong results on purpose) from the sources
    of BOTH builds - the hot kernels no longer carry an ABL parameter at all - and moved the retired, bit-identity-tested A/B kernels
    into kernel-bench-only files (lz_*_kbench.h, lz_small.hip).  The kernels' mangled names are embedded in the shared library (host
    stubs + code-object symbol table): the product library must hold the live kernels under their ABL-free signatures and none of the
    retired ones; lz_set_tuning refuses the knob values that used to select either."""
    blob = open(_capi.LIB_PATH, "rb").read()
    # k_qtw_mfma4<SCALE, U, T>, k_qtw_valu<SCALE, R, U, NT>, k_spmv_stream<FIXED_K>, k_gemm_tn_sl2<NT, KS, USE4>, k_gemm_tn_sreg<NT, KS>
    assert re.search(rb"k_qtw_mfma4ILi\d+ELi\d+ELi\d+EE", blob), "default Q^T w kernel not found - naming changed?"
    assert not re.search(rb"k_qtw_mfma4ILi\d+ELi\d+ELi\d+ELi\d+EE", blob)
    assert re.search(rb"k_qtw_valuILi\d+ELi\d+ELi\d+ELi\d+EE", blob) and not re.search(rb"k_qtw_valuILi\d+ELi\d+ELi\d+ELi\d+ELi\d+EE", blob)
    assert re.search(rb"k_spmv_streamILi\d+EE", blob) and not re.search(rb"k_spmv_streamILi\d+ELi\d+EE", blob)
    assert re.search(rb"k_gemm_tn_sl2ILi\d+ELi\d+ELb[01]EE", blob), "S-in-LDS Ritz kernel not found - naming changed?"
    assert not re.search(rb"k_gemm_tn_sl2ILi\d+ELi\d+ELi\d+", blob)
    assert re.search(rb"k_gemm_tn_sregILi\d+ELi\d+EE", blob) and not re.search(rb"k_gemm_tn_sregILi\d+ELi\d+ELi\d+", blob)
    assert b"k_pb_rows" in blob and not re.search(rb"k_pb_rowsILi", blob)
    for retired in (rb"k_small_run", rb"k_small_step", rb"k_gemm_tn_persist", rb"k_gemm_tn_ldsI", rb"k_qtw_mfmaILi", rb"k_gemm_tn_slILi"):
        assert retired not in blob, retired
    if os.path.isfile(_capi.KBENCH_LIB_PATH):  # ... which the kernel-bench build still carries (without ablation parameters either)
        kblob = open(_capi.KBENCH_LIB_PATH, "rb").read()
        for retired in (rb"k_small_run", rb"k_gemm_tn_persist", rb"k_gemm_tn_ldsI", rb"k_qtw_mfmaILi", rb"k_gemm_tn_slILi"):
            assert retired in kblob, retired
        assert not re.search(rb"k_qtw_mfma4ILi\d+ELi\d+ELi\d+ELi\d+EE", kblob) and not re.search(rb"k_pb_rowsILi", kblob)
    # the sources say the same: no ABL template parameter left anywhere, the retired kernels only in kernel-bench-only files
    csrc = os.path.join(ROOT, "lanczos_amd", "csrc")
    for fn in os.listdir(csrc):
        if fn.endswith((".hip", ".h")):
            src = open(os.path.join(csrc, fn)).read()
            assert "ABL" not in src, fn
            if not (fn.endswith("_kbench.h") or fn == "lz_small.hip"):
                for name in ("k_gemm_tn_persist", "k_gemm_tn_lds", "k_small_run"):
                    assert ("void " + name) not in src, (fn, name)
    lib = lanczos_amd.load_library()
    assert lib.lz_set_tuning(None, 1, 21) == -1  # (no handle on a CPU box; with one: tests/test_gpu_lanczos.py)


def test_two_hip_runtimes_are_detected_and_refused():
    """Round-1 teardown abort, root cause: PyTorch bundles its own libamdhip64 / libhsa-runtime64 /
    librccl under torch/lib and looks them up by UNVERSIONED names, which never match the system libraries' sonames - so
    importing torch AFTER liblanczos_hip.so maps a second HIP runtime into the process.  The opposite order gives one
    runtime (torch's), to which the library binds and from whose directory it takes RCCL.  Checked in child processes
    (this needs no GPU: it is pure dynamic-loader behaviour)."""
    import json
    import subprocess
    import sys

    code = r"""
import json, sys
sys.path.insert(0, %r)
order = sys.argv[1]
import lanczos_amd
from lanczos_amd import _capi, distributed
if order == "lz_first":
    lanczos_amd.load_library()
    import torch
else:
    import torch
    lanczos_amd.load_library()
out = {"maps": _capi.mapped_runtimes(), "info": _capi.runtime_info()}
try:
    _capi.check_single_runtime()
    out["refused"] = False
except _capi.LanczosHipError as e:
    out["refused"] = True
try:
    distributed.TorchBootstrap(init=False)
    out["bootstrap_refused"] = False
except _capi.LanczosHipError:
    out["bootstrap_refused"] = True
except Exception as e:  # process group not initialised etc.: the runtime check passed
    out["bootstrap_refused"] = False
print("OUT" + json.dumps(out))
""" % ROOT
    res = {}
    for order in ("lz_first", "torch_first"):
        p = subprocess.run([sys.executable, "-c", code, order], capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        res[order] = json.loads([l for l in p.stdout.splitlines() if l.startswith("OUT")][0][3:])
    bad, good = res["lz_first"], res["torch_first"]
    assert len(bad["maps"]["amdhip64"]) == 2 and len(bad["maps"]["hsa-runtime64"]) == 2 and bad["refused"] and bad["bootstrap_refused"]
    assert len(good["maps"]["amdhip64"]) == 1 and not good["refused"] and not good["bootstrap_refused"]
    assert "torch/lib" in good["info"]["hip"] and good["maps"]["amdhip64"] == [os.path.realpath(good["info"]["hip"])] or good["info"]["hip"] in good["maps"]["amdhip64"]


def test_checkpoint_matrix_key_is_independent_of_the_container_format():
    """a checkpoint names its operator by one fixed hash of the canonical CSR form, so CSR / CSC / COO / dense holders of
    the same matrix (and the CSR copy `execute_Lanczos` leaves in ``self.H``, Lanczos.py:137) are the same matrix to `resume_Lanczos`,
    with or without xxhash on the host; another matrix of the same size is not."""
    import scipy.sparse

    from lanczos_amd import _solver

    H = synthetic.laplacian_2d_5pt(12, 9).to_scipy()
    key = _solver._canonical_key(H)
    assert key.startswith("csr-blake2b:")
    for other in (H.tocsc(), H.tocoo(), H.toarray(), scipy.sparse.csr_matrix(H.toarray()), synthetic.laplacian_2d_5pt(12, 9)):
        assert _solver._canonical_key(other) == key
    assert _solver._canonical_key(H + scipy.sparse.identity(H.shape[0], format="csr")) != key
    assert _solver._canonical_key(synthetic.laplacian_2d_5pt(9, 12).to_scipy()) != key
