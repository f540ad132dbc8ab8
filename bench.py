#!/usr/bin/env python3
"""Headline benchmark: Lanczos iterations/s (+ per-kernel HBM roofline) on the
M = 1e7 2-D periodic 5-point Laplacian, k = 200, full re-orthogonalisation, fp64.

    python bench.py --gpus N --steps K --warmup W

A "step" is ONE complete Lanczos solve (k iterations: k SpMVs, k three-term
updates, k full re-orthogonalisations against the growing basis) on the
device-resident matrix; value = K * k / time.  N > 1 (launched with
torch.distributed.run) row-partitions the SAME problem over N GPUs (strong
scaling) with RCCL all-reduces and neighbour halo exchange issued by
liblanczos_hip.so; the control plane is used only for rendezvous,
barriers and the max-over-ranks of the timing (by default a torch-free Unix-socket
bootstrap that reads the launcher's RANK / WORLD_SIZE / MASTER_PORT).

Per-kernel numbers come from hipEvents recorded around every launch on the
library's compute stream inside the timed region (LZ_FLAG_PROFILE);
`roofline.achieved` = algorithmic bytes / event time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
FP64_MFMA_PEAK_TFLOPS = 78.6

WORKLOADS = {
    # name: (kind, dims, k)
    "lap2d_5pt_M1e7_k200": ("lap2d", (4000, 2500), 200),   # headline (BASELINE.json metric)
    "lap2d_5pt_M1e6_k100": ("lap2d", (1000, 1000), 100),   # configs[1]
    "lap3d_7pt_M1e8_k200": ("lap3d", (500, 500, 400), 200),  # configs[3] (8 GPUs)
    "lap2d_5pt_M1e7_k500": ("lap2d", (4000, 2500), 500),   # configs[4] (8 GPUs)
    "graph_M1e7_k200": ("graph", (10_000_000, 35_000_000), 200),  # configs[2]
    "lap2d_5pt_M1.25e6_k200": ("lap2d", (4000, 313), 200),  # one rank's share of the headline at N = 8 (compute floor)
    "lap2d_5pt_M2.5e6_k200": ("lap2d", (4000, 625), 200),   # ... at N = 4
    "lap2d_5pt_M5e6_k200": ("lap2d", (4000, 1250), 200),    # ... at N = 2
    "deuteron3d_N160_27pt_k400": ("deuteron27", (160,), 400),  # the reference's largest configured run (3Ddeuteron.py:63-95)
    "dense_M512_k20": ("dense_c1", (512,), 20),             # configs[0]: 512 x 512 random symmetric
    "dense_M32768_k100": ("dense", (32768,), 100),          # synthetic dense, 8.6 GB matrix (GEMV-bound)
    "dense_M131072_k100": ("dense", (131072,), 100),        # 137 GB matrix: 8 GPUs (17 GB block per rank)
    "tiny": ("lap2d", (256, 128), 24),
}


def workload_rows(kind, dims):
    if kind == "graph" or kind.startswith("dense"):
        return dims[0]
    return dims[0] ** 3 if kind == "deuteron27" else int(np.prod(dims))


def build_local(kind, dims, lo, hi):
    from lanczos_amd import synthetic

    if kind == "lap2d":
        return synthetic.laplacian_2d_5pt(*dims, rows=(lo, hi))
    if kind == "lap3d":
        return synthetic.laplacian_3d_7pt(*dims, rows=(lo, hi))
    if kind == "graph":
        full = synthetic.random_graph_laplacian(dims[0], dims[1], seed=1234)
        return full.row_slice(lo, hi)
    if kind == "dense":  # counter-based symmetric entries: each rank generates only its row block
        return synthetic.dense_symmetric_hashed(dims[0], rows=(lo, hi))
    if kind == "dense_c1":  # default_rng(0).standard_normal, (A + A^T) / 2
        return np.ascontiguousarray(synthetic.dense_symmetric(dims[0], seed=0)[lo:hi])
    if kind == "deuteron27":  # H = -T + V of 3Ddeuteron.py, assembled on the device by the Hamiltonian mirror
        from lanczos_amd import Hamiltonian

        N = dims[0]
        Hamiltonian.verbose = False
        Hamiltonian.vectorize_potential = True
        ham = Hamiltonian(N, 25, synthetic.deuteron_potential, 197.327**2 / (2 * 469.4592) / (25.0 / N) ** 2)
        H = ham.build_H("27")
        return synthetic.CSR(H.indptr, H.indices, H.data, H.shape).row_slice(lo, hi)
    raise ValueError(kind)


def gram_record(tg, k):
    """bench-line record of lz_ritz_gram (the device Gram matrix behind get_H_eigs' two checks)"""
    if not tg:
        return None
    rec = {"ms": round(tg["ms"], 3), "tflops_symmetric_half": round(tg["flops"] / max(tg["ms"], 1e-9) / 1e9, 2), "bound": "mfma",
           "peak_tflops": FP64_MFMA_PEAK_TFLOPS, "frac": round(tg["flops"] / max(tg["ms"], 1e-9) / 1e9 / FP64_MFMA_PEAK_TFLOPS, 4),
           "max_dev_from_identity": tg["max_dev_from_identity"],
           "note": "G = Y^T Y (k x k) of the resident Ritz vectors: upper 16 x 16 tiles in accumulators, Y streamed once, mirrored; flops counted as "
                   "M k (k + 1) (the symmetric half - the full product would be 2 M k^2); second call of the process; kernel + slice sum"}
    info = tg.get("info")
    if info and info.get("shader_clock_mhz", 0) > 0:
        rec.update({"shader_clock_mhz": round(info["shader_clock_mhz"], 1), "cycles_per_kstep": round(info["cycles_per_kstep"], 1),
                    "mfma_issue_floor_cycles_per_kstep": info["mfma_issue_floor_cycles_per_kstep"],
                    "mfma_issue_utilisation_in_cycles": round(info["mfma_issue_floor_cycles_per_kstep"] / max(info["cycles_per_kstep"], 1e-9), 4)})
    return rec


def spmv_format_arm(_capi, local, device, coding):
    """A/B beside a row-class coded SpMV (round 5): the SAME matrix through the uncoded kernel (knob 17 = 1: the CSR-order fixed-K kernel /
    CSR-stream, 12 bytes per entry) - the bandwidth-bound SpMV BASELINE.json's roofline target speaks of - on a handle of its own;
    12 launches, the library's hipEvents.  The coded kernel moves a fifth of those bytes, so ITS fraction of the HBM peak is lower while
    it is 2.5x faster: both are reported, each on the bytes it actually moves."""
    h = _capi.Handle(device)
    try:
        h.set_options(_capi.FLAG_FUSED_NORM | _capi.FLAG_PROFILE)
        h.set_tuning(_capi.TUNE_FIXED_LAYOUT, 1)
        h.set_tuning(_capi.TUNE_PROFILE_STRIDE, 1)
        M = local.shape[0]
        h.set_csr(M, 0, local.rowptr, local.colidx, local.vals)
        h.basis_alloc(2)
        h.basis_set_row(0, np.random.default_rng(0).uniform(-1, 1, M))
        h.step_spmv(0)
        h.timings()
        for _ in range(12):
            h.step_spmv(0)
        t = h.timings()["spmv"]
        us = 1e3 * t["ms"] / max(t["timed_launches"], 1)
        gbs = t["timed_bytes"] / max(t["ms"], 1e-9) / 1e6
        return {"kernel": h.spmv_plan() + " (uncoded, knob 17 = 1)", "avg_us": round(us, 2), "bytes_per_launch": t["timed_bytes"] / max(t["timed_launches"], 1),
                "achieved_GBps": round(gbs, 1), "frac_hbm_peak": round(gbs / HBM_PEAK_GBS, 4), "launches": int(t["timed_launches"]),
                "coded_kernel_of_this_run": {"coding": coding[0], "classes": coding[1]}}
    finally:
        h.close()


def class_surface(lanczos_amd, local, k):
    """Wall time of the reference's call sequence through lanczos_amd.Lanczos (see the call site)."""
    H = local.to_scipy()
    M = H.shape[0]
    cls = lanczos_amd.Lanczos
    keep = cls.verbose
    cls.verbose = False
    try:
        s = cls(H)
        rec = {"workload_rows": M, "k": k}
        for name in ("first_call", "second_call"):
            t = time.perf_counter()
            s.execute_Lanczos(k)
            wall = time.perf_counter() - t
            dev = s.timings["total_ms"] / 1e3
            rec[name] = {"wall_s": round(wall, 4), "lz_run_device_s": round(dev, 4), "overhead_s": round(wall - dev, 4),
                         **{kk: round(vv, 4) for kk, vv in s.host_timings.items()}}
        t = time.perf_counter()
        theta = s.H_eigvals
        rec["H_eigvals_s"] = round(time.perf_counter() - t, 4)  # eigh(H_eff) + Y = V S + the two get_H_eigs checks on the device Gram matrix
        big = 16.0 * M * k > 48e9  # V and H_eigvecs together on the host
        if not big:
            t = time.perf_counter()
            V = s.V
            rec["V_fetch_s"] = round(time.perf_counter() - t, 4)
            rec["V_fetch_gbs"] = round(V.nbytes / max(time.perf_counter() - t, 1e-9) / 1e9, 1)
            del V
            s._V = None
            t = time.perf_counter()
            Y = s.H_eigvecs
            rec["H_eigvecs_fetch_s"] = round(time.perf_counter() - t, 4)
            rec["H_eigvecs_fetch_gbs"] = round(Y.nbytes / max(time.perf_counter() - t, 1e-9) / 1e9, 1)
            del Y
        else:
            rec["V_fetch_s"] = rec["H_eigvecs_fetch_s"] = None
        rec["ritz_min_max"] = [float(theta.min()), float(theta.max())]
        rec["note"] = ("lanczos_amd.Lanczos(H) on a fresh object, default start vector (np.random.seed(99); uniform(-1, 1, M) on the host, "
                       "as the reference's CPU branch does); outside `value`")
        s.close()
        return rec
    finally:
        cls.verbose = keep


def cpu_baseline(kind, dims, k, budget_s=40.0):
    """Time the oracle's FAITHFUL restatement of the reference loop (all n rows swept, two
    n x M temporaries) on this host for a bounded number of iterations."""
    import psutil
    import scipy

    from lanczos_amd import synthetic
    from oracle import lanczos_ref as oracle

    M = workload_rows(kind, dims)
    avail = psutil.virtual_memory().available
    n_cpu = k
    while 3.3 * 8 * n_cpu * M > 0.6 * avail and n_cpu > 8:
        n_cpu //= 2
    H = build_local(kind, dims, 0, M)
    H = H.to_scipy() if hasattr(H, "to_scipy") else oracle.as_operator(H)  # dense: CSR, as the reference's GPU branch converts it
    v0 = oracle.start_vector(M, 99)
    V = np.zeros((n_cpu, M))
    V[0] = v0
    r = H * V[0]
    a0 = np.dot(r, V[0])
    r = r - a0 * V[0]
    times = []
    j = 0
    t_all = time.perf_counter()
    while j < n_cpu and (j < 1 or time.perf_counter() - t_all < budget_s) and j < 4:
        t0 = time.perf_counter()
        b = np.linalg.norm(r)
        V[j] = r / b
        oracle.reorthogonalize(V, j)          # faithful: sweeps all n_cpu rows
        r = H * V[j]
        a = np.dot(V[j], r)
        r = r - V[j] * a - V[j - 1] * b
        times.append(time.perf_counter() - t0)
        j += 1
    t_iter = float(np.mean(times))
    threads = 1
    try:
        from threadpoolctl import threadpool_info

        blas = ";".join(f"{i.get('internal_api')}:{i.get('num_threads')}" for i in threadpool_info())
        threads = max([int(i.get("num_threads") or 1) for i in threadpool_info()] + [1])
    except Exception:
        blas = "unknown"
    out = {
        "value": (1.0 / t_iter) * (n_cpu / k),  # reference cost per iteration is proportional to n (independent of j)
        "unit": "iterations/s",
        "cores": threads,  # threads the NumPy loop can actually use: the BLAS pool (np.dot / norm); everything else is one core
        "host_cpus": os.cpu_count(),
        "kind": "port",
        "sample": f"{len(times)} iteration(s) of the faithful NumPy loop at M={M}, n={n_cpu} "
        f"({'full' if n_cpu == k else 'reduced to fit host RAM; value scaled by n/k'}); {t_iter:.2f} s/iteration; "
        f"only np.dot/norm are multi-threaded ({blas}); numpy {np.__version__}, scipy {scipy.__version__}",
        "s_per_iteration": t_iter,
        "n": n_cpu,
    }
    return out


def ritz_clock(info):
    """The S-stationary kernel's own clock record (lz_ritz_info): what clock the part held under FP64 MFMA + HBM traffic and
    how close the kernel runs to the MFMA issue floor IN CYCLES - the wall-time fraction is the product of the two."""
    if not info:
        return {}
    out = {"chunked_rows": info["chunk_rows"]}
    if info["tiles"] > 0:
        out.update({"shader_clock_mhz": round(info["clock_mhz"], 1), "cycles_per_16row_tile": round(info["cycles_per_tile"], 1),
                    "mfma_issue_floor_cycles_per_tile": round(info["mfma_floor_cycles_per_tile"], 1),
                    "mfma_issue_utilisation_in_cycles": round(info["mfma_floor_cycles_per_tile"] / info["cycles_per_tile"], 4),
                    "peak_tflops_at_held_clock": round(FP64_MFMA_PEAK_TFLOPS * info["clock_mhz"] / 2400.0, 2)})
    return out


def spawn_ranks(n):
    """Launcher for `python bench.py --gpus N` without torch.distributed.run: N child processes (RANK / LOCAL_RANK /
    WORLD_SIZE / LZ_RDZV_KEY exported), rank 0's stdout (the JSON line) relayed, exit status = the first failing child's.
    The parent never touches the GPU, and nothing that has is ever re-exec'ed.  All children are polled: when one exits
    non-zero the others (which would sit in accept() or in a collective) are terminated - exactly the processes started here."""
    import subprocess
    import threading
    import uuid

    key = uuid.uuid4().hex[:16]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LZ_RDZV_KEY=key,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    out0 = []
    drain = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)  # rank 0 must never block on a full pipe
    drain.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = (r, p.returncode)
                break
        time.sleep(0.2)
    if failed is None:
        bad = [(r, p.returncode) for r, p in enumerate(procs) if p.returncode != 0]
        failed = bad[0] if bad else None
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    drain.join(timeout=30)
    sys.stdout.write(b"".join(x for x in out0 if x).decode())
    sys.stdout.flush()
    if failed is not None:
        print(f"bench.py: rank {failed[0]} exited with status {failed[1]}; the other ranks were terminated", file=sys.stderr)
        return abs(failed[1]) or 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="lap2d_5pt_M1e7_k200", choices=sorted(WORKLOADS))
    ap.add_argument("--k", type=int, default=0, help="override the number of Lanczos iterations")
    ap.add_argument("--options", type=int, default=0, help="extra lz_flags (A/B arms)")
    ap.add_argument("--tune", default="", help="lz_set_tuning knobs for A/B runs, e.g. 13=1,0=2560")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap-arm", action="store_true", help="N > 1: skip the extra halo-overlap measurement")
    ap.add_argument("--arm-timeout", type=float, default=240.0, help="seconds after which stalled extra arms are abandoned (main line still printed)")
    ap.add_argument("--no-class-surface", action="store_true", help="skip the drop-in class-surface wall-time record (N = 1 only)")
    ap.add_argument("--no-partial", action="store_true", help="skip the extra (untimed-in-value) partial re-orthogonalisation measurement")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel hipEvents (roofline fields become null)")
    ap.add_argument("--profile-stride", type=int, default=8, help="bracket only every n-th iteration with hipEvents (each event costs ~3 us)")
    ap.add_argument("--prewarm-s", type=float, default=0.6, help="seconds of untimed short solves before the warm-up steps")
    ap.add_argument("--no-prewarm", action="store_true", help="skip the untimed runtime pre-warm (used under rocprofv3 --pmc)")
    ap.add_argument("--device", type=int, default=-1, help="GPU index (default: LOCAL_RANK); lets several ranks share one GPU with --backend host")
    ap.add_argument("--overlap", action="store_true", help="N > 1: exchange the halo on a second stream behind the interior re-orthogonalisation update (LZ_FLAG_OVERLAP_HALO)")
    ap.add_argument("--one-reduce", action="store_true", help="N > 1: LZ_FLAG_ONE_REDUCE - one all-reduce per iteration (alpha, |r|^2 and the coefficients in one buffer)")
    ap.add_argument("--bootstrap", default="socket", choices=["socket", "torch"])
    ap.add_argument("--backend", default="rccl", choices=["rccl", "host"])
    ap.add_argument("--mode", default="auto", choices=["auto", "halo", "allgather"])
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Plain `python bench.py --gpus N`: this process becomes the launcher.  It has made no HIP call (and makes
        # none): N fresh children, one rank per GPU, rendezvous over the torch-free socket bootstrap.
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world

    if world > 1 and args.bootstrap == "torch":
        # torch brings its own ROCm runtime (torch/lib): it must be in the process BEFORE liblanczos_hip.so, which then
        # binds to that one runtime and takes RCCL from the same tree.  The opposite order maps two HIP runtimes into
        # the process (round 1: abort in free() at exit); TorchBootstrap refuses it.
        import torch  # noqa: F401
    import lanczos_amd
    from lanczos_amd import _capi, distributed, partition

    if world > 1:
        # Default: rendezvous, barriers and the max-over-ranks go over a pure-Python Unix-socket bootstrap keyed by
        # LZ_RDZV_KEY / the launcher's MASTER_PORT; no torch in the process at all.
        boot = distributed.TorchBootstrap() if args.bootstrap == "torch" else distributed.SocketBootstrap()
    else:
        boot = distributed.Bootstrap()
    lanczos_amd.load_library()

    if args.device >= 0:
        local_rank = args.device
    kind, dims, k = WORKLOADS[args.workload]
    if args.k:
        k = args.k
    M = workload_rows(kind, dims)
    bounds = partition.row_bounds(M, world)
    lo, hi = bounds[rank], bounds[rank + 1]

    t0 = time.perf_counter()
    local = build_local(kind, dims, lo, hi)
    t_build = time.perf_counter() - t0

    prof = 0 if args.no_profile else _capi.FLAG_PROFILE
    if args.overlap:
        prof |= _capi.FLAG_OVERLAP_HALO
    comm_used = "none" if world == 1 else args.backend
    tuning = {int(kv.split("=")[0]): int(kv.split("=")[1]) for kv in filter(None, args.tune.split(","))}
    try:
        solver = distributed.DistributedLanczos(local, M, boot, device_id=local_rank, backend=args.backend, mode=args.mode,
                                                options=args.options | prof, one_reduce=args.one_reduce, tuning=tuning)
        ok = True
    except _capi.LanczosHipError as e:
        if world == 1 or args.backend != "rccl":
            raise
        print(f"[rank {rank}] RCCL setup failed ({e}); falling back to host-staged collectives", file=sys.stderr)
        ok = False
    if world > 1 and not all(boot.allgather_obj(ok)):
        comm_used = "host-staged (RCCL init failed)"
        solver = distributed.DistributedLanczos(local, M, boot, device_id=local_rank, backend="host", mode=args.mode,
                                                options=args.options | prof, one_reduce=args.one_reduce, tuning=tuning)
    v0 = solver.start_vector(99)[lo:hi].copy()
    # Per-kernel events cost ~3 us each (1.8 % of the headline run when every launch is bracketed): sample every
    # stride-th iteration (centred, so the sampled launches have the same mean basis size as all launches).
    stride = max(1, args.profile_stride)
    solver.h.set_tuning(_capi.TUNE_PROFILE_STRIDE, stride)

    # Setup (not a step): let the runtime finish its one-time work (code-object load of every kernel variant, clock
    # ramp) on a short solve; a ~60 ms one-off stall was observed ~0.1 s after the first launches of a process.
    t_pre = time.perf_counter()
    while not args.no_prewarm:
        solver.execute_Lanczos(min(k, 12), v0_normalized_local=v0)
        # every rank must run the same number of solves (they contain collectives): agree on when to stop
        if all(boot.allgather_obj(time.perf_counter() - t_pre >= args.prewarm_s)):
            break

    for _ in range(args.warmup):
        solver.execute_Lanczos(k, v0_normalized_local=v0)
    solver.timings()  # reset accumulators

    boot.barrier()
    solver.h.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        alpha, beta = solver.execute_Lanczos(k, v0_normalized_local=v0)
    solver.h.synchronize()
    boot.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        elapsed = max(boot.allgather_obj(elapsed))
    tm = solver.timings()

    # sanity of the result (cheap): Ritz values finite, extreme one inside the Gershgorin bound
    try:
        theta = solver.get_H_eigs(fetch=False)
        tr_first = solver.timings()["ritz"]  # first call of the process: includes the kernel's code-object load
        theta = solver.get_H_eigs(fetch=False)
        tr = solver.timings()["ritz"]
        tr["ms_first_call"] = tr_first["ms"]
        tr["info"] = solver.h.ritz_info()  # chunked? + the back-transform kernel's own clock record of THIS call
        if tr["info"]["chunk_rows"] > 0:
            # No room for a second M x k array beside the basis (C4 on one GPU): lz_ritz_vectors only kept S.  What get_H_eigs
            # then runs is the Gram matrix accumulated chunk by chunk (every chunk: Y rows = V^T-layout x S, then Y^T Y) - time that.
            G = solver.h.ritz_gram()
            assert np.abs(G - np.eye(k)).max() < 1e-10
            tr = dict(solver.timings()["ritz"], ms_first_call=0.0, info=solver.h.ritz_info(), chunked_pass="back-transform of every row chunk + its Gram accumulation")
        # the n x n Gram matrix of the Ritz vectors (the two checks of get_H_eigs, Lanczos.py:157-158), resident Y only: second call
        tg = None
        if tr["info"]["chunk_rows"] == 0:
            G = solver.h.ritz_gram()
            solver.timings()
            G = solver.h.ritz_gram()
            tg = solver.timings()["ritz"]
            tg["info"] = solver.h.gram_info() if world == 1 else None
            tg["max_dev_from_identity"] = float(np.abs(G - np.eye(k)).max())
            assert tg["max_dev_from_identity"] < 1e-10 and np.array_equal(G, G.T)
    except _capi.LanczosHipError as e:  # e.g. no room for a second M x k array next to the basis
        tg = None
        print(f"[rank {rank}] Ritz back-transform skipped: {e}", file=sys.stderr)
        theta = np.linalg.eigvalsh(solver.H_eff)
        tr = {"ms": 0.0, "flops": 0.0, "ms_first_call": 0.0}
    assert np.isfinite(theta).all()
    assert not getattr(solver, "breakdown", False), "Lanczos breakdown in the timed runs: the coefficients are rounding noise"

    if rank == 0:
        iters = args.steps * k
        per_class = {}
        for name in ("spmv", "qtw", "update", "three_term"):
            c = tm[name]
            if c["timed_launches"] == 0 or c["ms"] <= 0:
                continue
            gbs = c["timed_bytes"] / (c["ms"] * 1e-3) / 1e9
            per_class[name] = {
                "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                "traffic": None, "avg_us": round(1e3 * c["ms"] / c["timed_launches"], 2), "launches": c["launches"],
                "timed_launches": c["timed_launches"], "bytes_per_launch": c["timed_bytes"] / c["timed_launches"],
                "share_of_device_time": round(c["ms"] * c["launches"] / c["timed_launches"] / max(tm["total_ms"], 1e-9), 4),
            }
        dominant = max(per_class, key=lambda n: tm[n]["ms"]) if per_class else None
        whole_bytes = sum(tm[n]["bytes"] for n in ("spmv", "qtw", "update", "three_term"))
        traffic_file = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.isfile(traffic_file):  # HBM bytes per launch from rocprofv3 --pmc passes (see profiles/README.md)
            try:
                traffic_all = json.load(open(traffic_file))
                measured = traffic_all.get(args.workload, {})
                stamp = traffic_all.get(args.workload + "_measured", "taken in round 2, kernels unchanged since")
                for name, v in measured.items():
                    if name in per_class:
                        per_class[name]["traffic"] = v
                        # NOT a counter of this run: HBM bytes per launch from separate `rocprofv3 --pmc` passes of the same
                        # command (FETCH_SIZE / WRITE_SIZE, corrected as the guide prescribes), kept in profiles/
                        per_class[name]["traffic_source"] = f"profiles/hbm_traffic.json (rocprofv3 --pmc passes of this workload; static copy; {stamp})"
            except Exception:
                pass
        line = {
            "metric": "Lanczos iterations/sec + SpMV GB/s vs HBM roofline, n=1e7 5-pt Laplacian k=200",  # BASELINE.json
            "value": round(iters / elapsed, 3),
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": args.workload, "matrix": kind, "dims": list(dims), "M": M, "k": k, "reorth": "full (reference CGS, 2 passes)",
                       "partition": f"row-block x{world}", "exchange": solver.plan.mode, "comm": comm_used,
                       "fused_norm_allreduce": bool(solver.options & _capi.FLAG_FUSED_NORM), "profile_stride": stride,
                       "halo_overlap": bool(solver.options & _capi.FLAG_OVERLAP_HALO),
                       "one_reduce": bool(solver.options & _capi.FLAG_ONE_REDUCE), "spmv_kernel": solver.h.spmv_plan(),
                       # row-class coding of a stencil matrix (round 5): [kind, classes]; the SpMV then streams one byte per row (+ the values
                       # for "offsets") instead of 12 bytes per entry - roofline_all.spmv counts the bytes of the format that ran
                       "spmv_coding": list(solver.h.spmv_coding()) if hasattr(solver.h, "spmv_coding") else None,
                       "step": "one full k-iteration Lanczos solve"},
            "roofline": dict(per_class[dominant], kernel=dominant) if dominant else None,
            "roofline_all": per_class,
            "spmv_gbps": per_class.get("spmv", {}).get("achieved"),
            "spmv_frac_hbm_peak": per_class.get("spmv", {}).get("frac"),
            "whole_iteration_gbps": round(whole_bytes / (tm["total_ms"] * 1e-3) / 1e9, 1) if tm["total_ms"] > 0 else None,
            "device_ms_per_step": round(tm["total_ms"] / args.steps, 3),
            "comm_ms_per_step": round(tm["comm"]["ms"] * tm["comm"]["launches"] / max(tm["comm"]["timed_launches"], 1) / args.steps, 3),
            "comm_avg_us_per_call": round(1e3 * tm["comm"]["ms"] / max(tm["comm"]["timed_launches"], 1), 2),
            "comm_calls_per_step": tm["comm"]["launches"] // max(args.steps, 1),
            # by kind and per Lanczos ITERATION (a bench step is a whole k-iteration solve): all-reduces + SpMV-input exchanges
            "comm_calls_per_iteration": {"allreduce": round(tm["allreduces"] / max(args.steps, 1) / k, 3),
                                         "exchange": round(tm["exchanges"] / max(args.steps, 1) / k, 3)},
            "final_ms_per_step": round(tm["final"]["ms"] * tm["final"]["launches"] / max(tm["final"]["timed_launches"], 1) / args.steps, 3),
            "setup_s": {"matrix_build": round(t_build, 2)},
            "device": solver.h.device_name(),
            "ritz_min_max": [float(theta.min()), float(theta.max())],
            "partial_reorth": None,
            "ritz_backtransform": {"ms": round(tr["ms"], 3), "ms_first_call": round(tr.get("ms_first_call", 0.0), 3), "tflops": round(tr["flops"] / max(tr["ms"], 1e-9) / 1e9, 2),
                                   "bound": "mfma", "peak_tflops": FP64_MFMA_PEAK_TFLOPS,
                                   "frac": round(tr["flops"] / max(tr["ms"], 1e-9) / 1e9 / FP64_MFMA_PEAK_TFLOPS, 4),
                                   "note": "Y = V^T-layout x S (M x k x k) FP64 MFMA GEMM, outside the timed steps; second call of the process (ms_first_call: the first)",
                                   **({"chunked_pass": tr["chunked_pass"]} if "chunked_pass" in tr else {}), **ritz_clock(tr.get("info"))},
            "ritz_gram": gram_record(tg, k),
        }
    else:
        line = None

    # ---- extra arms, reported separately and NEVER part of `value`.  They contain collectives; a watchdog prints the
    # main line and ends every rank if an arm stalls, so the headline result cannot be lost to an experiment.
    import threading

    arm_state = {"arm": "none"}

    def bail():  # pragma: no cover
        # An arm (they contain collectives) made no progress for --arm-timeout seconds: that is a hang to be
        # root-caused, not a success.  Rank 0 still delivers the already-measured main line, marked; every rank says
        # on stderr which arm it was in, and the process ends with a non-zero status (never restarted, never re-exec'ed).
        print(f"bench.py: rank {rank} stalled in extra arm '{arm_state['arm']}' for {args.arm_timeout:.0f} s", file=sys.stderr, flush=True)
        if rank == 0:
            line["stalled"] = {"arm": arm_state["arm"], "timeout_s": args.arm_timeout, "note": "main result measured before the stall"}
            print(json.dumps(line), flush=True)
        os._exit(3)

    watchdog = threading.Timer(args.arm_timeout, bail)
    watchdog.daemon = True
    watchdog.start()
    alpha_main, beta_main = alpha.copy(), beta.copy()
    H_eff_main = np.diag(alpha_main) + np.diag(beta_main, 1) + np.diag(beta_main, -1)

    partial = overlap_arm = one_reduce_arm = None
    try:
        # (1) the opt-in partial re-orthogonalisation mode (the north star's "selective" arm); the headline reproduces the
        # reference's full sweep at every step.
        partial = None
        if not args.no_partial:
            arm_state["arm"] = "partial_reorth"
            theta_full, S_full = np.linalg.eigh(solver.H_eff)
            # which Ritz values the full-sweep run has CONVERGED (residual estimate beta |s_ni| <= 1e-9 of the spectral scale): where the
            # Krylov space starts to exhaust itself (C3) the late coefficients - and with them the unconverged Ritz values - are
            # ill-conditioned functions of the rounding errors in ANY implementation, so the two modes are
            # compared on the converged ones as well as on all of them
            conv_full = np.abs(S_full[-1, :]) * abs(float(np.diag(solver.H_eff, 1)[-1])) <= 1e-9 * np.abs(theta_full).max()
            solver.h.set_options(solver.options | _capi.FLAG_REORTH_PARTIAL)
            solver.execute_Lanczos(k, v0_normalized_local=v0)
            solver.timings()
            boot.barrier()
            solver.h.synchronize()
            tp = time.perf_counter()
            for _ in range(2):
                solver.execute_Lanczos(k, v0_normalized_local=v0)
            solver.h.synchronize()
            boot.barrier()
            tp = time.perf_counter() - tp
            if world > 1:
                tp = max(boot.allgather_obj(tp))
            tmp = solver.timings()
            theta_part = np.linalg.eigvalsh(solver.H_eff)
            # semi-orthogonality of the basis this mode leaves (sqrt(eps) by design): device Gram matrix of its Ritz vectors
            # Y = V S, |Y^T Y - I|_max = |S^T (V^T V - I) S|_max
            solver.get_H_eigs(fetch=False)
            gdev = float(np.abs(solver.h.ritz_gram() - np.eye(k)).max())
            solver.timings()
            partial = {
                "basis_semi_orthogonality_max_dev": gdev,
                "iterations_per_s": round(2 * k / tp, 1), "ms_per_solve": round(1e3 * tp / 2, 3), "sweeps": solver.h.last_sweeps(), "of": k,
                "engine": solver.h.last_engine(), "host_syncs_inside_lz_run": solver.h.last_host_syncs(),
                "comm_calls_per_iteration": {"allreduce": round(tmp["allreduces"] / 2.0 / k, 3), "exchange": round(tmp["exchanges"] / 2.0 / k, 3)},
                "lookahead_misses": solver.h.last_sweep_misses(),
                "device_ms_per_solve": round(tmp["total_ms"] / 2, 3),
                "max_rel_ritz_diff_vs_full": float(np.abs(theta_part - theta_full).max() / np.abs(theta_full).max()),
                "converged_ritz_values": int(conv_full.sum()),
                "max_rel_diff_of_converged_ritz_values_vs_full": (float(np.abs(theta_part[None, :] - theta_full[conv_full][:, None]).min(axis=1).max()
                                                                        / np.abs(theta_full).max()) if conv_full.any() else None),
                "spmv_share_of_device_time": round(tmp["spmv"]["ms"] * tmp["spmv"]["launches"] / max(tmp["spmv"]["timed_launches"], 1) / max(tmp["total_ms"], 1e-9), 3),
                "whole_iteration_gbps": round(sum(tmp[c]["bytes"] for c in ("spmv", "qtw", "update", "three_term")) / max(tmp["total_ms"], 1e-9) / 1e6, 1),
                "whole_iteration_frac_hbm_peak": round(sum(tmp[c]["bytes"] for c in ("spmv", "qtw", "update", "three_term")) / max(tmp["total_ms"], 1e-9) / 1e6 / HBM_PEAK_GBS, 4),
                "note": "opt-in LZ_FLAG_REORTH_PARTIAL (Simon 1984): the reference's sweep kernels run only when semi-orthogonality "
                        "is about to be lost; basis orthogonal to sqrt(eps), Ritz values to O(eps||A||)",
            }
            solver.h.set_options(solver.options)


        # (2) N > 1, stencil halos over RCCL: the same solve with LZ_FLAG_OVERLAP_HALO (faces of V[j] updated first and
        # exchanged on a second stream behind the interior update).  Off by default until measured on a multi-GPU node -
        # this arm is that measurement.
        overlap_arm = None
        if world > 1 and not args.no_overlap_arm and not args.overlap and solver.plan.mode == "halo" and comm_used == "rccl":
            arm_state["arm"] = "halo_overlap"
            solver.h.set_options(solver.options | _capi.FLAG_OVERLAP_HALO)
            solver.execute_Lanczos(k, v0_normalized_local=v0)
            boot.barrier()
            solver.h.synchronize()
            to = time.perf_counter()
            for _ in range(2):
                a_o, b_o = solver.execute_Lanczos(k, v0_normalized_local=v0)
            solver.h.synchronize()
            boot.barrier()
            to = max(boot.allgather_obj(time.perf_counter() - to))
            solver.timings()
            overlap_arm = {"iterations_per_s": round(2 * k / to, 1), "ms_per_solve": round(1e3 * to / 2, 3),
                           "max_abs_coeff_diff_vs_default": float(max(np.abs(a_o - alpha_main).max(), np.abs(b_o - beta_main).max()))}
            solver.h.set_options(solver.options)
        # (3) N > 1: LZ_FLAG_ONE_REDUCE - alpha, ||r||^2 and the coefficients in ONE all-reduce per iteration (2 collectives per
        # step instead of 3; pass 1 dots the basis against two columns).  Opt-in until measured on a multi-GPU node - this arm
        # is that measurement.
        one_reduce_arm = None
        if world > 1 and not args.no_overlap_arm and not args.one_reduce:
            arm_state["arm"] = "one_reduce"
            solver.h.set_options(solver.options | _capi.FLAG_ONE_REDUCE)
            solver.execute_Lanczos(k, v0_normalized_local=v0)
            solver.timings()
            boot.barrier()
            solver.h.synchronize()
            t1r = time.perf_counter()
            for _ in range(2):
                a_1, b_1 = solver.execute_Lanczos(k, v0_normalized_local=v0)
            solver.h.synchronize()
            boot.barrier()
            t1r = max(boot.allgather_obj(time.perf_counter() - t1r))
            tm1 = solver.timings()
            one_reduce_arm = {"iterations_per_s": round(2 * k / t1r, 1), "ms_per_solve": round(1e3 * t1r / 2, 3),
                              "comm_calls_per_iteration": round(tm1["comm"]["launches"] / 2.0 / k, 2),
                              "max_abs_coeff_diff_vs_default": float(max(np.abs(a_1 - alpha_main).max(), np.abs(b_1 - beta_main).max()))}
            # ... and the partial (selective) loop with one all-reduce per step (LZ_FLAG_REORTH_PARTIAL | LZ_FLAG_ONE_REDUCE, round 5)
            if not args.no_partial:
                arm_state["arm"] = "partial_one_reduce"
                solver.h.set_options(solver.options | _capi.FLAG_ONE_REDUCE | _capi.FLAG_REORTH_PARTIAL)
                solver.execute_Lanczos(k, v0_normalized_local=v0)
                solver.timings()
                boot.barrier()
                solver.h.synchronize()
                t1p = time.perf_counter()
                for _ in range(2):
                    a_p, b_p = solver.execute_Lanczos(k, v0_normalized_local=v0)
                solver.h.synchronize()
                boot.barrier()
                t1p = max(boot.allgather_obj(time.perf_counter() - t1p))
                tmq = solver.timings()
                one_reduce_arm["partial"] = {
                    "iterations_per_s": round(2 * k / t1p, 1), "ms_per_solve": round(1e3 * t1p / 2, 3), "engine": solver.h.last_engine(),
                    "sweeps": solver.h.last_sweeps(), "lookahead_misses": solver.h.last_sweep_misses(),
                    "comm_calls_per_iteration": {"allreduce": round(tmq["allreduces"] / 2.0 / k, 3), "exchange": round(tmq["exchanges"] / 2.0 / k, 3)},
                    "max_rel_ritz_diff_vs_full": float(np.abs(np.linalg.eigvalsh(solver.H_eff) - np.linalg.eigvalsh(H_eff_main)).max()
                                                       / np.abs(np.linalg.eigvalsh(H_eff_main)).max())}
            solver.h.set_options(solver.options)
        # (4) N = 1: what the DROP-IN caller waits for.  `value` times lz_run; a user of the reference calls
        # Lanczos(H).execute_Lanczos(k) and reads .H_eigvals / .V / .H_eigvecs (Lanczos.py:75-163).  Wall seconds of each through
        # the class surface, on a fresh object: first call (start vector + content hash + pack + validate + H2D + layout + solve),
        # second call on the unchanged H (hash only + solve), then the lazily fetched results.
        if (world == 1 and rank == 0 and not args.no_class_surface and hasattr(local, "rowptr") and hasattr(solver.h, "spmv_coding")
                and solver.h.spmv_coding()[0] != "none"):  # (not in the PMC / rocprofv3 passes, which run with --no-class-surface)
            arm_state["arm"] = "spmv_uncoded_arm"
            line["spmv_uncoded_arm"] = spmv_format_arm(_capi, local, local_rank, solver.h.spmv_coding())
            coded_us = line["roofline_all"].get("spmv", {}).get("avg_us")
            if coded_us:
                # the headline metric's "SpMV GB/s vs HBM roofline": the coded kernel moves a fraction of the CSR bytes, so its GB/s and
                # fraction (spmv_gbps, spmv_frac_hbm_peak: on the bytes it moves) are LOWER than the uncoded kernel's while it is faster
                line["spmv_note"] = ("row-class coded SpMV (%s, %d classes): %.1f us per launch on %.0f MB; the uncoded CSR kernel of the same matrix in "
                                     "this run: %.1f us on %.0f MB = %.3f of the HBM peak (spmv_uncoded_arm) - %.2fx slower"
                                     % (solver.h.spmv_coding()[0], solver.h.spmv_coding()[1], coded_us, line["roofline_all"]["spmv"]["bytes_per_launch"] / 1e6,
                                        line["spmv_uncoded_arm"]["avg_us"], line["spmv_uncoded_arm"]["bytes_per_launch"] / 1e6,
                                        line["spmv_uncoded_arm"]["frac_hbm_peak"], line["spmv_uncoded_arm"]["avg_us"] / coded_us))
        if world == 1 and not args.no_class_surface and hasattr(local, "to_scipy"):
            arm_state["arm"] = "class_surface"
            if 16.0 * M * k > 120e9:
                # (C4: the class-surface object allocates its own 160 GB basis - release this solver's first; it is not used again)
                solver.h.close()
            line_cs = class_surface(lanczos_amd, local, k)
            if rank == 0:
                line["class_surface"] = line_cs
    except Exception as e:  # an extra arm failed (e.g. a collective returned an error on this rank): the measured main line is
        # still delivered, marked, and the job ends non-zero at once - the other ranks are inside collectives this rank has left
        print(f"bench.py: rank {rank} failed in extra arm '{arm_state['arm']}': {e}", file=sys.stderr, flush=True)
        if rank == 0:
            line["arm_error"] = {"arm": arm_state["arm"], "error": str(e), "note": "main result measured before the failure"}
        if world > 1:
            if rank == 0:
                print(json.dumps(line), flush=True)
            os._exit(4)
        try:
            solver.h.set_options(solver.options)
        except Exception:
            pass  # (the handle was released for the class-surface arm)
    watchdog.cancel()
    arm_state["arm"] = "none"

    if rank == 0:
        line["partial_reorth"] = partial
        line["halo_overlap_arm"] = overlap_arm
        line["one_reduce_arm"] = one_reduce_arm
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(kind, dims, k)
            except MemoryError as e:  # pragma: no cover
                line["cpu_baseline"] = {"value": None, "unit": "iterations/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(line), flush=True)
    if world > 1:
        boot.barrier()


if __name__ == "__main__":
    main()
