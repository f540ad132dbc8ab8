#!/usr/bin/env python3
"""Teardown probe for the round-1 abort (`free(): invalid pointer` / `double free` at process exit after a 2-rank
torch-bootstrap bench).  One rank of a small 2-rank partitioned solve (host-staged collectives so that both ranks can
share the single GPU of a test box), with the import order and dlopen scope chosen on the command line; prints which
ROCm runtime copies are mapped into the process right before a NORMAL interpreter exit.

    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/teardown_probe.py --order torch_first
    ... --order lz_first --scope global     # the round-1 configuration (refused by TorchBootstrap unless --force)
"""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--order", default="torch_first", choices=["torch_first", "lz_first"])
ap.add_argument("--scope", default="local", choices=["local", "global"])
ap.add_argument("--rccl", action="store_true", help="also dlopen RCCL and create a unique id, as bench.py --backend rccl does")
args = ap.parse_args()

import lanczos_amd  # noqa: E402
from lanczos_amd import _capi, distributed, partition, synthetic  # noqa: E402


def load():
    if args.scope == "global":  # round 1 loaded the library with RTLD_GLOBAL
        _capi._lib = None
        lib = ctypes.CDLL(_capi.LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        for name, (res, a) in _capi.SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, a
        _capi._lib = lib
    else:
        lanczos_amd.load_library()


if args.order == "lz_first":
    load()
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    boot = distributed.TorchBootstrap.__new__(distributed.TorchBootstrap)  # bypass the dual-runtime refusal: we WANT that state
    import torch

    boot._torch, boot._dist, boot.rank, boot.world = torch, dist, dist.get_rank(), dist.get_world_size()
else:
    boot = distributed.TorchBootstrap()
    load()

M, n = 96 * 80, 40
b = partition.row_bounds(M, boot.world)
lo, hi = b[boot.rank], b[boot.rank + 1]
if args.rccl:
    try:
        _capi.preload_rccl()
        _capi.Handle(0).unique_id()
    except _capi.LanczosHipError as e:
        print(f"[rank {boot.rank}] rccl: {e}", file=sys.stderr)
s = distributed.DistributedLanczos(synthetic.laplacian_2d_5pt(96, 80, rows=(lo, hi)), M, boot, device_id=0, backend="host")
a, bt = s.execute_Lanczos(n)
theta = s.get_H_eigs()
print("PROBE" + json.dumps({"rank": boot.rank, "order": args.order, "scope": args.scope, "maps": _capi.mapped_runtimes(),
                            "bound": _capi.runtime_info(), "theta_max": float(theta.max())}), flush=True)
boot.barrier()
# normal exit from here on: atexit hooks, interpreter teardown, then the runtimes' own exit handlers
