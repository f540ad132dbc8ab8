"""One rank of a ``Lanczos.devices = [...]`` run: ``python -m lanczos_amd._worker`` (started by ``_pool.WorkerPool``).

A FRESH process per GPU (the calling process never has to touch a GPU, and nothing that has is ever re-exec'ed): it
reads JSON commands (one per line) on stdin, answers one JSON line per command on the descriptor it was handed as
stdout, and moves bulk data (matrix, start vector, basis, Ritz vectors) through files in /dev/shm that the parent
creates and unlinks.  The rank's row block lives in a ``distributed.DistributedLanczos`` - the same object ``bench.py``
drives - so the data path is liblanczos_hip.so + RCCL (or the host-staged test backend); this file is plumbing only.
"""
from __future__ import annotations

import json
import os
import sys
import traceback

import numpy as np


def _attach(spec, mode="r"):
    """``spec`` = (path, dtype string, shape) of a parent-owned /dev/shm file."""
    path, dt, shape = spec
    return np.memmap(path, dtype=np.dtype(dt), mode=mode, shape=tuple(int(x) for x in shape))


class Rank:
    def __init__(self):
        from . import _capi, distributed

        self.capi, self.dist = _capi, distributed
        self.rank = int(os.environ["RANK"])
        self.world = int(os.environ["WORLD_SIZE"])
        self.device = int(os.environ.get("LZ_DEVICE", "0"))
        self.backend = os.environ.get("LZ_BACKEND", "rccl")
        self.boot = distributed.SocketBootstrap(timeout=float(os.environ.get("LZ_RDZV_TIMEOUT", "600"))) if self.world > 1 else distributed.Bootstrap()
        _capi.load_library()
        self.solver = None

    # ---- commands -------------------------------------------------------------------------------------------------------
    def cmd_matrix(self, m):
        from . import partition, synthetic

        if self.solver is not None:
            self.solver.h.close()
            self.solver = None
        M = int(m["M"])
        b = partition.row_bounds(M, self.world)
        lo, hi = b[self.rank], b[self.rank + 1]
        common = dict(boot=self.boot, device_id=self.device, backend=self.backend, options=int(m.get("options", 0)),
                      fused_norm=bool(m.get("fused_norm", True)), one_reduce=bool(m.get("one_reduce", False)), tuning=m.get("tuning") or None)
        kind = m["kind"]
        if kind == "csr":
            rowptr, colidx, vals = _attach(m["rowptr"]), _attach(m["colidx"]), _attach(m["vals"])
            a, e = int(rowptr[lo]), int(rowptr[hi])
            local = synthetic.CSR((np.asarray(rowptr[lo:hi + 1], dtype=np.int64) - a).astype(np.int32), np.asarray(colidx[a:e]),
                                  np.asarray(vals[a:e]), (hi - lo, M))
            self.solver = self.dist.DistributedLanczos(local, M, mode=m.get("mode", "auto"), **common)
        elif kind == "dense":
            A = _attach(m["A"])
            self.solver = self.dist.DistributedLanczos(np.ascontiguousarray(A[lo:hi]), M, mode=m.get("mode", "auto"), **common)
        elif kind == "stencil":
            pot = _attach(m["potential"]) if m.get("potential") is not None else None
            self.solver = self.dist.DistributedLanczos.from_stencil(
                tuple(m["dims"]), int(m["points"]), T_factor=float(m["T_factor"]), weights4=tuple(m["weights4"]), negate_T=bool(m["negate_T"]),
                potential_params=m.get("potential_params"), potential_local=None if pot is None else np.asarray(pot[lo:hi]), **common)
        else:
            raise ValueError(f"unknown matrix kind {kind!r}")
        self.lo, self.hi, self.M = lo, hi, M
        return {"rows": hi - lo, "lo": lo, "exchange": self.solver.plan.mode, "spmv": self.solver.h.spmv_plan(), "device": self.solver.h.device_name()}

    def cmd_run(self, m):
        s = self.solver
        s.h.set_options(int(m["options"]))
        s.options = int(m["options"])
        v0 = _attach(m["v0"])
        alpha, beta = s.execute_Lanczos(int(m["n"]), v0_normalized_local=np.asarray(v0[self.lo:self.hi]))
        return {"alpha": alpha, "beta": beta, "breakdown": bool(s.h.breakdown), "sweeps": s.h.last_sweeps(), "engine": s.h.last_engine(),
                "timings": s.h.timings()}

    def cmd_residual(self, m):
        out = _attach(m["r"], "r+")  # (M,): this rank's rows of the residual entering step n
        out[self.lo:self.hi] = self.solver.h.get_residual()
        out.flush()
        return {}

    def cmd_resume(self, m):
        s = self.solver
        s.h.set_options(int(m["options"]))
        s.options = int(m["options"])
        V = _attach(m["V"])  # (j0, M)
        r = _attach(m["r"])
        alpha, beta = s.resume_Lanczos(int(m["n"]), np.ascontiguousarray(V[:, self.lo:self.hi]), np.asarray(r[self.lo:self.hi]), m["alpha"], m["beta"])
        return {"alpha": alpha, "beta": beta, "breakdown": bool(s.h.breakdown), "sweeps": s.h.last_sweeps(), "engine": s.h.last_engine(),
                "timings": s.h.timings()}

    def cmd_ritz(self, m):
        S = np.array(_attach(m["S"]))
        self.solver.h.ritz_vectors(S, fetch=False)
        return {}

    def cmd_gram(self, m):
        return {"G": self.solver.h.ritz_gram()}  # collective: summed over the ranks, identical everywhere

    def cmd_quality(self, m):
        return {"q": self.solver.h.ritz_quality()}  # collective

    def cmd_fetch_basis(self, m):
        out = _attach(m["V"], "r+")  # (n, M): this rank writes the columns [lo, hi) of every row
        h = self.solver.h
        ptr = out.ctypes.data + 8 * self.lo
        h.check(h.lib.lz_get_basis(h._h, self.capi.C.cast(ptr, self.capi._D), self.M))
        out.flush()
        return {}

    def cmd_fetch_basis_block(self, m):
        out = _attach(m["V"], "r+")  # (n, r1 - r0)
        r0, r1 = (int(x) for x in m["rows"])
        a, b = max(r0, self.lo), min(r1, self.hi)
        if b > a:
            out[:, a - r0:b - r0] = self.solver.h.get_basis_block(a - self.lo, b - self.lo)
            out.flush()
        return {}

    def cmd_fetch_ritz(self, m):
        out = _attach(m["Y"], "r+")  # (M, n) C-order: this rank's rows are one contiguous block
        h = self.solver.h
        r0, r1 = (int(x) for x in m.get("rows", (0, self.M)))
        a, b = max(r0, self.lo), min(r1, self.hi)
        if b > a:
            block = h.ritz_fetch_rows(a - self.lo, b - self.lo)
            out[a - r0:b - r0] = block
            out.flush()
        return {}

    def cmd_ping(self, m):
        return {"rank": self.rank, "runtime": self.capi.runtime_info()}

    def close(self):
        if self.solver is not None:
            self.solver.h.close()
            self.solver = None


def main():
    # the protocol owns the stdout descriptor; anything else that prints (progress lines, library chatter) goes to stderr
    proto = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    sys.stdout = sys.stderr
    from .distributed import _dec, _enc

    def reply(obj):
        proto.write(json.dumps(_enc(obj)) + "\n")
        proto.flush()

    try:
        rank = Rank()
    except BaseException as e:  # rendezvous / library load failed: tell the parent why before dying
        reply({"error": f"{type(e).__name__}: {e}", "traceback": traceback.format_exc()})
        raise
    reply({"ready": True, "pid": os.getpid()})
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        msg = _dec(json.loads(line))
        if msg.get("cmd") == "close":
            break
        try:
            out = getattr(rank, "cmd_" + msg["cmd"])(msg)
            reply(out)
        except BaseException as e:
            reply({"error": f"{type(e).__name__}: {e}", "traceback": traceback.format_exc()})
            if not isinstance(e, Exception):
                raise
    rank.close()
    reply({"closed": True})


if __name__ == "__main__":
    main()
