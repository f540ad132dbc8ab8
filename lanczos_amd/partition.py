"""Row-block partition of the Lanczos problem over ranks (host-side planning).

The reference is single-process; this is the build's
own design.  Rank ``p`` owns the contiguous rows
``[p*chunk, min(M, (p+1)*chunk))`` of H, of every Krylov vector and of r, with
``chunk`` a multiple of 32 so every rank's slice is 256-byte aligned in the
padded global numbering.  The SpMV input is completed in one of two ways:

* ``halo``      - each rank receives exactly the remote entries its rows touch
                  (stencils: the two neighbouring slabs' faces) into a ghost tail
                  stored right behind the owned part of the basis row;
* ``allgather`` - every rank gathers the whole padded vector (irregular graphs,
                  where almost every remote entry is touched anyway).

Pure NumPy; no device or communication calls, so it is unit-tested on CPU.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

PAD = 32  # doubles; must match lz_padded_rows() in the HIP library


def round_up(x, m):
    return (int(x) + m - 1) // m * m


def chunk_size(M, world):
    return round_up(-(-int(M) // int(world)), PAD)


def row_bounds(M, world):
    """Row offsets [b_0 .. b_world] of the uniform padded partition."""
    c = chunk_size(M, world)
    return [min(int(M), p * c) for p in range(world + 1)]


@dataclass
class ExchangePlan:
    mode: str                 # "none" | "halo" | "allgather"
    rows: int
    rows_pad: int
    ncols_ext: int            # length of the extended local vector the local colidx index
    colidx: np.ndarray        # int32 local column indices
    chunk: int = 0            # allgather: padded rows per rank
    peers: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    send_counts: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))
    send_idx: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))    # local row indices, peer-major
    recv_counts: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))
    ghost_cols: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))  # global ids of the ghost tail, in order


def stencil3d_columns(dims, points, rows):
    """Global column ids of the periodic 7-/27-point stencil rows ``rows`` of a ``Nx x Ny x Nz`` grid, ``(len(rows), points)``,
    unsorted (flat index ``x + Nx*y + Nx*Ny*z``, neighbours wrap: Hamiltonian.py:73-99)."""
    Nx, Ny, Nz = (int(d) for d in dims)
    r = np.asarray(rows, dtype=np.int64)
    x, y, z = r % Nx, (r // Nx) % Ny, r // (Nx * Ny)
    offs = [(dx, dy, dz) for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)
            if int(points) == 27 or abs(dx) + abs(dy) + abs(dz) <= 1]
    return np.stack([((x + dx) % Nx) + ((y + dy) % Ny) * Nx + ((z + dz) % Nz) * (Nx * Ny) for dx, dy, dz in offs], axis=1)


def plan_stencil_slab(dims, points, world, rank):
    """Halo plan of one rank's row block of the periodic 3-D stencil operator WITHOUT building the matrix: only rows within
    one stencil reach of either end of the block can touch another rank's rows (``2*Nx*Ny``: a z-neighbour combined with
    an in-plane periodic wrap moves a column by less than two planes), so the planner of ``plan_exchange`` runs on those
    boundary rows alone.  Returns ``(plan, ghost_ranges)``: ``plan.colidx`` is ``None`` (the
    device assembles the renumbered columns itself, lz_build_stencil3d_block) and ``ghost_ranges`` lists the ghost tail as
    contiguous ``(global_start, length)`` runs, in tail order."""
    Nx, Ny, Nz = (int(d) for d in dims)
    M = Nx * Ny * Nz
    chunk = chunk_size(M, world)
    lo, hi = min(M, rank * chunk), min(M, (rank + 1) * chunk)
    rows = hi - lo
    if world == 1:
        return ExchangePlan("none", rows, round_up(rows, PAD), M, None), []
    reach = 2 * Nx * Ny
    ids = np.arange(rows, dtype=np.int64)
    ids = ids if rows <= 2 * reach else np.concatenate([ids[:reach], ids[rows - reach:]])
    cols = stencil3d_columns(dims, points, lo + ids)
    k = cols.shape[1]
    plan = plan_exchange(np.arange(len(ids) + 1, dtype=np.int64) * k, cols.reshape(-1), M, world, rank, mode="halo", row_ids=ids)
    plan.colidx = None
    g = plan.ghost_cols
    if len(g) == 0:
        return plan, []
    cut = np.flatnonzero(np.diff(g) != 1) + 1
    starts = np.concatenate([[0], cut])
    ends = np.concatenate([cut, [len(g)]])
    ranges = [(int(g[a]), int(b - a)) for a, b in zip(starts, ends)]
    if len(ranges) > 16:
        raise ValueError("the ghost tail of this block is not a handful of contiguous runs (block boundaries cut through too many planes)")
    return plan, ranges


def plan_exchange(rowptr, colidx_global, M, world, rank, mode="auto", allgather_threshold=0.25, row_ids=None):
    """Plan the SpMV input exchange for this rank's row block.

    ``rowptr`` / ``colidx_global``: CSR of the owned rows with GLOBAL column ids.
    The send lists are derived from the owned rows alone, using the structural
    symmetry of a Hermitian H (row i has an entry in a column owned by q  <=>  q has a
    row with an entry in column i); ``check_plans`` verifies that across ranks.
    """
    rowptr = np.asarray(rowptr)
    cols = np.asarray(colidx_global, dtype=np.int64)
    chunk = chunk_size(M, world)
    lo, hi = min(M, rank * chunk), min(M, (rank + 1) * chunk)
    rows = hi - lo
    rows_pad = round_up(rows, PAD)
    # row_ids: the CSR holds only these local rows (a stencil's boundary rows, plan_stencil_slab); default: all owned rows
    row_ids = np.arange(rows, dtype=np.int64) if row_ids is None else np.asarray(row_ids, dtype=np.int64)
    assert len(rowptr) == len(row_ids) + 1
    if world == 1:
        return ExchangePlan("none", rows, rows_pad, int(M), cols.astype(np.int32))
    own = (cols >= lo) & (cols < hi)
    ghost_cols = np.unique(cols[~own])
    if mode == "auto":
        mode = "allgather" if len(ghost_cols) > allgather_threshold * max(rows, 1) else "halo"
    if mode == "allgather":
        # padded global numbering == global numbering because every rank starts at rank*chunk
        return ExchangePlan("allgather", rows, rows_pad, chunk * world, cols.astype(np.int32), chunk=chunk)
    if mode != "halo":
        raise ValueError(f"unknown exchange mode {mode!r}")
    local = np.where(own, cols - lo, 0)
    local[~own] = rows_pad + np.searchsorted(ghost_cols, cols[~own])
    owner = ghost_cols // chunk
    peers, recv_counts = np.unique(owner, return_counts=True)
    # rows of mine that touch a column owned by q -> q needs those x entries (structural symmetry)
    row_of = np.repeat(row_ids, np.diff(rowptr))
    pair = np.unique((cols[~own] // chunk) * rows_pad + row_of[~own])
    send_owner, send_row = pair // rows_pad, pair % rows_pad
    speers, send_counts = np.unique(send_owner, return_counts=True)
    all_peers = np.union1d(peers, speers)
    rc = np.zeros(len(all_peers), np.int64)
    sc = np.zeros(len(all_peers), np.int64)
    rc[np.searchsorted(all_peers, peers)] = recv_counts
    sc[np.searchsorted(all_peers, speers)] = send_counts
    return ExchangePlan(
        "halo", rows, rows_pad, rows_pad + len(ghost_cols), local.astype(np.int32), chunk=chunk,
        peers=all_peers.astype(np.int32), send_counts=sc, send_idx=send_row.astype(np.int32), recv_counts=rc, ghost_cols=ghost_cols,
    )


def check_plans(plan, rank, gathered):
    """``gathered[q]`` = (peers, send_counts, recv_counts) of rank q.  Raises if what this
    rank sends to q is not what q expects (H not structurally symmetric)."""
    if plan.mode != "halo":
        return
    for p, sc, rc in zip(plan.peers, plan.send_counts, plan.recv_counts):
        qp, qsc, qrc = gathered[int(p)]
        k = np.searchsorted(qp, rank)
        ok = k < len(qp) and qp[k] == rank and qrc[k] == sc and qsc[k] == rc
        if not ok:
            raise ValueError(f"halo plan mismatch between ranks {rank} and {int(p)}: H is not structurally symmetric")


def assemble_x_ext(plan, x_local, fetch_global):
    """NumPy model of what the device exchange produces (tests): the extended local vector."""
    if plan.mode == "none":
        return x_local
    if plan.mode == "allgather":
        return fetch_global(None)
    x = np.zeros(plan.ncols_ext)
    x[: plan.rows] = x_local
    x[plan.rows_pad :] = fetch_global(plan.ghost_cols)
    return x
