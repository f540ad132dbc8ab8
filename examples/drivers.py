#!/usr/bin/env python3
"""The three problems the reference ships as driver scripts, through the drop-in classes (no plotting):

    python examples/drivers.py box1d      [--N 500  --n 50]      a finite well in a 1-D box, dense matrix          (1Dbox.py)
    python examples/drivers.py deuteron1d [--N 1001 --n 1001]    radial-like 1-D deuteron, sparse, n = M           (1Ddeuteron.py)
    python examples/drivers.py deuteron3d [--N 160  --n 400]     27-point deuteron Hamiltonian, the largest run    (3Ddeuteron.py)
                                          [--devices 0,1,2,3,4,5,6,7]  the same call surface, basis split over the GPUs

Each builds the matrix the corresponding script builds (vectorised; the scripts use Python loops), runs
``Lanczos(H).execute_Lanczos(n)`` / ``get_H_eigs`` and prints the lowest Ritz values next to the quality figure
``print_good_eigs`` uses.  tests/test_gpu_drivers.py holds them to the golden fixtures the reference produced on
these very matrices.
"""
import argparse
import os
import sys
import time

import numpy as np
import scipy.sparse

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanczos_amd import Hamiltonian, Lanczos  # noqa: E402

HBARC = 197.327        # MeV fm
REST_ENERGY = 469.4592  # MeV: the reduced mass of the two nucleons


def deuteron_potential(x, y=0.0, z=0.0):
    """hard core + Gaussian-like well (the parameters of 1Ddeuteron.py:10-16 / 3Ddeuteron.py:51-61)"""
    r = np.sqrt(np.asarray(x) ** 2 + np.asarray(y) ** 2 + np.asarray(z) ** 2)
    e_well = 54.531
    return 40.0 * e_well * np.exp(-((r / 0.25) ** 4.0)) - 65.4823128982115 * np.exp(-((r / 1.7) ** 4.0))


def kinetic_factor(L, N):
    dx = float(L) / N
    return HBARC**2 / (2 * REST_ENERGY) / dx**2


def box1d_matrix(N=500):
    """1Dbox.py:5-22: -d2/dx2 with Dirichlet ends plus a well of depth 10 over the middle half; a dense ndarray"""
    pot = np.zeros(N)
    pot[N // 4 : (3 * N) // 4] = -10
    return np.diag(2 + pot) - np.diag(np.ones(N - 1), 1) - np.diag(np.ones(N - 1), -1)


def deuteron1d_matrix(N=1001, L=25):
    """1Ddeuteron.py:6-54: H = -T + V on N points of [0, L]; T's end rows are one-sided, V's last entry is left out"""
    Tf = kinetic_factor(L, N)
    V = deuteron_potential(np.linspace(0, L, N))
    V[N - 1] = 0.0
    T = scipy.sparse.diags([np.full(N - 1, Tf), np.r_[-Tf, np.full(N - 2, -2 * Tf), -Tf], np.full(N - 1, Tf)], [-1, 0, 1], format="csr")
    return -T + scipy.sparse.diags(V, format="csr")


def deuteron3d_system(N=160, L=25):
    return Hamiltonian(N, L, deuteron_potential, kinetic_factor(L, N))


def deuteron3d_matrix(N=160, L=25, points="27"):
    """3Ddeuteron.py:63-84: the periodic N^3 grid, 27-point Laplacian, H = -T + V as sorted CSR (assembled on the device)"""
    system = deuteron3d_system(N, L)
    system.create_sparse_T(points)
    system.create_sparse_V()
    H = -system.T_sparse + system.V_sparse
    H.sort_indices()
    return H


def solve(H, n, seed=99, use_cuda=True, devices=None, backend="rccl", verbose=True):
    Lanczos.verbose = verbose
    s = Lanczos(H)
    if devices:
        s.devices = list(devices)
        s.comm_backend = backend
    t0 = time.perf_counter()
    s.execute_Lanczos(n, seed=seed, use_cuda=use_cuda)
    s.get_H_eigs()
    s.wall_s = time.perf_counter() - t0
    return s


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("problem", choices=["box1d", "deuteron1d", "deuteron3d"])
    ap.add_argument("--N", type=int, default=0)
    ap.add_argument("--n", type=int, default=0)
    ap.add_argument("--points", default="27", choices=["7", "27"])
    ap.add_argument("--devices", default="", help="comma-separated GPU indices: the basis is row-block partitioned over them")
    ap.add_argument("--backend", default="rccl", choices=["rccl", "host"])
    ap.add_argument("--descriptor", action="store_true", help="deuteron3d: hand Lanczos the closed-form operator, assembled on the device(s); "
                                                                "the matrix never exists on the host")
    ap.add_argument("--print-nr", type=int, default=10)
    ap.add_argument("--save", default="", help="write eigvals / eigvecs as .npy with this prefix (3Ddeuteron.py:98-99)")
    a = ap.parse_args(argv)
    devices = [int(d) for d in a.devices.split(",")] if a.devices else None
    if a.problem == "box1d":
        N, n = a.N or 500, a.n or 50
        s = solve(box1d_matrix(N), n, devices=devices, backend=a.backend)
    elif a.problem == "deuteron1d":
        N = a.N or 1001
        s = solve(deuteron1d_matrix(N), a.n or N, devices=devices, backend=a.backend)
    else:
        N, n = a.N or 160, a.n or 400
        H = deuteron3d_system(N).operator(a.points) if a.descriptor else deuteron3d_matrix(N, points=a.points)
        s = solve(H, n, seed=78, use_cuda=False, devices=devices, backend=a.backend)  # 3Ddeuteron.py:95 passes use_cuda=False; it runs on the GPU here
    print("%d Lanczos steps on M = %d in %.2f s (matrix upload, solve, Ritz back-transform)" % (s.n, s.M, s.wall_s))
    s.print_good_eigs(print_nr=min(a.print_nr, s.n))
    if a.save:
        np.save(a.save + "eigvals.npy", s.H_eigvals)
        np.save(a.save + "eigvecs.npy", s.H_eigvecs)
    return s


if __name__ == "__main__":
    main()
