"""The reference's usage (Python/Regular/3Ddeuteron.py:74-100) on the GPUs of one node - one attribute more than on one GPU.

    python examples/multi_gpu_drop_in.py [N] [n] [gpu ids ...]      e.g.  python examples/multi_gpu_drop_in.py 64 100 0 1 2 3

With more than one GPU id the calling process never touches a GPU: it spawns one worker per id (lanczos_amd/_pool.py), each
assembles its own slab of H = -T + V on its device (the matrix never exists on the host) and the row-block-partitioned solver
runs over RCCL."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from a checkout
from lanczos_amd import Hamiltonian, Lanczos, synthetic

N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
gpus = [int(a) for a in sys.argv[3:]]
L = 25.0
T_factor = 197.327**2 / (2 * 469.4592) / (L / N) ** 2

Hamiltonian.vectorize_potential = True
ham = Hamiltonian(N, L, synthetic.deuteron_potential, T_factor)
if len(gpus) > 1:
    Lanczos.devices = gpus            # the one attribute
TEST = Lanczos(ham.operator("27"))    # or Lanczos(H) with H = ham.build_H("27") / the reference's -T + V
TEST.execute_Lanczos(n, use_cuda=False, seed=78)   # 3Ddeuteron.py:95, unedited
print("lowest Ritz values:", np.array2string(TEST.H_eigvals[:4], precision=6))
TEST.print_good_eigs(print_nr=5)
TEST.close()
