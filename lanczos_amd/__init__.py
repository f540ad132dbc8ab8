"""lanczos_amd - MI355X (gfx950) native Lanczos eigensolver.

Drop-in for the ``Lanczos`` / ``IrrLanczos`` classes of jgslunde/Lanczos
(Python/Regular/Lanczos.py, Python/Irregular/IrrLanczos.py): same ``H``, ``n``,
``v0`` in, same ``H_eff``, ``V``, Ritz values and vectors out.  The Krylov loop
(SpMV, alpha/beta inner products, three-term recurrence, full
re-orthogonalisation) and the Ritz back-transform run as hand-written HIP
kernels in ``liblanczos_hip.so``, reached through ctypes - no PyTorch.
"""
from ._capi import (FLAG_FUSED_NORM, FLAG_REORTH_PARTIAL, FLAG_PROFILE, FLAG_QTW_MFMA, FLAG_QTW_VALU, FLAG_SPMV_SCALAR, LanczosHipError,
                    load_library)
from .hamiltonian import Hamiltonian
from .irregular import IrrLanczos
from .regular import Lanczos
from ._pool import StencilOperator

__all__ = ["Lanczos", "IrrLanczos", "Hamiltonian", "StencilOperator", "LanczosHipError", "load_library", "FLAG_PROFILE", "FLAG_QTW_MFMA", "FLAG_QTW_VALU",
           "FLAG_SPMV_SCALAR", "FLAG_FUSED_NORM", "FLAG_REORTH_PARTIAL"]
__version__ = "0.1.0"
