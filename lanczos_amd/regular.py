"""``Lanczos`` - drop-in for the class in the reference's Python/Regular/Lanczos.py."""
from ._solver import LanczosBase


class Lanczos(LanczosBase):
    """Symmetric Lanczos with full re-orthogonalisation on one MI355X.

    Usage is the reference's (Python/Regular/Lanczos.py:11-18): construct with a
    Hermitian ``H`` (SciPy sparse or dense ndarray), call ``execute_Lanczos(n)``,
    then read ``H_eff``, ``V``, ``H_eigvals``, ``H_eigvecs`` (``get_H_eigs`` runs
    lazily).
    """

    def execute_Lanczos(self, n, seed=99, use_cuda=True, v0=None):
        """Lanczos.py:75-141.  ``use_cuda=True`` (default) runs the HIP path."""
        self._execute(n, seed, use_cuda, v0)
