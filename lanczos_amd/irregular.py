"""``IrrLanczos`` - drop-in for the symmetric half of the reference's
Python/Irregular/IrrLanczos.py (``execute_LanczosOld`` and friends)."""
import numpy as np

from ._solver import LanczosBase

_TWO_SIDED = (
    "the two-sided (bi-orthogonal) Lanczos of IrrLanczos.py:77-187/390-443 is outside this build's hot path "
    "(SURVEY.md section 2 #3: broken at the reference snapshot, no valid oracle); use execute_LanczosOld"
)


class IrrLanczos(LanczosBase):
    """Same solver as ``Lanczos``; differs like the reference copy does: ``get_H_eigs``
    runs no asserts (IrrLanczos.py:297-299), ``get_H_eigsOld`` only the norm check (:279-280)."""

    _check_eigs = ()

    def execute_LanczosOld(self, n, seed=99, use_cuda=True, v0=None):
        """IrrLanczos.py:193-260 (identical arithmetic to Regular on the CPU branch)."""
        self._execute(n, seed, use_cuda, v0)

    def execute_Lanczos(self, n, seed=99, use_cuda=True, v0=None, dtype=np.float64):
        if n > self.M:
            raise ValueError("n cannot be larger than M!")
        assert np.shape(self.H)[0] == np.shape(self.H)[1]
        raise NotImplementedError(_TWO_SIDED)

    def get_H_eigsOld(self):
        self._ritz(("normalized",))

    def print_good_eigs(self, tol=0.01, print_nr=20, print_bad=True, normal_eq=False):
        """IrrLanczos.py:331-353: as Regular's, Ritz values listed by increasing magnitude and
        optionally square-rooted (``normal_eq``: H was formed as the normal equations H^T H)."""
        eigvals, q = self.H_eigvals, self._eigvec_quality()
        if normal_eq:
            eigvals = np.sqrt(eigvals)
        order = np.argsort(np.abs(eigvals))
        print("__________EIGENVALUE AND EIGVENVECTOR COMPARISON__________")
        print("%12s %12s" % ("Eigval", "Eigvec InnerProd"))
        for i in range(print_nr):
            k = order[i]
            tag = "" if abs(1 - q[k]) < tol else " --- BAD"
            print("%12.4f%12.6f%s" % (eigvals[k], q[k], tag))

    def print_good_eigsOld(self, tol=0.01, print_nr=20, print_bad=True, normal_eq=False):
        """IrrLanczos.py:308-328: unsorted listing."""
        eigvals, q = self.H_eigvals, self._eigvec_quality()
        if normal_eq:
            eigvals = np.sqrt(eigvals)
        print("__________EIGENVALUE AND EIGVENVECTOR COMPARISON__________")
        print("%12s %12s" % ("Eigval", "Eigvec InnerProd"))
        for i in range(print_nr):
            tag = "" if abs(1 - q[i]) < tol else " --- BAD"
            print("%12.4f%12.6f%s" % (eigvals[i], q[i], tag))

    @staticmethod
    def bireorthogonalize(V1, V2, q_basis, p_basis, j, use_cuda=True, mem_safe=False):
        raise NotImplementedError(_TWO_SIDED)
