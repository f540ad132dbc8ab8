"""``IrrLanczos`` - drop-in for the reference's Python/Irregular/IrrLanczos.py: the symmetric solver
(``execute_LanczosOld`` and friends) and the two-sided (bi-orthogonal) variant ``execute_Lanczos`` /
``bireorthogonalize`` (IrrLanczos.py:77-187, 390-443), both executed by liblanczos_hip.so."""
import numpy as np
import scipy.sparse

from . import _capi
from ._solver import LanczosBase, _pack_matrix, _use_cuda_or_notice


class IrrLanczos(LanczosBase):
    """Same solver as ``Lanczos``; differs like the reference copy does: ``get_H_eigs``
    runs no asserts (IrrLanczos.py:297-299), ``get_H_eigsOld`` only the norm check (:279-280)."""

    _check_eigs = ()

    def execute_LanczosOld(self, n, seed=99, use_cuda=True, v0=None):
        """IrrLanczos.py:193-260 (identical arithmetic to Regular on the CPU branch)."""
        self._execute(n, seed, use_cuda, v0)

    def execute_Lanczos(self, n, seed=99, use_cuda=True, v0=None, dtype=np.float64):
        """Two-sided Lanczos (IrrLanczos.py:77-187): right/left Krylov bases ``q``/``p`` of ``H`` and ``H^T`` with
        ``q_i . p_j = +-delta_ij``, every new pair bi-orthogonalised against orthonormalised copies of the bases
        (``bireorthogonalize``).  Publishes the non-symmetric tridiagonal ``H_eff`` (sub-diagonal ``beta``,
        super-diagonal ``gamma``, laid out as :165-174 does) and ``V = q.T``.

        Reference behaviour kept: only ``v0=None`` works (the second start vector exists only on that branch, :98-102);
        the start pair comes from the global legacy RNG (two draws) and is scaled to ``q0 . p0 = +-1``; no breakdown
        check.  ``dtype=np.float32``: inputs rounded to float32 where the reference rounds them, recurrence in float64 (the
        device path's only precision), results published as float32, with a one-line notice."""
        if n > self.M:
            raise ValueError("n cannot be larger than M!")
        assert self.H.shape[0] == self.H.shape[1]
        self._say("+++ Executing Lanczos algorithm")
        self.n = n
        _use_cuda_or_notice(self, use_cuda)
        if self._multi():
            raise NotImplementedError("the two-sided variant runs on one GPU (devices must be None or a single entry)")
        dtype = np.dtype(dtype)
        if dtype not in (np.dtype(np.float64), np.dtype(np.float32)):
            raise NotImplementedError("dtype must be float64 or float32")
        if dtype != np.float64:
            # The reference would run its whole recurrence in that precision (:90-96, 114-125).  The device path has float64
            # kernels only: the INPUTS are rounded to `dtype` exactly where the reference rounds them (the matrix at :91-92, the
            # start pair at :115-119), the recurrence runs in float64, and alpha / beta / gamma / V are published in `dtype` -
            # results at least as accurate as a float32 recurrence, not bit-comparable with one (no caller in the reference
            # passes dtype).
            print("+++ dtype=%s: inputs rounded to %s, recurrence in float64 on the MI355X, results published as %s." % (dtype.name, dtype.name, dtype.name))
        M = self.M
        H = scipy.sparse.csr_matrix(scipy.sparse.csr_matrix(self.H, dtype=dtype), dtype=np.float64)  # :91
        HT = scipy.sparse.csr_matrix(H.transpose(), dtype=np.float64)                                # :92
        np.random.seed(seed)
        if v0 is None:
            v0 = np.random.uniform(-1, 1, size=(M))
            v1 = np.random.uniform(-1, 1, size=(M))
        else:
            v0 = np.array(v0)
            # :104 reads v1, which the reference assigns only when v0 is None
            raise UnboundLocalError("local variable 'v1' referenced before assignment")
        dot = np.sqrt(np.abs(np.dot(v0, v1)))
        v0 = v0 / dot
        v1 = v1 / dot * np.sign(np.dot(v0, v1))
        if n < 2:
            # with n == 1 the loop body never runs and :163 reads an unassigned residual
            raise UnboundLocalError("local variable 'r' referenced before assignment")
        if dtype != np.float64:
            v0, v1 = v0.astype(dtype).astype(np.float64), v1.astype(dtype).astype(np.float64)  # q[0] = v0, p[0] = v1 into dtype arrays (:115-119)

        kind, rowptr, colidx, vals = _pack_matrix(H)
        _, t_rowptr, t_colidx, t_vals = _pack_matrix(HT)
        h = self._get_handle()
        h.set_options(self.options)
        h.set_csr(M, 0, rowptr, colidx, vals)
        self._matrix_key = self._matrix_key_alt = None  # (the symmetric solver's cache key does not describe this upload)
        symmetric = (len(t_colidx) == len(colidx) and np.array_equal(t_rowptr, rowptr) and np.array_equal(t_colidx, colidx)
                     and np.array_equal(t_vals, vals))
        if symmetric:
            h.set_csr_transpose()  # H^T x runs on H: one matrix resident
        else:
            h.set_csr_transpose(t_rowptr, t_colidx, t_vals)
        alpha, beta, gamma = h.run_two_sided(n, v0, v1)
        self._timings = h.timings()
        self.sweeps = n - 1
        if dtype != np.float64:
            alpha, beta, gamma = alpha.astype(dtype), beta.astype(dtype), gamma.astype(dtype)

        # H_eff exactly as :165-174 lays it out (row i >= 1 carries gamma[i-1], not gamma[i], right of the diagonal)
        H_eff = np.zeros((n, n))
        H_eff[0, 0] = alpha[0]
        H_eff[0, 1] = gamma[0]
        H_eff[-1, -2] = beta[-1]
        H_eff[-1, -1] = alpha[-1]
        for i in range(1, n - 1):
            H_eff[i, i - 1] = beta[i - 1]
            H_eff[i, i] = alpha[i]
            H_eff[i, i + 1] = gamma[i - 1]
        self._alpha, self._beta, self._gamma = alpha, beta, gamma
        self._H_eff = H_eff
        self._V = None if dtype == np.float64 else h.get_basis().T.astype(dtype)  # (the reference's q is a dtype array, :114)
        self.H = H  # the reference's GPU branch leaves a SciPy CSR in self.H (:183)
        self.H_eigs_have_been_found = False
        self._say("+++ Lanczos executed successfully.")
        self.Lanczos_has_been_executed = True

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
        """In place on row ``j`` of the four (n, M) arrays, the reference's default branch (IrrLanczos.py:408-441):
        project ``V1[j]`` on ``p_basis[:j]`` and ``V2[j]`` on ``q_basis[:j]`` (sequential Gram-Schmidt), rescale the
        pair to ``V1[j] . V2[j] = +-1``, then extend the two orthonormal bases by row ``j``.  ``j = 0`` (nothing to project on)
        updates all four rows and then raises ``ValueError`` exactly as the reference does (its closing ``np.max`` runs over an
        empty array).  ``mem_safe=True`` is the
        reference's other branch (:398-407, no caller there): one sweep of ``V1[j]`` against all rows of ``V2`` and of
        ``V2[j]`` against all rows of ``V1``, each coefficient over that row's own squared norm; ``q_basis`` /
        ``p_basis`` are not read.  Both run on the device."""
        _use_cuda_or_notice(LanczosBase, use_cuda)
        if mem_safe:
            V1, V2 = np.asarray(V1), np.asarray(V2)
            n, M = V1.shape
            if not 0 <= j < n:
                raise ValueError("bireorthogonalize needs 0 <= j < n")
            h = _capi.Handle(LanczosBase.device_id)
            try:
                eye_ptr = np.arange(M + 1, dtype=np.int32)
                h.set_csr(M, 0, eye_ptr, eye_ptr[:-1], np.ones(M))  # length carrier only; no matvec is run
                h.bi_alloc(n)
                for which, a in enumerate((V1, V2)):
                    for i in range(n):  # the sweep reads every row, also those past j (IrrLanczos.py:399-400)
                        h.bi_set_row(which, i, a[i])
                h.step_bireorth_mem_safe(j)
                V1[j] = h.bi_get_row(0, j)
                V2[j] = h.bi_get_row(1, j)
            finally:
                h.close()
            return
        arrs = [np.asarray(a) for a in (V1, V2, q_basis, p_basis)]
        n, M = arrs[0].shape
        if not 0 <= j < n:
            raise IndexError("index %d is out of bounds for axis 0 with size %d" % (j, n))
        h = _capi.Handle(LanczosBase.device_id)
        try:
            eye_ptr = np.arange(M + 1, dtype=np.int32)
            h.set_csr(M, 0, eye_ptr, eye_ptr[:-1], np.ones(M))  # length carrier only; no matvec is run
            h.bi_alloc(n)
            for which, a in enumerate(arrs):
                for i in ([j] if which < 2 else range(j)):  # the step reads row j of the pair and rows < j of the bases
                    h.bi_set_row(which, i, a[i])
            h.step_bireorth(j)
            for which, a in enumerate(arrs):
                a[j] = h.bi_get_row(which, j)
        finally:
            h.close()
        if j == 0:
            # The reference's last statement (IrrLanczos.py:441) takes the maximum over the j earlier basis rows: with j = 0 that
            # is np.max of an empty array, which raises - AFTER all four rows have been updated in place (pinned by
            # tests/golden/bireorth_default_j0.npz, an output of the reference's own static method).  Same here.
            raise ValueError("zero-size array to reduction operation maximum which has no identity")
