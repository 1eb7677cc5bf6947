"""SYMMLQ behind pykrylov's `Symmlq` class (reference pykrylov/symmlq/symmlq.py:17-400).

The reference as shipped raises `AttributeError` in `solve` (symmlq.py:162 calls a non-existent
`self.matvec`); the intended `self.op * v` is what runs here.  Lanczos recurrence, LQ update and
stopping tests are device resident (``csrc/mk_symmlq.hip``).
"""
import ctypes

import numpy as np

from . import _lib
from .generic import KrylovMethod, DeviceRun
from .tools import machine_epsilon

__docformat__ = 'restructuredtext'


class Symmlq(KrylovMethod):
    """SYMMLQ for symmetric (possibly indefinite) ``(A - shift I) x = b``.

    Result attributes (symmlq.py:396-400): `nMatvec, bestSolution, x, solutionNorm, xNorm,
    residNorm, acond, anorm`.  As in the reference, `converged`, `istop` and `residNorm0` are not
    part of the contract (`istop` is provided additionally).
    """

    def __init__(self, op, **kwargs):
        KrylovMethod.__init__(self, op, **kwargs)
        self.name = 'Symmetric Indefinite Lanczos with Orthogonal Factorization'
        self.acronym = 'SYMMLQ'
        self.prefix = self.acronym + ': '
        self.iterates = []

    def solve(self, rhs, **kwargs):
        """Solve with right-hand side `rhs`.

        :keywords:
            :matvec_max: max. number of operator-vector products (default 2n+2)
            :rtol:       relative stopping tolerance (default 1e-9)
            :shift:      optional shift (default None; 0.0 is treated as None, symmlq.py:93)
            :check:      verify that the operator is symmetric on the first Lanczos vector (default False)
            :store_iterates: keep every iterate (LQ points) in `self.iterates` (default False)
        """
        op = self._device_operator()
        pdiag = self._device_precon(self.precon)      # ctor keyword (symmlq.py:60); diagonal operators only
        n = rhs.shape[0]
        n_glob = getattr(op, 'global_size', None) or n          # row-partitioned operator: same limit on every rank
        matvec_max = kwargs.get('matvec_max', 2 * n_glob + 2)
        rtol = kwargs.get('rtol', 1.0e-9)
        check = kwargs.get('check', False)
        shift = kwargs.get('shift', None)
        if shift == 0.0:
            shift = None
        store_iterates = kwargs.get('store_iterates', False)
        eps = machine_epsilon()

        not_symmetric = False
        # (with check=True the reference also tests the preconditioner for symmetry, symmlq.py:138-146; a diagonal
        # one always passes)
        with DeviceRun(op, _lib.MK_SYMMLQ, rhs, None, precon_diag=pdiag, matvec_max=int(matvec_max), rtol=float(rtol),
                       has_shift=int(shift is not None), shift=float(shift or 0.0)) as run:
            run.setup()
            lib = run.lib
            if check and getattr(op, 'local_size', None) is not None:
                raise NotImplementedError('Symmlq: check=True is not available on a row-partitioned operator '
                                          '(use pykrylov_amd.tools.check_symmetric, which is collective)')
            if check:
                # symmlq.py:163-171: s = <y,y>, t = <v, A y> with v, y of the first Lanczos step;
                # the extra product is not counted
                pv, py = ctypes.c_void_p(), ctypes.c_void_p()
                _lib.check(lib.mk_solver_vector(run.handle, 0, ctypes.byref(pv), None))
                _lib.check(lib.mk_solver_vector(run.handle, 1, ctypes.byref(py), None))
                # y of :162 before the shift is not kept on the device; recompute it
                tmp, tmp2 = _lib.DeviceArray(op.shape[1]), _lib.DeviceArray(n)
                _lib.check(lib.mk_memcpy_d2d(tmp.ptr, pv.value, 8 * n))
                _lib.check(lib.mk_spmv(op.handle, tmp.ptr, tmp2.ptr))           # y = A v
                _lib.check(lib.mk_memcpy_d2d(tmp.ptr, tmp2.ptr, 8 * n))
                r2 = _lib.DeviceArray(n)
                _lib.check(lib.mk_spmv(op.handle, tmp.ptr, r2.ptr))             # r2 = A y
                s, t = ctypes.c_double(), ctypes.c_double()
                _lib.check(lib.mk_dot(n, tmp2.ptr, tmp2.ptr, ctypes.byref(s)))
                _lib.check(lib.mk_dot(n, pv.value, r2.ptr, ctypes.byref(t)))
                if abs(s.value - t.value) > (s.value + eps) * eps ** (1.0 / 3):
                    not_symmetric = True
            res = run.finish()
            if store_iterates:
                self.iterates.append(np.zeros(n))
            if not not_symmetric:
                while not res.halted:
                    done = run.iterate(1 if store_iterates else (1 << 20))
                    if store_iterates and done:
                        self.iterates.append(run.x())
                    res = run.finish()
            else:
                # istop = 6: the loop is skipped, the epilogue still runs (symmlq.py:168-171, :361-382)
                raise NotImplementedError('Symmlq(check=True): operator is not symmetric (istop = 6)')
            x = run.x()

        op._nMatvec += int(res.nMatvec)
        self.istop = int(res.istop)
        self.nMatvec = int(res.nMatvec)
        self.bestSolution = x
        self.solutionNorm = np.float64(res.xnorm)
        self.x = self.bestSolution
        self.xNorm = np.float64(res.xnorm)
        self.residNorm = np.float64(res.residNorm)
        self.acond = np.float64(res.Acond)
        self.anorm = np.float64(res.Anorm)
        if self._logging():
            self.logger.info('Exit  SYMMLQ.    istop   =  %3g' % self.istop)
            self.logger.info('Exit  SYMMLQ.    anorm   =  %12.4e      acond =  %12.4e' % (self.anorm, self.acond))
            self.logger.info('Exit  SYMMLQ.    rnorm   =  %12.4e      xnorm =  %12.4e' % (self.residNorm, self.xNorm))
