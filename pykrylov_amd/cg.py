"""Conjugate gradients behind pykrylov's `CG` class (reference pykrylov/cg/cg.py:10-165).

Same constructor, same `solve` keywords, same result attributes; the loop itself runs on
the GPU (``csrc/mk_cg.hip``: 3 fused kernels per iteration, no host round trip).
"""
import numpy as np

from . import _lib
from .generic import KrylovMethod, DeviceRun, DevicePrecon, HostPrecon
from .tools import check_symmetric

__docformat__ = 'restructuredtext'


class CG(KrylovMethod):
    """Conjugate gradient method for symmetric positive definite ``A x = b``.

    Per iteration: 1 operator-vector product, 2 dot products, 3 vector updates (cg.py:25-26),
    fused on the device into an SpMV+dot kernel and two streaming kernels.
    """

    def __init__(self, op, **kwargs):
        KrylovMethod.__init__(self, op, **kwargs)
        self.name = 'Conjugate Gradient'
        self.acronym = 'CG'
        self.prefix = self.acronym + ': '
        self.resids = []
        self.iterates = []
        self.infiniteDescent = None

    def solve(self, rhs, **kwargs):
        """Solve with right-hand side `rhs` (a Numpy array).

        :keywords:
            :guess:           initial guess (default 0)
            :matvec_max:      max. number of operator-vector products (default 2n)
            :check_symmetric: verify symmetry of the operator first (default False)
            :check_curvature: stop on non-positive curvature p'Ap (default True)
            :store_resids:    keep every residual vector in `self.resids` (default False)
            :store_iterates:  keep every iterate in `self.iterates` (default False)
        """
        op = self._device_operator()
        pdiag = self._device_precon(self.precon)
        n = getattr(op, 'global_size', None) or rhs.shape[0]    # (row-partitioned operator: the global size)
        store_resids = kwargs.get('store_resids', False)
        store_iterates = kwargs.get('store_iterates', False)

        if kwargs.get('check_symmetric', False):
            if not check_symmetric(op):
                self.logger.error('Coefficient operator is not symmetric')
                return                                             # cg.py:69-72

        guess = kwargs.get('guess', None)
        matvec_max = kwargs.get('matvec_max', 2 * n)

        def prec(r):                                                   # y = precon * r for `store_resids` (cg.py:96,133)
            if pdiag is None:
                return r
            if isinstance(pdiag, DevicePrecon):
                return pdiag.apply(r)
            if isinstance(pdiag, HostPrecon):                           # (same dispatch as the device loop's callback)
                p = pdiag.precon
                return p * r if hasattr(p, '__mul__') else p(r)
            return pdiag * r

        with DeviceRun(op, _lib.MK_CG, rhs, guess, precon_diag=pdiag, abstol=float(self.abstol),
                       reltol=float(self.reltol),
                       matvec_max=int(matvec_max),
                       check_curvature=int(bool(kwargs.get('check_curvature', True)))) as run:
            if store_resids or store_iterates:
                # vectors are wanted after every pass: step the device loop one pass at a time
                run.setup()
                res = run.finish()
                if store_iterates:
                    self.iterates.append(run.x())
                if store_resids:
                    self.resids.append(prec(run.vector(0)))
                while not res.halted:
                    run.iterate(1)
                    res = run.finish()
                    if res.definite:
                        if store_iterates:
                            self.iterates.append(run.x())
                        if store_resids:
                            self.resids.append(prec(run.vector(0)))
            else:
                res = run.run()
            x = run.x()
            hist = run.history()
            if not res.definite:
                self.infiniteDescent = run.vector(1)               # cg.py:122

        op._nMatvec += int(res.nMatvec)
        self.residNorm0 = np.float64(res.residNorm0)
        self.residHistory.extend(np.float64(h) for h in hist)     # accumulates across calls (generic.py:81)
        if self._logging():
            hdr = '%6s  %7s' % ('Matvec', 'Resid')
            self.logger.info(hdr)
            self.logger.info('-' * len(hdr))
            first = int(res.nMatvec) - (len(hist) - 1)
            for k, h in enumerate(hist):
                self.logger.info('%6d  %7.1e' % (first + k, h))
            if not res.definite:
                self.logger.error('Coefficient operator is not positive definite')
        self.converged = bool(res.converged)
        self.definite = bool(res.definite)
        self.nMatvec = int(res.nMatvec)
        self.bestSolution = self.x = x
        self.residNorm = np.float64(res.residNorm)
