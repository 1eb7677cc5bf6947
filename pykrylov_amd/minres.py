"""MINRES behind pykrylov's `Minres` class (reference pykrylov/minres/minres.py:23-410).

Same `solve` keywords (note: they are read in `solve`, not in the constructor, minres.py:121-130)
and result attributes.  The Lanczos recurrence, the QR update and all stopping tests run on the GPU
(``csrc/mk_minres.hip``: 3 kernels per iteration; the reference's four vector copies per iteration
are pointer rotations).
"""
import numpy as np

from . import _lib
from .generic import KrylovMethod, DeviceRun
from .tools import check_symmetric, machine_epsilon

__docformat__ = 'restructuredtext'


class Minres(KrylovMethod):
    """MINRES for symmetric (possibly indefinite or singular) ``(A - shift I) x = b``.

    Result attributes (minres.py:395-408): `x, bestSolution, istop, itn, nMatvec (= itn), rnorm,
    residNorm, Arnorm, Anorm, Acond, ynorm, converged, residHistory, dir_errors_window`, and
    `status` when the direct-error test stopped the run (istop = 10).

    `istop`: 0 x = 0 is the solution; 1/2 converged to `rtol` (residual / least-squares sense);
    3/4 converged as far as `eps` / `Acond` allow; 6 iteration limit; 7 operator not symmetric;
    10 truncated direct error below `etol`; -1 b is an eigenvector (minres.py:87-98).
    """

    def __init__(self, op, **kwargs):
        KrylovMethod.__init__(self, op, **kwargs)
        self.name = 'Minimum Residual'
        self.acronym = 'MINRES'
        self.prefix = self.acronym + ': '
        self.residHistory = []
        self.resids = []
        self.dir_errors_window = []
        self.iterates = []
        self.eps = machine_epsilon()
        self.first = 'Enter minres.   '
        self.last = 'Exit  minres.   '
        self.msg = [' beta2 = 0.  If M = I, b and x are eigenvectors    ',
                    ' beta1 = 0.  The exact solution is  x = 0          ',
                    ' A solution to Ax = b was found, given rtol        ',
                    ' A least-squares solution was found, given rtol    ',
                    ' Reasonable accuracy achieved, given eps           ',
                    ' x has converged to an eigenvector                 ',
                    ' acond has exceeded 0.1/eps                        ',
                    ' The iteration limit was reached                   ',
                    ' Aname  does not define a symmetric matrix         ',
                    ' Mname  does not define a symmetric matrix         ',
                    ' Mname  does not define a pos-def preconditioner   ',
                    ' The truncated direct error is small enough        ']

    def normof2(self, x, y):
        return np.sqrt(x ** 2 + y ** 2)

    def solve(self, b, **kwargs):
        """Solve with right-hand side `b`.

        :keywords:
            :precon:  preconditioner M^-1 given as a diagonal operator (`precon.diag`); must be positive definite
            :shift:   solve (A - shift I) x = b (default 0)
            :show:    print the reference's iteration log -- header, one row per printed iteration (minres.py:362-379:
                      itn, x[0], test1, test2, Anorm, Acond, gbar / Anorm) and the exit summary (default True, as in
                      the reference).  The rows are printed pass by pass, which costs one device round trip per
                      iteration: a debugging aid, like the reference's.
            :check:   verify symmetry of A first (default True; 20 extra products on the operator)
            :itnlim:  iteration limit (default 5n)
            :rtol:    relative residual tolerance (default 1e-12)
            :etol:    truncated-direct-error tolerance (default 1e-6; pass 0.0 for an rtol-driven run)
            :window:  length of the direct-error window (default 5)
            :store_iterates: keep every iterate in `self.iterates` (default False)
        """
        A = self._device_operator()
        n = b.shape[0]
        n_glob = getattr(A, 'global_size', None) or n           # row-partitioned operator: same limit on every rank
        precon = kwargs.get('precon', None)                   # minres.py:121: a `solve` keyword, not a ctor one
        pdiag = self._device_precon(precon)
        shift = kwargs.get('shift', 0.0)
        show = kwargs.get('show', True)
        check = kwargs.get('check', True)
        itnlim = kwargs.get('itnlim', 5 * n_glob)
        rtol = kwargs.get('rtol', 1.0e-12)
        etol = kwargs.get('etol', 1.0e-6)
        store_iterates = kwargs.get('store_iterates', False)
        window = kwargs.get('window', 5)
        kwargs.get('store_resids', False)      # read and ignored, exactly like the reference (minres.py:128)

        self.residHistory = []                 # MINRES resets its histories on every solve (minres.py:132-133)
        self.dir_errors_window = []
        self.iterates = []

        if show:
            print(self.first + 'Solution of symmetric Ax = b')
            print('n      =  %3d     precon =  %4s           shift  =  %23.14e' % (n, (precon is not None), shift))
            print('itnlim =  %3d     rtol   =  %11.2e\n' % (itnlim, rtol))

        symmetric_ok = True
        # (the reference also runs check_symmetric on the preconditioner, minres.py:192-196; a diagonal one passes)
        with DeviceRun(A, _lib.MK_MINRES, b, None, precon_diag=pdiag, shift=float(shift), itnlim=int(itnlim), rtol=float(rtol),
                       etol=float(etol), window=int(window)) as run:
            run.setup()
            res = run.finish()
            if check and not check_symmetric(A):                              # minres.py:186-190
                symmetric_ok = False
            elif store_iterates or show:
                if store_iterates:
                    self.iterates.append(run.x())
                if show:                                                      # minres.py:210-214
                    print(' ' * 2)
                    print('   Itn     x[0]     Compatible    LS' + '       norm(A)  cond(A) gbar/|A|')
                eps = np.finfo(np.double).eps
                while not res.halted:
                    last_itn = int(res.itn)
                    run.iterate(1)
                    res = run.finish()
                    itn = int(res.itn)
                    if itn == last_itn:
                        continue
                    if store_iterates:
                        self.iterates.append(run.x())
                    if show:                                                  # minres.py:362-383, same rule, same formats
                        Anorm, Acond, ynorm, rnorm = res.Anorm, res.Acond, res.ynorm, res.residNorm
                        epsx, epsr = Anorm * ynorm * eps, Anorm * ynorm * rtol
                        test1, test2, gbar = rnorm / (Anorm * ynorm), res.aux[0], res.aux[1]
                        prnt = (n <= 40 or itn <= 10 or itn >= itnlim - 10 or itn % 10 == 0 or rnorm <= 10 * epsx
                                or rnorm <= 10 * epsr or Acond <= 1e-2 / eps or int(res.istop) != 0)
                        if prnt:
                            print('%6g %12.5e %10.3e' % (itn, run.x_first(), test1) + ' %10.3e' % test2
                                  + ' %8.1e %8.1e %8.1e' % (Anorm, Acond, gbar / Anorm))
                        if int(res.istop) <= 0 and itn % 10 == 0:
                            print(' ')
            else:
                while not res.halted:
                    run.iterate(1 << 20)
                    res = run.finish()
            x = run.x()
            hist = run.history()
            derr = np.empty(len(hist))
            _lib.check(run.lib.mk_solver_history2(run.handle, derr.ctypes.data, len(hist)))

        istop, itn = int(res.istop), int(res.itn)
        rnorm, Arnorm, Anorm, Acond, ynorm = res.residNorm, res.Arnorm, res.Anorm, res.Acond, res.ynorm
        if not symmetric_ok:
            istop, itn, rnorm, Arnorm, Anorm, Acond, ynorm = 7, 0, 0.0, 0.0, 0.0, 0.0, 0.0
            x = np.zeros(n)
            hist, derr = hist[:0], derr[:0]
        A._nMatvec += itn
        self.residNorm0 = np.float64(res.residNorm0)
        self.residHistory = [np.float64(h) for h in hist]
        self.dir_errors_window = [np.float64(e) for e in derr[window:]]       # defined once itn > window

        if show:
            last = self.last
            print(last + ' istop   =  %3g               itn   =%5g' % (istop, itn))
            print(last + ' Anorm   =  %12.4e      Acond =  %12.4e' % (Anorm, Acond))
            print(last + ' rnorm   =  %12.4e      ynorm =  %12.4e' % (rnorm, ynorm))
            print(last + ' Arnorm  =  %12.4e' % Arnorm)
            print(last + self.msg[istop + 1])

        self.converged = istop in [1, 2, 3, 4, 10]
        if istop == 10:
            self.status = 'direct error small'
        self.x = self.bestSolution = x
        self.istop = istop
        self.itn = self.nMatvec = itn
        self.rnorm = self.residNorm = np.float64(rnorm)
        self.Arnorm = np.float64(Arnorm)
        self.Anorm = np.float64(Anorm)
        self.Acond = np.float64(Acond)
        self.ynorm = np.float64(ynorm)
        return
