import ctypes

import numpy as np

from .. import _lib
from ..generic import KrylovMethod, DeviceRun, as_f64_vector

__docformat__ = 'restructuredtext'

_STATUS = {0: 'solution is zero', 1: 'residual small', 2: 'residual small', 4: 'residual small',
           5: 'residual small', 3: 'ill-conditioned operator', 6: 'ill-conditioned operator',
           7: 'max iterations', 8: 'direct error small'}


class _LlsBase(KrylovMethod):
    """Shared device glue of the four least-squares solvers."""

    kind = None

    def __init__(self, A, **kwargs):
        KrylovMethod.__init__(self, A, **kwargs)
        self.A = A
        self.x = None
        self.var = None
        self.itn = 0
        self.istop = 0
        self.optimal = False
        self.resids = []
        self.normal_eqns_resids = []
        self.dir_errors_window = []
        self.iterates = []
        self.msg = ['The exact solution is  x = 0                              ',
                    'Ax - b is small enough, given atol, btol                  ',
                    'The least-squares solution is good enough, given atol     ',
                    'The estimate of cond(Abar) has exceeded conlim            ',
                    'Ax - b is small enough for this machine                   ',
                    'The least-squares solution is good enough for this machine',
                    'Cond(Abar) seems to be too large for this machine         ',
                    'The iteration limit has been reached                      ',
                    'The truncated direct error is small enough, given etol    ']

    def _trnc_dir_err(self, res):
        """trncDirErr of the last pass for the `show` summaries: the window's ratio trnc / sqrt(xNrgNorm2) the device
        records (lsqr.py:310-318), times sqrt(xNrgNorm2); 0 while the window is not full (the reference's initial value)."""
        d = self.dir_errors_window
        return float(d[-1]) * float(np.sqrt(res.aux[3])) if len(d) and np.isfinite(d[-1]) else 0.0

    def _lls_diag(self, P, size, which):
        """M / N for the device loop: the fp64 diagonal of an operator that exposes one (DiagonalOperator,
        linop.py:473-516; applied inside the kernels), else the callable itself -- the reference applies them as
        functions, `u = M(Mu)` (lsqr.py:190) -- to be called back on the host at those sites."""
        if P is None:
            return None
        diag = getattr(P, 'diag', None)
        if diag is None or callable(diag):
            if not callable(P) and not hasattr(P, '__mul__'):
                raise TypeError('%s: %s must be callable (the reference evaluates `%s(vector)`); got a %s'
                                % (self.__class__.__name__, which, which, type(P).__name__))
            return P
        return as_f64_vector(diag, size, which + '.diag')

    @staticmethod
    def _host_thunk(P, size, errors):
        def call(user, ip, op_):
            try:
                vin = np.ctypeslib.as_array(ctypes.cast(ip, ctypes.POINTER(ctypes.c_double)), shape=(size,)).copy()
                out = np.asarray(P(vin) if callable(P) else P * vin)
                if out.shape != (size,):
                    raise ValueError('preconditioner returned shape %s, expected (%d,)' % (out.shape, size))
                np.ctypeslib.as_array(ctypes.cast(op_, ctypes.POINTER(ctypes.c_double)), shape=(size,))[:] = out
                return 0
            except BaseException as exc:                    # noqa: B902  (must not propagate through the C frames)
                errors.append(exc)
                return 1
        return _lib.PRECON_FN(call)

    def _run(self, rhs, itnlim, damp, atol, btol, conlim, M, N, kwargs, x_rows=False):
        A = self._device_operator()
        # M and N: the reference calls them as functions, `u = M(Mu)`, `v = N(Nv)` (lsqr.py:190,202); on the device
        # path they must be diagonal (an operator with a `.diag` array: DiagonalOperator, linop.py:473-516)
        dm = self._lls_diag(M, A.shape[0], 'M')
        dn = self._lls_diag(N, A.shape[1], 'N')
        if kwargs.get('wantvar', False):
            raise NotImplementedError('wantvar is broken in the reference as well (lsqr.py:155)')
        m, n = A.shape
        etol = kwargs.get('etol', 1.0e-6)
        window = kwargs.get('window', 5)
        store_iterates = kwargs.get('store_iterates', False)
        b = as_f64_vector(np.asarray(rhs).squeeze()[:m], m, 'rhs')
        At = A.T
        lib = _lib.init()
        self.iterates = []
        # DeviceRun sizes its rhs buffer from the operator's input size: build the run by hand for m x n
        d_rhs = _lib.DeviceArray.from_numpy(b)
        p = _lib.MkParams()
        p.struct_size = ctypes.sizeof(_lib.MkParams)
        p.kind = self.kind
        p.itnlim = int(itnlim)
        p.damp, p.atol, p.btol, p.conlim, p.etol = float(damp), float(atol), float(btol), float(conlim), float(etol)
        p.window = int(window)
        handle = ctypes.c_void_p()
        _lib.check(lib.mk_solver_create(A.handle, ctypes.byref(p), ctypes.byref(handle)))
        cb_errors = []
        cb_m = self._host_thunk(dm, m, cb_errors) if (dm is not None and not isinstance(dm, np.ndarray)) else None
        cb_n = self._host_thunk(dn, n, cb_errors) if (dn is not None and not isinstance(dn, np.ndarray)) else None
        d_dm = _lib.DeviceArray.from_numpy(dm) if isinstance(dm, np.ndarray) else None
        d_dn = _lib.DeviceArray.from_numpy(dn) if isinstance(dn, np.ndarray) else None
        try:
            _lib.check(lib.mk_solver_set_transpose(handle, At.handle))
            if cb_m is not None or cb_n is not None:
                none = ctypes.cast(None, _lib.PRECON_FN)
                _lib.check(lib.mk_solver_set_lls_precon_callback(handle, cb_m or none, None, cb_n or none, None))
            if d_dm is not None or d_dn is not None:
                _lib.check(lib.mk_solver_set_lls_precon(handle, None if d_dm is None else d_dm.ptr,
                                                        None if d_dn is None else d_dn.ptr))
            def chk(rc):
                if rc != 0 and hasattr(A, 'raise_pending'):
                    A.raise_pending()                         # what a matrix-free operator raised in its callback
                if rc != 0 and cb_errors:
                    raise cb_errors[0]                        # ... or M / N in theirs
                _lib.check(rc)
            chk(lib.mk_solver_setup(handle, d_rhs.ptr, None))
            res = _lib.MkResult()
            _lib.check(lib.mk_solver_finish(handle, ctypes.byref(res)))
            nx = m if x_rows else n

            def get_x():
                px = ctypes.c_void_p()
                _lib.check(lib.mk_solver_x(handle, ctypes.byref(px)))
                return _lib.download(px.value, nx)
            if store_iterates:
                self.iterates.append(get_x())
            table = kwargs.get('_table')                     # `show=True`: the reference's iteration log (see _ShowTable)

            def x_first():
                px = ctypes.c_void_p()
                _lib.check(lib.mk_solver_x(handle, ctypes.byref(px)))
                v0 = ctypes.c_double()
                _lib.check(lib.mk_memcpy_d2h(ctypes.byref(v0), px.value, 8))
                return v0.value
            if table is not None:
                table.start(res, x_first())
            while not res.halted:
                done = ctypes.c_int64()
                chk(lib.mk_solver_iterate(handle, 1 if (store_iterates or table is not None) else (1 << 20),
                                          ctypes.byref(done)))
                last_itn = int(res.itn)
                _lib.check(lib.mk_solver_finish(handle, ctypes.byref(res)))
                if store_iterates and int(res.itn) > last_itn:
                    self.iterates.append(get_x())
                if table is not None:
                    table.after_pass(res, x_first() if int(res.itn) > last_itn else None, int(res.itn) > last_itn)
            x = get_x()
            hist = np.empty(int(res.hist_len))
            derr = np.empty(int(res.hist_len))
            _lib.check(lib.mk_solver_history(handle, hist.ctypes.data, len(hist)))
            _lib.check(lib.mk_solver_history2(handle, derr.ctypes.data, len(derr)))
            r = None
            if self.kind == _lib.MK_CRAIG:
                pr = ctypes.c_void_p()
                _lib.check(lib.mk_solver_vector(handle, 0, ctypes.byref(pr), None))
                r = _lib.download(pr.value, m)
        finally:
            lib.mk_solver_destroy(handle)
            d_rhs.free()
            for buf in (d_dm, d_dn):
                if buf is not None:
                    buf.free()
        itn = int(res.itn)
        A._nMatvec += itn
        At._nMatvec += itn + (1 if res.residNorm0 > 0 else 0)
        self._hist = hist
        self.dir_errors_window = [np.float64(e) for e in derr[window:]]
        return res, x, r, itn


class _ShowTable(object):
    """The iteration log the reference prints with ``show=True`` (lsqr.py:222-234,394-411; lsmr.py:288-296,450-473), from
    the device loop stepped one pass at a time.  One quantity of a row is only settled by the NEXT pass's gate kernel
    (LSQR: cond(A), lsqr.py:338; LSMR: norm(x) for the print rule, lsmr.py:415), so row k is printed after pass k + 1 --
    or after the halting gate -- from the snapshot taken after pass k.  Same print rule, same formats; costs a device
    round trip per iteration, like every `show` mode: a debugging aid."""

    def __init__(self, kind, n, itnlim, atol, btol, ctol=0.0, pfreq=20):
        self.kind, self.n, self.itnlim, self.atol, self.btol, self.ctol = kind, n, itnlim, atol, btol, ctol
        self.pending = None
        self.pfreq, self.pcount = pfreq, 0
        self.lsqr = kind == _lib.MK_LSQR
        self.head = ('   Itn      x(1)       r1norm     r2norm ' + ' Compatible   LS      Norm A   Cond A') if self.lsqr else \
            ('   itn      x(1)       norm r    norm Ar ' + ' compatible   LS      norm A   cond A')

    @staticmethod
    def _snap(res, x0):
        return dict(itn=int(res.itn), x0=x0, r1=res.aux[0], r2=res.residNorm, normr=res.aux[1], normar=res.aux[2],
                    Anorm=res.Anorm, Acond=res.Acond, Arnorm=res.Arnorm, xnorm=res.xnorm, bnorm=res.residNorm0,
                    istop=int(res.istop))

    def start(self, res, x0):
        s = self._snap(res, x0)
        if res.halted:                                       # x = 0 is the solution: the reference prints msg[0] and returns
            return
        print(' ')
        print(self.head)
        beta = s['bnorm']
        if self.lsqr:                                        # lsqr.py:229-234 (test2 = alpha / beta; Arnorm = alpha beta)
            print('%6g %12.5e' % (0, x0) + ' %10.3e %10.3e' % (s['r1'], s['r2']) + '  %8.1e %8.1e' % (1.0, s['Arnorm'] / beta / beta))
        else:                                                # lsmr.py:291-296 (normA = alpha at this point)
            print(''.join(['%6g %12.5e' % (0, x0), ' %10.3e %10.3e' % (s['normr'], s['normar']),
                           '  %8.1e %8.1e' % (1, s['Anorm'] / beta)]))

    def _row(self, s, late):
        itn, bnorm = s['itn'], s['bnorm']
        if self.lsqr:
            Acond = late['Acond']                            # settled by the gate that followed
            test1 = s['r2'] / bnorm
            test2 = np.inf if (s['Anorm'] == 0. or s['r2'] == 0.) else s['Arnorm'] / (s['Anorm'] * s['r2'])
            test3 = np.inf if Acond == 0.0 else 1.0 / Acond
            rtol = self.btol + self.atol * s['Anorm'] * s['xnorm'] / bnorm
            prnt = (self.n <= 40 or itn <= 10 or itn >= self.itnlim - 10 or itn % 10 == 0 or test3 <= 2 * 0.0
                    or test2 <= 10 * self.atol or test1 <= 10 * rtol or late['istop'] != 0)     # (ctol stays 0: lsqr.py:161-163)
            if prnt:
                print('%6g %12.5e' % (itn, s['x0']) + ' %10.3e %10.3e' % (s['r1'], s['r2']) + '  %8.1e %8.1e' % (test1, test2)
                      + ' %8.1e %8.1e' % (s['Anorm'], Acond))
        else:
            normx = late['xnorm']
            test1 = s['normr'] / bnorm
            test2 = s['normar'] / (s['Anorm'] * s['normr'])
            test3 = 1 / s['Acond']
            rtol = self.btol + self.atol * s['Anorm'] * normx / bnorm
            prnt = (self.n <= 40 or itn <= 10 or itn >= self.itnlim - 10 or itn % 10 == 0 or test3 <= 1.1 * self.ctol
                    or test2 <= 1.1 * self.atol or test1 <= 1.1 * rtol or late['istop'] != 0)
            if prnt:
                if self.pcount >= self.pfreq:
                    self.pcount = 0
                    print(' ')
                    print(self.head)
                self.pcount += 1
                print(''.join(['%6g %12.5e' % (itn, s['x0']), ' %10.3e %10.3e' % (s['normr'], s['normar']),
                               '  %8.1e %8.1e' % (test1, test2), ' %8.1e %8.1e' % (s['Anorm'], s['Acond'])]))

    def after_pass(self, res, x0, advanced):
        now = self._snap(res, x0)
        if self.pending is not None and (advanced or res.halted):
            late = dict(now, istop=now['istop'] if res.halted and not advanced else 0)
            self._row(self.pending, late)
            self.pending = None
        if advanced:
            self.pending = now
            if res.halted:                                   # (a pass and its halting gate in one call)
                self._row(now, now)
                self.pending = None


class LSQRFramework(_LlsBase):
    """LSQR for ``A x = b`` / ``min |b - A x|`` / damped least squares (lsqr.py:26-453).

    Result attributes: `x, bestSolution, istop, itn, nMatvec (= 2 itn), r1norm, r2norm, residNorm, Anorm,
    Acond, Arnorm, xnorm, var (None), optimal, status`.
    """
    kind = _lib.MK_LSQR

    def solve(self, rhs, itnlim=0, damp=0.0, M=None, N=None, atol=1.0e-9, btol=1.0e-9, conlim=1.0e+8,
              show=False, wantvar=False, **kwargs):
        m, n = self.A.shape
        if itnlim == 0:
            itnlim = 3 * n
        kwargs['wantvar'] = wantvar
        if show:                                             # lsqr.py:168-175
            print(' ')
            print('LSQR            Least-squares solution of  Ax = b')
            print('The matrix A has %8d rows and %8d cols' % (m, n))
            print('damp = %20.14e     wantvar = %-5s' % (damp, repr(wantvar)))
            print('atol = %8.2e                 conlim = %8.2e' % (atol, conlim))
            print('btol = %8.2e                 itnlim = %8g' % (btol, itnlim))
            kwargs['_table'] = _ShowTable(self.kind, n, itnlim, atol, btol)
        res, x, _, itn = self._run(rhs, itnlim, damp, atol, btol, conlim, M, N, kwargs)
        if kwargs.get('store_resids', False):
            self.resids = [np.float64(res.residNorm0)] + [np.float64(h) for h in self._hist]
        istop = int(res.istop)
        if show and istop == 0:
            print(self.msg[0])                               # lsqr.py:215
        elif show:                                           # lsqr.py:418-434
            print(' ')
            print('LSQR finished')
            print(self.msg[istop])
            print(' ')
            print('istop =%8g   r1norm =%8.1e' % (istop, res.aux[0]) + '   ' + 'Anorm =%8.1e   Arnorm =%8.1e' % (res.Anorm, res.Arnorm))
            print('itn   =%8g   r2norm =%8.1e' % (itn, res.residNorm) + '   ' + 'Acond =%8.1e   xnorm  =%8.1e' % (res.Acond, res.xnorm))
            print('                  bnorm  =%8.1e' % res.residNorm0)
            print('xNrgNorm2 = %7.1e   trnDirErr = %7.1e' % (res.aux[3], self._trnc_dir_err(res)))
            print(' ')
        self.status = _STATUS[istop]
        self.optimal = istop in [1, 2, 4, 5, 8]
        self.x = self.bestSolution = x
        self.istop = istop
        self.itn = itn
        self.nMatvec = 2 * itn
        self.r1norm = np.float64(res.aux[0])
        self.r2norm = np.float64(res.residNorm)
        self.residNorm = self.r2norm
        self.Anorm = np.float64(res.Anorm)
        self.Acond = np.float64(res.Acond)
        self.Arnorm = np.float64(res.Arnorm)
        self.xnorm = np.float64(res.xnorm)
        self.var = None
        return


class LSMRFramework(_LlsBase):
    """LSMR (lsmr.py:28-492).  Like the reference, `solve` RETURNS
    ``(x, istop, itn, normr, normar, normA, condA, normx)`` and only sets `self.x`."""
    kind = _lib.MK_LSMR

    def solve(self, b, damp=0.0, atol=1e-9, btol=1e-9, conlim=1e8, M=None, N=None, itnlim=None, show=False,
              **kwargs):
        m, n = getattr(self.A, 'global_shape', self.A.shape)    # (row blocks on several GPUs: the same limit everywhere)
        if itnlim is None:
            itnlim = min([m, n])
        if show:                                             # lsmr.py:196-206
            print(' ')
            print('LSMR            Least-squares solution of  Ax = b')
            print('The matrix A has %8g rows  and %8g cols' % (m, n))
            print('damp = %20.14e' % (damp))
            print('atol = %8.2e                 conlim = %8.2e' % (atol, conlim))
            print('btol = %8.2e               itnlim = %8g' % (btol, itnlim))
            kwargs['_table'] = _ShowTable(self.kind, n, itnlim, atol, btol, ctol=(1.0 / conlim if conlim > 0 else 0.0))
        res, x, _, itn = self._run(b, itnlim, damp, atol, btol, conlim, M, N, kwargs)
        istop = int(res.istop)
        if kwargs.get('store_resids', False):
            self.resids = [np.float64(res.residNorm0)] + [np.float64(h) for h in self._hist]
        if show and istop == 0 and itn == 0:
            print(self.msg[0])                               # lsmr.py:286
        elif show:                                           # lsmr.py:479-489
            print(' ')
            print('LSMR finished')
            print(self.msg[istop])
            print('istop =%8g    normr =%8.1e' % (istop, res.aux[1]), '    normA =%8.1e    normAr =%8.1e' % (res.Anorm, res.aux[2]))
            print('itn   =%8g    condA =%8.1e' % (itn, res.Acond), '    normx =%8.1e' % (res.xnorm))
            print('Estimated energy norm of x: %7.1e' % np.sqrt(res.aux[3]))
        self.x = x
        return (x, istop, itn, np.float64(res.aux[1]), np.float64(res.aux[2]), np.float64(res.Anorm),
                np.float64(res.Acond), np.float64(res.xnorm))


class CRAIGFramework(_LlsBase):
    """CRAIG (craig.py:30-520).  Result attributes: `x, bestSolution, r, istop, itn, nMatvec, r1norm, r2norm,
    Arnorm, xnorm, optimal, status`."""
    kind = _lib.MK_CRAIG

    def solve(self, rhs, itnlim=0, damp=0.0, M=None, N=None, atol=1.0e-9, btol=1.0e-9, conlim=1.0e+8,
              show=False, wantvar=False, **kwargs):
        m, n = self.A.shape
        if itnlim == 0:
            itnlim = 3 * n
        kwargs['wantvar'] = wantvar
        if show:                                             # craig.py:193-200 (its iteration log is commented out there)
            print(' ')
            print('CRAIG           Least-squares solution of  Ax = b')
            print('The matrix A has %8d rows and %8d cols' % (m, n))
            print('damp = %20.14e     wantvar = %-5s' % (damp, repr(wantvar)))
            print('atol = %8.2e                 conlim = %8.2e' % (atol, conlim))
            print('btol = %8.2e                 itnlim = %8g' % (btol, itnlim))
        res, x, r, itn = self._run(rhs, itnlim, damp, atol, btol, conlim, M, N, kwargs)
        istop = int(res.istop)
        if show and istop == 0:
            print(self.msg[0])                               # craig.py:239
        elif show:                                           # craig.py:483-499
            print(' ')
            print('CRAIG finished')
            print(self.msg[istop])
            print(' ')
            print('istop =%8g   r1norm =%8.1e' % (istop, res.aux[0]))
            print('itn   =%8g   r2norm =%8.1e' % (itn, res.residNorm))
            print('                  bnorm  =%8.1e' % res.residNorm0)
            print('xNrgNorm2 = %7.1e   trnDirErr = %7.1e' % (res.aux[3], self._trnc_dir_err(res)))
            print(' ')
        self.dir_errors_d_window = self.dir_errors_window
        self.status = _STATUS[istop]
        self.optimal = istop in [1, 2, 4, 5, 8]
        self.x = self.bestSolution = x
        self.r = r
        self.istop = istop
        self.itn = itn
        self.nMatvec = 2 * itn
        self.r1norm = np.float64(res.aux[0])
        self.r2norm = np.float64(res.residNorm)
        self.Arnorm = np.float64(res.Arnorm)
        self.xnorm = np.float64(res.xnorm)
        return


class CRAIGMRFramework(_LlsBase):
    """CRAIG-MR (craigmr.py:12-241; its per-iteration debugging print at :190 is dropped).  `x` has
    ``nargout`` entries as in the reference (craigmr.py:112)."""
    kind = _lib.MK_CRAIGMR

    def solve(self, b, damp=0.0, atol=1e-9, btol=1e-9, conlim=1e8, M=None, N=None, itnlim=None, show=False,
              **kwargs):
        m, n = getattr(self.A, 'global_shape', self.A.shape)
        if itnlim is None:
            itnlim = min([m, n])
        res, x, _, itn = self._run(b, itnlim, damp, atol, btol, conlim, M, N, kwargs, x_rows=True)
        istop = int(res.istop)
        if show:
            print(' ')
            print('CRAIG-MR finished')
            print(self.msg[istop])
        self.status = _STATUS[istop]
        self.optimal = istop in [1, 2, 4, 5, 8]
        self.x = self.bestSolution = x
        self.istop = istop
        self.itn = itn
        self.nMatvec = 2 * itn
        return
