"""`KrylovMethod`: the solver protocol of pykrylov (reference pykrylov/generic/generic.py:11-98),
plus the glue every device-resident solver of this package shares.

A solver object keeps pykrylov's constructor keywords (`abstol`, `reltol`, `precon`,
`logger`; unknown keywords are ignored as in generic.py:74-77) and result attributes.
`solve()` hands the whole loop to `libmikrylov.so`.  With a :class:`pykrylov_amd.linop.CsrOperator` (matrix
resident in HBM) the products run on the device too; any other operator of the reference's protocol
(``op * vector``) is called back on the host at each product site while the rest of the loop stays on the device
(:class:`pykrylov_amd.linop.HostOperatorShell`).  There is no host implementation of the loops in this package.
"""
import ctypes
import logging
import os
import time

import numpy as np

from . import _lib

__docformat__ = 'restructuredtext'

null_log = logging.getLogger('krylov')
null_log.setLevel(logging.INFO)
null_log.addHandler(logging.NullHandler())


class KrylovMethod(object):
    """Template of all Krylov solvers (generic.py:11-98).

    :parameters:
        :op:  the operator; ``y = op * x`` is the product with the coefficient matrix.

    :keywords:
        :abstol:  absolute stopping tolerance (default 1.0e-8)
        :reltol:  relative stopping tolerance (default 1.0e-6)
        :precon:  optional preconditioner
        :logger:  a `logging.Logger` (default: a null logger)
    """

    def __init__(self, op, **kwargs):
        self.prefix = 'Generic: '
        self.name = 'Generic Krylov Method (must be subclassed)'
        self.op = op
        self.abstol = kwargs.get('abstol', 1.0e-8)
        self.reltol = kwargs.get('reltol', 1.0e-6)
        self.precon = kwargs.get('precon', None)
        self.logger = kwargs.get('logger', null_log)
        self.residNorm = None
        self.residNorm0 = None
        self.residHistory = []
        self.nMatvec = 0
        self.nIter = 0
        self.converged = False
        self.bestSolution = None
        self.x = self.bestSolution

    def _write(self, msg):
        self.logger.info(msg)

    def solve(self, rhs, **kwargs):
        raise NotImplementedError('This method must be subclassed')

    # ------------------------------------------------------------------ device glue
    def _device_operator(self):
        """The operator as the device loop sees it: a CsrOperator (matrix in HBM, products on the device) or, for
        any other operator of the reference's protocol, a shell whose products call back into it on the host
        (the loop itself -- dots, updates, recurrences, stopping tests -- stays on the device either way)."""
        from .linop import CsrOperator, HostOperatorShell
        if isinstance(self.op, CsrOperator):
            return self.op
        view = getattr(self.op, '_device_view', None)        # block operators whose blocks all live on the device
        if view is not None:
            dev = view()
            if dev is not None:
                return dev
        sh = getattr(self, '_host_shell', None)
        if sh is None or sh.host_op is not self.op:
            if not hasattr(self.op, 'shape') or not hasattr(self.op, '__mul__'):
                raise TypeError('%s needs an operator with `.shape` and `op * vector`; got %r'
                                % (self.__class__.__name__, type(self.op).__name__))
            sh = self._host_shell = HostOperatorShell(self.op)
        return sh

    def _device_precon(self, precon):
        """How the device loop applies ``precon * r`` (generic.py:76; cg.py:91-92, bicgstab.py:96-99, ...):

        * an operator that exposes its diagonal (``DiagonalOperator``, linop.py:473-516 -- the docs' Jacobi
          `DiagonalPrec`, examples/bmark.py:14-23) is returned as an fp64 array and applied INSIDE the kernels;
        * anything else with ``precon * vector`` is returned wrapped in :class:`HostPrecon`: the loop stays on the
          device and calls it back on the host at each preconditioner site."""
        if precon is None:
            return None
        diag = getattr(precon, 'diag', None)
        if diag is not None and not callable(diag):
            n = self.op.shape[0]
            return as_f64_vector(diag, getattr(self.op, 'local_size', None) or n, 'precon.diag')
        # * a device matrix (CsrOperator -- e.g. the inverted diagonal blocks of `tools.block_jacobi`) or a block
        #   operator of device matrices is applied ON the device, as a product at the same sites
        from .linop import CsrOperator
        dev = precon if isinstance(precon, CsrOperator) else None
        if dev is None and hasattr(precon, '_device_view'):
            dev = precon._device_view()
        if dev is not None and getattr(dev, 'local_size', None) is None:
            nloc = getattr(self.op, 'local_size', None) or self.op.shape[0]
            if dev.shape != (nloc, nloc):
                raise ValueError('%s: precon has shape %s, expected %s' % (self.__class__.__name__, dev.shape, (nloc, nloc)))
            return DevicePrecon(precon, dev)
        if not hasattr(precon, '__mul__') and not callable(precon):
            raise TypeError('%s: precon must support `precon * vector`; got %r'
                            % (self.__class__.__name__, type(precon).__name__))
        if getattr(self.op, 'local_size', None) is not None:
            raise NotImplementedError('%s: on a row-partitioned operator a preconditioner must be rank local and live on '
                                      'the device: a DiagonalOperator slice, or a device matrix / composite of the local '
                                      'size (e.g. pykrylov_amd.tools.block_jacobi(op, bs) of the partitioned operator); '
                                      'host callables are not called back across ranks' % self.__class__.__name__)
        return HostPrecon(precon)

    def _logging(self):
        return self.logger is not null_log and self.logger.isEnabledFor(logging.INFO)


class DevicePrecon(object):
    """A preconditioner that is itself a device operator (a CsrOperator, or a block operator with a device view): the
    loop evaluates ``precon * r`` as a product on the device (mk_solver_set_precon_csr) -- no host round trip."""

    def __init__(self, precon, dev):
        self.precon, self.dev = precon, dev

    def apply(self, r):
        return self.dev * r


class HostPrecon(object):
    """A general preconditioner for the device loop: ``precon * r`` evaluated on the host through a callback."""

    def __init__(self, precon):
        self.precon = precon
        self.error = None
        self.calls = 0


def as_f64_vector(v, n, what):
    a = np.asarray(v)
    if a.dtype.kind == 'c':
        raise TypeError('%s: complex data is not supported on the device path (fp64 only)' % what)
    if a.shape != (n,):
        raise ValueError('%s has shape %s, expected (%d,)' % (what, a.shape, n))
    return np.ascontiguousarray(a, dtype=np.float64)


class DeviceRun(object):
    """One solve on the device: owns the ``mk_solver`` handle and the rhs / guess buffers."""

    def __init__(self, op, kind, rhs, guess=None, precon_diag=None, **params):
        self.lib = _lib.init()
        self.op = op
        self.d_prec = None
        # `transpose`: the device operator A.T the least-squares kinds need (mk_solver_set_transpose); their rhs has
        # nrows(A) entries.  `placement_draws`: see _draws().
        transpose = params.pop('transpose', None)
        draws = params.pop('placement_draws', None)
        if draws is not None:
            self.placement_draws = int(draws)
        n = getattr(op, 'local_size', None) or (op.shape[0] if transpose is not None else op.shape[1])
        self.n = n                                           # entries of rhs / guess
        #  entries of the solution: ncols(A) for the least-squares kinds (CRAIG-MR's lives in m-space)
        self.n_x = n if (transpose is None or kind == _lib.MK_CRAIGMR) else op.shape[1]
        # rhs / guess: host arrays (copied to HBM) or DeviceArray objects already resident there
        self._borrowed = [b for b in (rhs, guess) if isinstance(b, _lib.DeviceArray)]
        self.d_rhs = rhs if isinstance(rhs, _lib.DeviceArray) else \
            _lib.DeviceArray.from_numpy(as_f64_vector(rhs, n, 'rhs'))
        self.d_guess = None if guess is None else (
            guess if isinstance(guess, _lib.DeviceArray) else
            _lib.DeviceArray.from_numpy(as_f64_vector(guess, n, 'guess')))
        if self.d_rhs.n != n or (self.d_guess is not None and self.d_guess.n != n):
            raise ValueError('rhs / guess must have %d entries' % n)
        p = _lib.MkParams()
        p.struct_size = ctypes.sizeof(_lib.MkParams)
        p.kind = kind
        for k, v in params.items():
            setattr(p, k, v)
        self._params = p
        self.handle = ctypes.c_void_p()
        _lib.check(self.lib.mk_solver_create(op.handle, ctypes.byref(p), ctypes.byref(self.handle)))
        self.transpose = transpose
        if transpose is not None:
            try:
                _lib.check(self.lib.mk_solver_set_transpose(self.handle, transpose.handle))
            except Exception:
                self.lib.mk_solver_destroy(self.handle)
                self.handle = ctypes.c_void_p()
                raise
        self.host_precon = None
        self.device_precon = None
        if isinstance(precon_diag, DevicePrecon):
            self.device_precon, precon_diag = precon_diag, None
            try:
                _lib.check(self.lib.mk_solver_set_precon_csr(self.handle, self.device_precon.dev.handle))
            except Exception:
                self.lib.mk_solver_destroy(self.handle)      # (no handle may outlive a failed constructor)
                self.handle = ctypes.c_void_p()
                raise
        if isinstance(precon_diag, HostPrecon):
            hp, self.host_precon, precon_diag = precon_diag, precon_diag, None

            def call(user, rp, yp):
                try:
                    r = np.ctypeslib.as_array(ctypes.cast(rp, ctypes.POINTER(ctypes.c_double)), shape=(n,)).copy()
                    y = np.asarray(hp.precon * r if hasattr(hp.precon, '__mul__') else hp.precon(r))
                    if y.shape != (n,):
                        raise ValueError('precon * r has shape %s, expected (%d,)' % (y.shape, n))
                    np.ctypeslib.as_array(ctypes.cast(yp, ctypes.POINTER(ctypes.c_double)), shape=(n,))[:] = y
                    hp.calls += 1
                    return 0
                except BaseException as exc:                # noqa: B902  (must not propagate through the C frames)
                    if hp.error is None:                    # (keep the root cause, not a follow-up failure)
                        hp.error = exc
                    return 1
            self._precon_cb = _lib.PRECON_FN(call)          # keep the thunk alive
            try:
                _lib.check(self.lib.mk_solver_set_precon_callback(self.handle, self._precon_cb, None))
            except Exception:
                self.lib.mk_solver_destroy(self.handle)      # (no handle may outlive a failed constructor)
                self.handle = ctypes.c_void_p()
                raise
        if precon_diag is not None:
            self.d_prec = precon_diag if isinstance(precon_diag, _lib.DeviceArray) else \
                _lib.DeviceArray.from_numpy(as_f64_vector(precon_diag, n, 'precon.diag'))
            if isinstance(precon_diag, _lib.DeviceArray):
                self._borrowed.append(precon_diag)
            _lib.check(self.lib.mk_solver_set_precon_diag(self.handle, self.d_prec.ptr))
        self.result = _lib.MkResult()
        self._setup_done = False
        self.placement = {"count": 1, "chosen": 0, "per_draw_ms_per_pass": [], "probe_seconds": 0.0, "passes_per_draw": 0}

    def _check(self, rc):
        if rc != 0 and hasattr(self.op, 'raise_pending'):
            self.op.raise_pending()                           # what a matrix-free operator raised in its callback
        if rc != 0 and self.host_precon is not None and self.host_precon.error is not None:
            err, self.host_precon.error = self.host_precon.error, None
            raise err
        _lib.check(rc)

    # Placement draws -- OPT-IN since round 4 (MK_PLACEMENT_DRAWS=k > 1, or `placement_draws=k` on the
    # DeviceRun).  How fast the fused update kernels run on a large problem depends on where the solver's vectors happen
    # to lie in HBM -- a property of the allocations, constant for their lifetime, 3-6 % of a CG pass at 512^3.  With
    # draws on, the first set-up creates k solver objects (each allocates its own vectors; spacer allocations in between
    # keep the draws apart), runs 4 + 12 real passes on each and keeps the fastest; the kept object is set up again
    # afterwards, so nothing of the probe survives but the choice.  A draw costs about 0.2 s and 4 GB (transient) at
    # 512^3 and pays back only after thousands of passes, which is why it is not the default: a solve that converges in
    # ~100 passes would get slower.  What was drawn is recorded in `self.placement`.
    # AUTOMATIC, QUICK draws (round 5) for the one class where they decide the product kernel too: CG with fused passes on a
    # brick-march matrix (formats 9 - 11) streams no matrix data to speak of, everything its kernels move lives in the solver's
    # own vectors, and about three draws in eight land in the fast state (pass 1.72 instead of 1.88 ms at 512^3,
    # profiles/r05_placement_draws_fused_march.txt).  There 6 draws of 2 + 6 passes run by themselves (about 25 ms each at
    # 512^3; all six: the states come in four levels, and stopping at the first draw that beats the first one was seen to
    # settle for a middle one) unless MK_PLACEMENT_DRAWS / `placement_draws` says otherwise (1 = off).
    def _auto_draw_class(self):
        if self._params.kind != _lib.MK_CG:
            return False
        fmt = ctypes.c_int32(0)
        try:
            _lib.check(self.lib.mk_csr_format_info(self.op.handle, ctypes.byref(fmt), None, None, None, None))
        except Exception:
            return False
        return 9 <= fmt.value <= 11

    def _draws(self):
        if self._setup_done or getattr(self, '_drawn', False):
            return 1
        want = getattr(self, 'placement_draws', None)
        self._quick_draws = False
        if want is None:
            env = os.environ.get('MK_PLACEMENT_DRAWS')
            want = int(env) if env else 0
        if want == 0:                                        # nobody asked: the automatic rule
            want, self._quick_draws = 6, True
        min_mb = float(os.environ.get('MK_PLACEMENT_MIN_MB', '256'))      # (tests lower it to draw on small problems)
        if want <= 1 or 8 * self.n <= min_mb * 1024 * 1024:
            return 1
        if getattr(self.op, 'local_size', None) is not None or self.host_precon is not None:
            return 1
        from .linop import CsrOperator
        if not isinstance(self.op, CsrOperator):             # (matrix-free shells: every pass calls back to the host)
            return 1
        if self._quick_draws and not self._auto_draw_class():
            return 1
        # the search costs about 65 passes' worth of time and buys 3-9 % per pass: it pays after about a thousand passes, so a
        # caller whose iteration budget is below that keeps the first solver object
        if self._quick_draws and 0 < int(self._params.matvec_max) < 1000:
            return 1
        return min(want, 8)

    def _apply_precon(self, handle):
        if getattr(self, 'transpose', None) is not None:
            _lib.check(self.lib.mk_solver_set_transpose(handle, self.transpose.handle))
        if self.device_precon is not None:
            _lib.check(self.lib.mk_solver_set_precon_csr(handle, self.device_precon.dev.handle))
        if self.d_prec is not None:
            _lib.check(self.lib.mk_solver_set_precon_diag(handle, self.d_prec.ptr))

    def _timed_passes(self, handle, warm=4, passes=12):
        g = None if self.d_guess is None else self.d_guess.ptr
        self._check(self.lib.mk_solver_setup(handle, self.d_rhs.ptr, g))
        done = ctypes.c_int64(0)
        self._check(self.lib.mk_solver_iterate(handle, warm, ctypes.byref(done)))
        self._check(self.lib.mk_sync())
        t0 = time.perf_counter()
        self._check(self.lib.mk_solver_iterate(handle, passes, ctypes.byref(done)))
        self._check(self.lib.mk_sync())
        return (time.perf_counter() - t0) / max(1, done.value) if done.value == passes else float('inf')

    def _draw_placement(self, draws):
        self._drawn = True
        spacers, best, best_t = [], self.handle, None
        t_probe = time.perf_counter()
        per_draw, chosen = [], 0
        quick = getattr(self, '_quick_draws', False)
        warm, passes = (2, 6) if quick else (4, 12)
        try:
            best_t = self._timed_passes(self.handle, warm, passes)
            per_draw.append(1e3 * best_t)
            for k in range(1, draws):
                try:
                    sp_mb = int(os.environ.get('MK_PLACEMENT_SPACER_MB', '176'))
                    spacers.append(_lib.DeviceArray(((sp_mb + (sp_mb // 2) * k) << 20) // 8 + 512 * k, zero=False))
                    h = ctypes.c_void_p()
                    _lib.check(self.lib.mk_solver_create(self.op.handle, ctypes.byref(self._params), ctypes.byref(h)))
                except Exception:
                    break                                     # (no memory for another draw: keep what we have)
                try:
                    self._apply_precon(h)
                    t = self._timed_passes(h, warm, passes)
                except Exception:
                    self.lib.mk_solver_destroy(h)
                    break
                per_draw.append(1e3 * t)
                if t < 0.99 * best_t:
                    self.lib.mk_solver_destroy(best)
                    best, best_t, chosen = h, t, k
                else:
                    self.lib.mk_solver_destroy(h)
        finally:
            self.handle = best
            for sp in spacers:
                sp.free()
            self.placement = {"count": len(per_draw), "chosen": chosen, "per_draw_ms_per_pass": per_draw,
                              "probe_seconds": time.perf_counter() - t_probe, "passes_per_draw": warm + passes,
                              "automatic": bool(quick)}

    def setup(self):
        draws = self._draws()
        if draws > 1:
            self._draw_placement(draws)
        self._check(self.lib.mk_solver_setup(self.handle, self.d_rhs.ptr,
                                             None if self.d_guess is None else self.d_guess.ptr))
        self._setup_done = True

    def iterate(self, max_iters):
        done = ctypes.c_int64(0)
        self._check(self.lib.mk_solver_iterate(self.handle, int(max_iters), ctypes.byref(done)))
        return done.value

    def finish(self):
        _lib.check(self.lib.mk_solver_finish(self.handle, ctypes.byref(self.result)))
        return self.result

    def run(self):
        """setup + loop until the reference's `while` condition fails + epilogue."""
        self.setup()
        res = self.finish()
        while not res.halted:
            self.iterate(1 << 20)
            res = self.finish()
        return res

    def x(self):
        p = ctypes.c_void_p()
        _lib.check(self.lib.mk_solver_x(self.handle, ctypes.byref(p)))
        return _lib.download(p.value, self.n_x)

    def vector(self, index):
        p, ln = ctypes.c_void_p(), ctypes.c_int64()
        _lib.check(self.lib.mk_solver_vector(self.handle, index, ctypes.byref(p), ctypes.byref(ln)))
        return _lib.download(p.value, ln.value)

    def x_first(self):
        """x[0] of the current iterate (eight bytes from the device: what the reference's `show` tables print)."""
        px = ctypes.c_void_p()
        _lib.check(self.lib.mk_solver_x(self.handle, ctypes.byref(px)))
        v = ctypes.c_double()
        _lib.check(self.lib.mk_memcpy_d2h(ctypes.byref(v), px.value, 8))
        return v.value

    def history(self):
        n = int(self.result.hist_len)
        out = np.empty(n, dtype=np.float64)
        _lib.check(self.lib.mk_solver_history(self.handle, out.ctypes.data, n))
        return out

    def time_product(self, which=0, launches=200):
        """Average duration (us) of the pass's product kernel `which` (0 first, 1 second / A.T), launched back to
        back without its gate and bracketed by one HIP event pair.  Destroys the loop's vectors: call it last."""
        avg = ctypes.c_double(float('nan'))
        _lib.check(self.lib.mk_solver_time_product(self.handle, int(which), int(launches), ctypes.byref(avg)))
        return avg.value

    def timing(self):
        it_ms, sp_ms, cnt = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        _lib.check(self.lib.mk_solver_timing(self.handle, ctypes.byref(it_ms), ctypes.byref(sp_ms),
                                             ctypes.byref(cnt)))
        return {"iterate_ms": it_ms.value, "spmv_ms": sp_ms.value, "spmv_launches": cnt.value}

    def close(self):
        if getattr(self, 'handle', None) is not None and self.handle.value:
            self.lib.mk_solver_destroy(self.handle)
            self.handle = ctypes.c_void_p()
        for b in (self.d_rhs, self.d_guess, getattr(self, 'd_prec', None)):
            if b is not None and not any(b is x for x in getattr(self, '_borrowed', [])):
                b.free()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def solve_guess_matvec_max(solver, kind, rhs, kwargs, count_guess_product):
    """Shared body of the BiCGSTAB / CGS / TFQMR `solve` methods: keywords `guess` and `matvec_max`
    (default 2n), result attributes `converged, nMatvec, bestSolution, x, residNorm, residNorm0`
    (reference bicgstab.py:148-151, cgs.py:120-123, tfqmr.py:156-159).  These solvers keep no history."""
    op = solver._device_operator()
    pdiag = solver._device_precon(solver.precon)
    n = getattr(op, 'global_size', None) or rhs.shape[0]        # same limit on every rank of a partitioned run
    guess = kwargs.get('guess', None)
    matvec_max = kwargs.get('matvec_max', 2 * n)
    with DeviceRun(op, kind, rhs, guess, precon_diag=pdiag, abstol=float(solver.abstol),
                   reltol=float(solver.reltol), matvec_max=int(matvec_max)) as run:
        res = run.run()
        x = run.x()
    # the product that forms the initial residual from a guess is counted by the operator, but by the
    # solver only in BiCGSTAB (bicgstab.py:64-65 vs cgs.py:59-60, tfqmr.py:58-59)
    op._nMatvec += int(res.nMatvec) + (1 if (guess is not None and not count_guess_product) else 0)
    solver.residNorm0 = np.float64(res.residNorm0)
    if solver._logging():
        solver.logger.info('Initial residual = %8.2e' % solver.residNorm0)
        solver.logger.info('Threshold = %8.2e' % res.threshold)
        solver.logger.info('%6d  %8.2e' % (res.nMatvec, res.residNorm))
    solver.converged = bool(res.converged)
    solver.nMatvec = int(res.nMatvec)
    solver.bestSolution = solver.x = x
    solver.residNorm = np.float64(res.residNorm)
    return res
