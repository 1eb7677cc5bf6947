"""CPU oracle: a NumPy restatement of pykrylov's solver loops.   TEST INFRASTRUCTURE ONLY.

Nothing in ``pykrylov_amd/`` may import this module; it exists so that tests,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg have a
checker that travels to the GPU box (the reference itself cannot).

Pinned: ``tests/test_oracle_golden.py`` checks every function here bit-for-bit
(iteration counts, residual histories, iterates) against fixtures produced by
the real reference (``tests/golden/make_golden.py``).

Each function follows the op sequence of the cited reference lines, including
the in-place update order (``p *= beta; p -= r``), because NumPy rounds every
elementary operation separately and the order is therefore part of the result.
All solvers take the operator as a plain callable ``A(v) -> new ndarray`` (what
``LinearOperator.__mul__`` amounts to, reference ``pykrylov/linop/linop.py:356-360``)
and an optional :class:`Reductions` object through which every dot / norm goes
(so tests can record the reduction trace or substitute the GPU's summation order).
"""
import numpy as np

EPS = np.finfo(np.double).eps            # tools/utils.py:7-9


class Reductions(object):
    """All inner products of a solve go through here, in call order.

    ``dot_impl(a, b, site)`` may be replaced (e.g. by ``gpu_order.GpuDots``) to
    reproduce a different summation order; ``site`` names the call site.
    ``trace`` collects every result (dots as returned, norms after the sqrt), the
    same thing ``make_golden.py`` records from the reference.
    """

    def __init__(self, dot_impl=None):
        self.dot_impl = dot_impl
        self.trace = []

    def dot(self, a, b, site=""):
        v = np.float64(np.dot(a, b)) if self.dot_impl is None else np.float64(self.dot_impl(a, b, site))
        self.trace.append(float(v))
        return v

    def norm(self, a, site=""):
        # np.linalg.norm of a real 1-D array is sqrt(dot(a, a)) (numpy/linalg/_linalg.py)
        if self.dot_impl is None:
            v = np.float64(np.linalg.norm(a))
        else:
            v = np.sqrt(np.float64(self.dot_impl(a, a, site)))
        self.trace.append(float(v))
        return v


def _setup(rhs, guess):
    n = rhs.shape[0]
    if guess is None:
        return n, np.zeros(n, dtype=np.float64)
    return n, np.array(guess, dtype=np.float64)     # .astype() copies (cg.py:77)


# --------------------------------------------------------------------------- #
# CG   -- reference pykrylov/cg/cg.py:46-165
# --------------------------------------------------------------------------- #
def cg(A, rhs, abstol=1.0e-8, reltol=1.0e-6, guess=None, matvec_max=None, precon=None,
       check_curvature=True, red=None):
    red = red or Reductions()
    n, x = _setup(rhs, guess)
    if matvec_max is None:
        matvec_max = 2 * n                                   # cg.py:82
    nmv = 0
    r = -rhs                                                 # cg.py:85
    if guess is not None:
        r += A(x)                                            # cg.py:86-88
        nmv += 1
    y = precon(r) if precon is not None else r               # cg.py:91-94
    ry = red.dot(r, y, "cg.ry0")                             # cg.py:99
    resid0 = resid = np.abs(np.sqrt(ry))                     # cg.py:100
    hist = [resid0]
    thresh = max(abstol, reltol * resid0)                    # cg.py:102
    p = -r                                                   # cg.py:104
    definite = True
    descent = None
    while resid > thresh and nmv < matvec_max and definite:  # cg.py:113
        Ap = A(p)
        nmv += 1
        pAp = red.dot(p, Ap, "cg.pAp")                       # cg.py:117
        if check_curvature and pAp <= 0:                     # cg.py:119-124 (real data: imag == 0)
            descent = p
            definite = False
            continue
        alpha = ry / pAp                                     # cg.py:127
        x += alpha * p                                       # cg.py:130
        r += alpha * Ap                                      # cg.py:131
        y = precon(r) if precon is not None else r           # cg.py:137-140
        ry_next = red.dot(r, y, "cg.ry")                     # cg.py:146
        beta = ry_next / ry                                  # cg.py:149
        p *= beta                                            # cg.py:150
        p -= r                                               # cg.py:151
        ry = ry_next
        resid = np.abs(np.sqrt(ry))                          # cg.py:154
        hist.append(resid)
    return dict(x=x, nMatvec=nmv, residNorm=resid, residNorm0=resid0, residHistory=np.array(hist),
                converged=bool(resid <= thresh), definite=definite, infiniteDescent=descent,
                threshold=thresh, trace=np.array(red.trace))


# --------------------------------------------------------------------------- #
# Bi-CGSTAB   -- reference pykrylov/bicgstab/bicgstab.py:43-151
# --------------------------------------------------------------------------- #
def bicgstab(A, rhs, abstol=1.0e-8, reltol=1.0e-6, guess=None, matvec_max=None, precon=None, red=None):
    red = red or Reductions()
    n, x = _setup(rhs, guess)
    if matvec_max is None:
        matvec_max = 2 * n
    nmv = 0
    r0 = rhs                                                 # bicgstab.py:62 (alias, never written)
    if guess is not None:
        r0 = rhs - A(x)                                      # bicgstab.py:63-65
        nmv += 1
    rho = alpha = omega = 1.0
    rho_next = red.dot(r0, r0, "bicgstab.rho0")              # bicgstab.py:68
    resid = resid0 = np.abs(np.sqrt(rho_next))
    thresh = max(abstol, reltol * resid0)
    finished = bool(resid <= thresh or nmv >= matvec_max)    # bicgstab.py:72
    if not finished:
        r = r0.copy()
        p = np.zeros(n)
        v = np.zeros(n)
    while not finished:
        beta = rho_next / rho * alpha / omega                # bicgstab.py:87
        rho = rho_next
        p *= beta                                            # bicgstab.py:91-93
        p -= beta * omega * v
        p += r
        q = precon(p) if precon is not None else p
        v = A(q)                                             # bicgstab.py:101
        nmv += 1
        alpha = rho / red.dot(r0, v, "bicgstab.r0v")         # bicgstab.py:103
        s = r - alpha * v                                    # bicgstab.py:104
        resid = red.norm(s, "bicgstab.s")                    # bicgstab.py:107
        if resid <= thresh:                                  # bicgstab.py:111-114
            x += alpha * q
            finished = True
            continue
        if nmv >= matvec_max:                                # bicgstab.py:116-118
            finished = True
            continue
        z = precon(s) if precon is not None else s
        t = A(z)                                             # bicgstab.py:125
        nmv += 1
        omega = red.dot(t, s, "bicgstab.ts") / red.dot(t, t, "bicgstab.tt")    # bicgstab.py:126
        rho_next = -omega * red.dot(r0, t, "bicgstab.r0t")   # bicgstab.py:127
        r = s - omega * t                                    # bicgstab.py:130
        z *= omega                                           # bicgstab.py:135 (overwrites s when z is s)
        x += z                                               # bicgstab.py:136
        x += alpha * q                                       # bicgstab.py:137
        resid = red.norm(r, "bicgstab.r")                    # bicgstab.py:139
        if resid <= thresh or nmv >= matvec_max:
            finished = True
    return dict(x=x, nMatvec=nmv, residNorm=resid, residNorm0=resid0, converged=bool(resid <= thresh),
                threshold=thresh, trace=np.array(red.trace))


# --------------------------------------------------------------------------- #
# CGS   -- reference pykrylov/cgs/cgs.py:40-123
# --------------------------------------------------------------------------- #
def cgs(A, rhs, abstol=1.0e-8, reltol=1.0e-6, guess=None, matvec_max=None, precon=None, red=None):
    red = red or Reductions()
    n, x = _setup(rhs, guess)
    if matvec_max is None:
        matvec_max = 2 * n
    nmv = 0
    r0 = rhs
    if guess is not None:
        r0 = rhs - A(x)                                      # cgs.py:59-60 (not counted in nMatvec)
    rho = red.dot(r0, r0, "cgs.rho0")                        # cgs.py:62
    resid = resid0 = np.abs(np.sqrt(rho))
    thresh = max(abstol, reltol * resid0)
    finished = bool(resid <= thresh or nmv >= matvec_max)
    if not finished:
        r = r0.copy()
        u = r0                                               # cgs.py:73 (alias until rebound)
        p = r0.copy()
    while not finished:
        y = precon(p) if precon is not None else p
        v = A(y)                                             # cgs.py:83
        nmv += 1
        sigma = red.dot(r0, v, "cgs.sigma")                  # cgs.py:84
        alpha = rho / sigma
        q = u - alpha * v                                    # cgs.py:86
        z = precon(u + q) if precon is not None else u + q   # cgs.py:88-91
        x += alpha * z                                       # cgs.py:94
        Az = A(z)                                            # cgs.py:95
        nmv += 1
        r -= alpha * Az                                      # cgs.py:96
        resid = red.norm(r, "cgs.r")                         # cgs.py:99
        if resid <= thresh or nmv >= matvec_max:
            finished = True
            continue
        rho_next = red.dot(r0, r, "cgs.rho")                 # cgs.py:105
        beta = rho_next / rho
        rho = rho_next
        u = r + beta * q                                     # cgs.py:108
        p *= beta                                            # cgs.py:111-114
        p += q
        p *= beta
        p += u
    return dict(x=x, nMatvec=nmv, residNorm=resid, residNorm0=resid0, converged=bool(resid <= thresh),
                threshold=thresh, trace=np.array(red.trace))


# --------------------------------------------------------------------------- #
# TFQMR   -- reference pykrylov/tfqmr/tfqmr.py:39-159
# --------------------------------------------------------------------------- #
def tfqmr(A, rhs, abstol=1.0e-8, reltol=1.0e-6, guess=None, matvec_max=None, precon=None, red=None):
    red = red or Reductions()
    n, x = _setup(rhs, guess)
    if matvec_max is None:
        matvec_max = 2 * n
    nmv = 0
    r0 = rhs
    if guess is not None:
        r0 = rhs - A(x)                                      # tfqmr.py:58-59 (not counted)
    rho = red.dot(r0, r0, "tfqmr.rho0")
    resid = resid0 = np.abs(np.sqrt(rho))
    thresh = max(abstol, reltol * resid0)
    finished = bool(resid <= thresh or nmv >= matvec_max)
    m = 0.0              # the reference leaves m unbound here and raises (tfqmr.py:156); fixed: m = 0
    if not finished:
        y = r0.copy()
        w = r0.copy()
        d = np.zeros(n)
        theta = 0.0
        eta = 0.0
        k = 0
        z = precon(y) if precon is not None else y          # tfqmr.py:77-80 (z aliases y)
        u = A(z)
        nmv += 1
        v = u.copy()
    while not finished:
        k += 1
        sigma = red.dot(r0, v, "tfqmr.sigma")                # tfqmr.py:88
        alpha = rho / sigma
        # first half-step
        w -= alpha * u                                       # tfqmr.py:92
        d *= theta * theta * eta / alpha                     # tfqmr.py:93
        d += z
        theta = red.norm(w, "tfqmr.w1") / resid              # tfqmr.py:95
        c = 1.0 / np.sqrt(1 + theta * theta)
        resid *= theta * c
        eta = c * c * alpha
        x += eta * d                                         # tfqmr.py:99
        m = 2.0 * k - 1.0
        if resid * np.sqrt(m + 1) < thresh or nmv >= matvec_max:
            finished = True
            continue
        # second half-step
        m += 1
        y -= alpha * v                                       # tfqmr.py:107 (also changes z when z is y)
        z = precon(y) if precon is not None else y
        u = A(z)                                             # tfqmr.py:114
        nmv += 1
        w -= alpha * u
        d *= theta * theta * eta / alpha
        d += z
        theta = red.norm(w, "tfqmr.w2") / resid
        c = 1.0 / np.sqrt(1 + theta * theta)
        resid *= theta * c
        eta = c * c * alpha
        x += eta * d
        if resid * np.sqrt(m + 1) < thresh or nmv >= matvec_max:
            finished = True
            continue
        rho_next = red.dot(r0, w, "tfqmr.rho")               # tfqmr.py:128
        beta = rho_next / rho
        rho = rho_next
        y *= beta                                            # tfqmr.py:133-134
        y += w
        v *= beta                                            # tfqmr.py:137-139
        v += u
        v *= beta
        z = precon(y) if precon is not None else y
        u = A(z)                                             # tfqmr.py:147
        nmv += 1
        v += u                                               # tfqmr.py:150
    return dict(x=x, nMatvec=nmv, residNorm=resid, residNorm0=resid0,
                converged=bool(resid * np.sqrt(m + 1) < thresh), threshold=thresh,
                trace=np.array(red.trace))


# --------------------------------------------------------------------------- #
# cheap symmetry test   -- reference pykrylov/tools/utils.py:63-85
# --------------------------------------------------------------------------- #
def check_symmetric(A, n, repeats=10, red=None):
    red = red or Reductions()
    np.random.seed(1)                                        # utils.py:74 (reseeds the global RNG)
    for _ in range(repeats):
        x = np.random.random(n)
        w = A(x)
        r = A(w)
        s = red.dot(w, w, "sym.ww")
        t = red.dot(x, r, "sym.xr")
        if abs(s - t) > (s + EPS) * EPS ** (1.0 / 3):
            return False
    return True


def _sq(a):
    """x**2 exactly as the reference writes it.  For NumPy scalars this is libm's pow(x, 2.0), which is
    NOT always the correctly rounded x*x (about 1 in 1200 arguments differs by one ulp).  The device
    computes x*x; tests that demand bit equality with the device swap this hook for ``lambda a: a * a``."""
    return a ** 2


def _hyp(a, b):
    return np.sqrt(_sq(a) + _sq(b))                          # minres.py:112-113


# --------------------------------------------------------------------------- #
# MINRES   -- reference pykrylov/minres/minres.py:115-410
# --------------------------------------------------------------------------- #
def minres(A, b, precon=None, shift=0.0, check=True, itnlim=None, rtol=1.0e-12, etol=1.0e-6,
           window=5, red=None):
    red = red or Reductions()
    n = b.shape[0]
    if itnlim is None:
        itnlim = 5 * n
    eps = EPS
    x = np.zeros(n)
    x_nrg2 = 0.0
    d_err = np.zeros(window)
    dir_errors = []
    hist = []
    status = None
    istop = 0
    itn = 0
    Anorm = Acond = rnorm = ynorm = 0.0
    done = False

    r1 = b                                                   # minres.py:161 (alias)
    y = precon(b) if precon is not None else b.copy()
    beta1 = red.dot(b, y, "minres.beta1")                    # minres.py:166
    if beta1 < 0:
        istop = 9
        done = True
    if beta1 == 0.0:
        done = True
    if beta1 > 0:
        beta1 = np.sqrt(beta1)
    resid0 = beta1
    if check:                                                # minres.py:186-190
        if not check_symmetric(A, n, red=Reductions(red.dot_impl)):
            istop = 7
            done = True
    if check and precon is not None:
        if not check_symmetric(precon, n, red=Reductions(red.dot_impl)):
            istop = 8
            done = True

    oldb = 0.0; beta = beta1; dbar = 0.0; epsln = 0.0        # minres.py:202-205
    qrnorm = beta1; phibar = beta1; rhs1 = beta1; Arnorm = 0.0
    rhs2 = 0.0; tnorm2 = 0.0; ynorm2 = 0.0
    cs = -1.0; sn = 0.0
    w = np.zeros(n)
    w2 = np.zeros(n)
    r2 = r1.copy()

    if not done:
        while itn < itnlim:
            itn += 1
            s = 1.0 / beta                                   # minres.py:236-237
            v = s * y
            y = A(v)                                         # minres.py:239-240
            y -= shift * v
            if itn >= 2:
                y = y - (beta / oldb) * r1                   # minres.py:243
            alfa = red.dot(v, y, "minres.alfa")              # minres.py:245
            y = (-alfa / beta) * r2 + y                      # minres.py:246
            r1 = r2.copy()
            r2 = y.copy()
            if precon is not None:
                y = precon(r2)
            oldb = beta
            beta = red.dot(r2, y, "minres.beta")             # minres.py:251
            if beta < 0:
                istop = 6
                break
            beta = np.sqrt(beta)
            tnorm2 = tnorm2 + _sq(alfa) + _sq(oldb) + _sq(beta)
            if itn == 1:
                if beta / beta1 <= 10 * eps:
                    istop = -1
                gmax = abs(alfa)
                gmin = gmax
            # previous rotation                                minres.py:270-278
            oldeps = epsln
            delta = cs * dbar + sn * alfa
            gbar = sn * dbar - cs * alfa
            epsln = sn * beta
            dbar = -cs * beta
            root = _hyp(gbar, dbar)
            Arnorm = phibar * root
            # next rotation                                    minres.py:282-287
            gamma = _hyp(gbar, beta)
            gamma = max(gamma, eps)
            cs = gbar / gamma
            sn = beta / gamma
            phi = cs * phibar
            phibar = sn * phibar
            # solution update                                  minres.py:291-297
            denom = 1.0 / gamma
            w1 = w2.copy()
            w2 = w.copy()
            w = (v - oldeps * w1 - delta * w2) * denom
            x += phi * w
            # direct-error window                              minres.py:302-310
            x_nrg2 += phi * phi
            d_err[itn % window] = phi
            if itn > window:
                trnc = np.linalg.norm(d_err)                 # length-`window` host array (minres.py:306)
                red.trace.append(float(trnc))
                x_nrg = np.sqrt(x_nrg2)
                dir_errors.append(trnc / x_nrg)
                if trnc < etol * x_nrg:
                    istop = 10
            gmax = max(gmax, gamma)
            gmin = min(gmin, gamma)
            z = rhs1 / gamma
            ynorm2 = _sq(z) + ynorm2
            rhs1 = rhs2 - delta * z
            rhs2 = -epsln * z
            # norm estimates and stopping tests                minres.py:323-361
            Anorm = np.sqrt(tnorm2)
            ynorm = np.sqrt(ynorm2)
            epsa = Anorm * eps
            epsx = Anorm * ynorm * eps
            qrnorm = phibar
            rnorm = qrnorm
            test1 = rnorm / (Anorm * ynorm)
            test2 = root / Anorm
            hist.append(rnorm)
            Acond = gmax / gmin
            if istop == 0:
                t1 = 1 + test1
                t2 = 1 + test2
                if t2 <= 1: istop = 2
                if t1 <= 1: istop = 1
                if itn >= itnlim: istop = 6
                if Acond >= 0.1 / eps: istop = 4
                if epsx >= beta1: istop = 3
                if test2 <= rtol: istop = 2
                if test1 <= rtol: istop = 1
            if istop > 0:
                break
    if istop == 10:
        status = "direct error small"
    return dict(x=x, istop=istop, itn=itn, nMatvec=itn, rnorm=rnorm, residNorm=rnorm, Arnorm=Arnorm,
                Anorm=Anorm, Acond=Acond, ynorm=ynorm, residNorm0=resid0,
                converged=istop in (1, 2, 3, 4, 10), status=status, residHistory=np.array(hist),
                dir_errors_window=np.array(dir_errors), trace=np.array(red.trace))


# --------------------------------------------------------------------------- #
# SYMMLQ   -- reference pykrylov/symmlq/symmlq.py:65-400
# (the reference calls a non-existent self.matvec at :162; restated as op * v)
# --------------------------------------------------------------------------- #
def symmlq(A, rhs, precon=None, matvec_max=None, rtol=1.0e-9, check=False, shift=None, red=None):
    red = red or Reductions()
    n = rhs.shape[0]
    if matvec_max is None:
        matvec_max = 2 * n + 2
    if shift == 0.0:
        shift = None
    eps = EPS
    nmv = 0
    istop = 0; ynorm = 0; w = np.zeros(n); acond = 0
    itn = 0; xnorm = 0; x = np.zeros(n); done = False
    anorm = 0; rnorm = 0; v = np.zeros(n)

    r1 = rhs.copy()
    y = precon(r1) if precon is not None else rhs.copy()
    b1 = y[0]
    beta1 = red.dot(r1, y, "symmlq.beta1")                   # symmlq.py:134
    if check and precon is not None:                         # symmlq.py:138-146
        r2 = precon(y)
        s = red.dot(y, y, "symmlq.chk.yy")
        t = red.dot(r1, r2, "symmlq.chk.r1r2")
        if np.abs(s - t) > (s + eps) * eps ** (1.0 / 3):
            istop = 7
            done = True
    if beta1 < 0:
        istop = 8
        done = True
    if beta1 == 0:
        done = True
    # quantities the reference only defines inside "if beta1 > 0" but reads afterwards
    cgnorm = lqnorm = 0.0; rhs1 = 0.0; diag = 1.0; snprod = 1; bstep = 0; ynorm2 = 0
    if beta1 > 0:
        beta1 = np.sqrt(beta1)
        s = 1.0 / beta1
        v = s * y
        y = A(v)                                             # symmlq.py:162
        nmv += 1
        if check:                                            # symmlq.py:163-171 (not counted)
            r2 = A(y)
            s = red.dot(y, y, "symmlq.chk.yy2")
            t = red.dot(v, r2, "symmlq.chk.vr2")
            if abs(s - t) > (s + eps) * eps ** (1.0 / 3):
                istop = 6
                done = True
        if shift is not None:
            y -= shift * v
        alfa = red.dot(v, y, "symmlq.alfa1")                 # symmlq.py:178
        y -= (alfa / beta1) * r1
        z = red.dot(v, y, "symmlq.vy")                       # symmlq.py:183-186 (local reorthogonalisation)
        s = red.dot(v, v, "symmlq.vv")
        y -= (z / s) * v
        r2 = y.copy()
        if precon is not None:
            y = precon(r2)
        oldb = beta1
        beta = red.dot(r2, y, "symmlq.beta2")                # symmlq.py:190
        if beta < 0:
            istop = 8
            done = True
        beta = np.sqrt(beta)
        if beta <= eps:
            istop = -1
        denom = np.sqrt(s) * red.norm(r2, "symmlq.r2") + eps  # symmlq.py:203
        s = z / denom
        t = red.dot(v, r2, "symmlq.vr2")                     # symmlq.py:205
        t = t / denom
        cgnorm = beta1; rhs2 = 0; tnorm = _sq(alfa) + _sq(beta)   # symmlq.py:212-217
        gbar = alfa; bstep = 0; ynorm2 = 0
        dbar = beta; snprod = 1; gmax = np.abs(alfa) + eps
        rhs1 = beta1; x1cg = 0; gmin = gmax
        qrnorm = beta1

    if not done:
        while nmv < matvec_max:                              # symmlq.py:235
            itn += 1
            anorm = np.sqrt(tnorm)
            ynorm = np.sqrt(ynorm2)
            epsa = anorm * eps
            epsx = anorm * ynorm * eps
            epsr = anorm * ynorm * rtol
            diag = gbar
            if diag == 0:
                diag = epsa
            lqnorm = np.sqrt(_sq(rhs1) + _sq(rhs2))
            qrnorm = snprod * beta1
            cgnorm = qrnorm * beta / np.abs(diag)
            if lqnorm < cgnorm:                              # symmlq.py:257-261
                acond = gmax / gmin
            else:
                denom = min(gmin, np.abs(diag))
                acond = gmax / denom
            if istop == 0:                                   # symmlq.py:271-276
                if nmv >= matvec_max: istop = 5
                if acond >= 0.1 / eps: istop = 4
                if epsx >= beta1: istop = 3
                if cgnorm <= epsx: istop = 2
                if cgnorm <= epsr: istop = 1
            if istop != 0:
                break
            s = 1 / beta                                     # symmlq.py:300-306
            v = s * y
            y = A(v)
            nmv += 1
            if shift is not None:
                y -= shift * v
            y -= (beta / oldb) * r1
            alfa = red.dot(v, y, "symmlq.alfa")
            y -= (alfa / beta) * r2
            r1 = r2.copy()
            r2 = y.copy()
            if precon is not None:
                y = precon(r2)
            oldb = beta
            beta = red.dot(r2, y, "symmlq.beta")             # symmlq.py:311
            if beta < 0:
                istop = 6
                break
            beta = np.sqrt(beta)
            tnorm = tnorm + _sq(alfa) + _sq(oldb) + _sq(beta)
            gamma = np.sqrt(_sq(gbar) + _sq(oldb))           # symmlq.py:322-328
            cs = gbar / gamma
            sn = oldb / gamma
            delta = cs * dbar + sn * alfa
            gbar = sn * dbar - cs * alfa
            epsln = sn * beta
            dbar = -cs * beta
            z = rhs1 / gamma                                 # symmlq.py:332-336
            s = z * cs
            t = z * sn
            x += s * w + t * v
            w *= sn
            w -= cs * v
            bstep = snprod * cs * z + bstep                  # symmlq.py:343-349
            snprod = snprod * sn
            gmax = max(gmax, gamma)
            gmin = min(gmin, gamma)
            ynorm2 = _sq(z) + ynorm2
            rhs1 = rhs2 - delta * z
            rhs2 = -epsln * z

    if cgnorm < lqnorm:                                      # symmlq.py:361-365 (move to the CG point)
        zbar = rhs1 / diag
        bstep = snprod * zbar + bstep
        ynorm = np.sqrt(ynorm2 + _sq(zbar))
        x += zbar * w
    if beta1 != 0:
        bstep = bstep / beta1                                # symmlq.py:369
    y = precon(rhs) if precon is not None else rhs.copy()
    x += bstep * y
    y = A(x)                                                 # symmlq.py:378-382 (final residual)
    nmv += 1
    if shift is not None:
        y -= shift * x
    r1 = rhs - y
    rnorm = red.norm(r1, "symmlq.rnorm")
    xnorm = red.norm(x, "symmlq.xnorm")
    return dict(x=x, nMatvec=nmv, residNorm=rnorm, xNorm=xnorm, solutionNorm=xnorm, anorm=anorm,
                acond=acond, istop=istop, itn=itn, trace=np.array(red.trace))
