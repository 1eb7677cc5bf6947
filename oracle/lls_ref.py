"""CPU oracle for the least-squares family (LSQR, LSMR, CRAIG, CRAIG-MR).   TEST INFRASTRUCTURE ONLY.

NumPy restatement of reference ``pykrylov/lls/{lsqr,lsmr,craig,craigmr}.py`` for the unpreconditioned
case (``M = N = None``).  All four share the Golub-Kahan bidiagonalisation step

    Mu = A v - alpha Mu ;  beta = sqrt(<Mu, Mu>) ;  u = Mu / beta      (u aliases Mu: both are normalised)
    Nv = A' u - beta Nv ;  alpha = sqrt(<Nv, Nv>) ; v = Nv / alpha

(lsqr.py:252-271, lsmr.py:311-332, craig.py:302-329, craigmr.py:136-157) and differ in the scalar
recurrences and vector updates that follow.  Scalars are Python floats here as in the reference
(``from math import sqrt``).  Pinned bit-for-bit by ``tests/test_oracle_golden.py`` against fixtures
generated from the real reference.
"""
from math import sqrt

import numpy as np

from .krylov_ref import Reductions

inf = float("inf")


def _sq(a):
    """``a**2`` as the reference writes it (libm pow for Python floats; see krylov_ref._sq)."""
    return a ** 2


def _hyp(x, y):
    return sqrt(x * x + y * y)                                   # lsqr.py:23 normof2


def _hyp4(a, b, c, d):
    return sqrt(a * a + b * b + c * c + d * d)                   # lsqr.py:24 normof4


class _Metric(object):
    """The optional preconditioners of the lls solvers: callables M (m-space) and N (n-space) applied as
    `u = M(Mu)`, `v = N(Nv)` (lsqr.py:189-190,201-202,253-254,265-266 and the same lines of lsmr / craig /
    craigmr), together with the unpreconditioned companions Mu, Nv the reference carries along."""

    def __init__(self, M=None, N=None):
        self.M, self.N = M, N
        self.Mu = self.Nv = None


def _gk_start(A, At, b, red, tag, met=None):
    """First bidiagonalisation vectors (lsqr.py:188-209): beta M u = b, alpha N v = A' u."""
    met = met or _Metric()
    Mu = b.copy()
    u = met.M(Mu) if met.M is not None else Mu
    alpha = 0.0
    v = Nv = None
    beta = sqrt(red.dot(u, Mu, tag + ".beta0"))
    if beta > 0:
        u /= beta
        if met.M is not None:
            Mu /= beta
        Nv = At(u)
        v = met.N(Nv) if met.N is not None else Nv
        alpha = sqrt(red.dot(v, Nv, tag + ".alpha0"))
    if alpha > 0:
        v /= alpha
        if met.N is not None:
            Nv /= alpha
    met.Mu, met.Nv = Mu, Nv
    return u, v, alpha, beta


def _gk_step(A, At, u, v, alpha, red, tag, met=None):
    """One bidiagonalisation step (lsqr.py:252-272); without preconditioners u is Mu and v is Nv, as in the
    reference, where the names alias."""
    met = met or _Metric()
    Mu = met.Mu if met.M is not None else u
    Nv = met.Nv if met.N is not None else v
    Mu_new = A(v) - alpha * Mu
    u_new = met.M(Mu_new) if met.M is not None else Mu_new
    beta = sqrt(red.dot(u_new, Mu_new, tag + ".beta"))
    if beta > 0:
        u_new /= beta
        if met.M is not None:
            Mu_new /= beta
        Nv_new = At(u_new) - beta * Nv
        v_new = met.N(Nv_new) if met.N is not None else Nv_new
        alpha = sqrt(red.dot(v_new, Nv_new, tag + ".alpha"))
        if alpha > 0:
            v_new /= alpha
            if met.N is not None:
                Nv_new /= alpha
        v = v_new
        met.Nv = Nv_new
    met.Mu = Mu_new
    return u_new, v, alpha, beta


# --------------------------------------------------------------------------- #
# LSQR   -- reference pykrylov/lls/lsqr.py:86-453
# --------------------------------------------------------------------------- #
def lsqr(A, At, shape, rhs, itnlim=0, damp=0.0, atol=1.0e-9, btol=1.0e-9, conlim=1.0e+8, etol=1.0e-6,
         window=5, red=None, M=None, N=None):
    red = red or Reductions()
    m, n = shape
    if itnlim == 0:
        itnlim = 3 * n
    dampsq = damp * damp
    itn = istop = 0
    ctol = 0.0            # lsqr.py:161-163: the reference assigns self.ctol, the local stays 0
    Anorm = Acond = 0.
    z = xnorm = xxnorm = ddnorm = res2 = 0.
    cs2 = -1.
    sn2 = 0.
    x = np.zeros(n)
    x_nrg2 = 0.0
    d_err = np.zeros(window)
    dir_errors = []
    met = _Metric(M, N)
    u, v, alpha, beta = _gk_start(A, At, rhs[:m], red, "lsqr", met)
    if alpha > 0:
        w = v.copy()
    x_is_zero = False
    Arnorm = alpha * beta
    if Arnorm == 0.0:
        x_is_zero = True
        istop = 0
    rhobar = alpha
    phibar = beta
    bnorm = beta
    rnorm = beta
    r1norm = rnorm
    r2norm = rnorm
    while itn < itnlim and not x_is_zero:
        itn += 1
        u, v, alpha_new, beta = _gk_step(A, At, u, v, alpha, red, "lsqr", met)
        if beta > 0:
            Anorm = _hyp4(Anorm, alpha, beta, damp)              # lsqr.py:262 (alpha is still the old one)
        alpha = alpha_new
        rhobar1 = _hyp(rhobar, damp)                             # lsqr.py:277-281
        cs1 = rhobar / rhobar1
        sn1 = damp / rhobar1
        psi = sn1 * phibar
        phibar = cs1 * phibar
        rho = _hyp(rhobar1, beta)                                # lsqr.py:286-293
        cs = rhobar1 / rho
        sn = beta / rho
        theta = sn * alpha
        rhobar = -cs * alpha
        phi = cs * phibar
        phibar = sn * phibar
        tau = sn * phi
        t1 = phi / rho                                           # lsqr.py:297-304
        t2 = -theta / rho
        dk = (1.0 / rho) * w
        x += t1 * w
        w *= t2
        w += v
        ddnorm = ddnorm + _sq(red.norm(dk, "lsqr.dk"))
        x_nrg2 += phi * phi                                      # lsqr.py:310-318
        d_err[itn % window] = phi
        if itn > window:
            trnc = np.linalg.norm(d_err)
            red.trace.append(float(trnc))
            x_nrg = sqrt(x_nrg2)
            dir_errors.append(trnc / x_nrg)
            if trnc < etol * x_nrg:
                istop = 8
        delta = sn2 * rho                                        # lsqr.py:324-332
        gambar = -cs2 * rho
        rhs_ = phi - delta * z
        zbar = rhs_ / gambar
        xnorm = sqrt(xxnorm + _sq(zbar))
        gamma = _hyp(gambar, theta)
        cs2 = gambar / gamma
        sn2 = theta / gamma
        z = rhs_ / gamma
        xxnorm += z * z
        Acond = Anorm * sqrt(ddnorm)                             # lsqr.py:338-342
        res1 = _sq(phibar)
        res2 = res2 + _sq(psi)
        rnorm = sqrt(res1 + res2)
        Arnorm = alpha * abs(tau)
        r1sq = _sq(rnorm) - dampsq * xxnorm                      # lsqr.py:353-356
        r1norm = sqrt(abs(r1sq))
        if r1sq < 0:
            r1norm = -r1norm
        r2norm = rnorm
        test1 = rnorm / bnorm                                    # lsqr.py:361-371
        if Anorm == 0. or rnorm == 0.:
            test2 = inf
        else:
            test2 = Arnorm / (Anorm * rnorm)
        if Acond == 0.0:
            test3 = inf
        else:
            test3 = 1.0 / Acond
        t1_ = test1 / (1 + Anorm * xnorm / bnorm)
        rtol = btol + atol * Anorm * xnorm / bnorm
        if itn >= itnlim: istop = 7                              # lsqr.py:383-392
        if 1 + test3 <= 1: istop = 6
        if 1 + test2 <= 1: istop = 5
        if 1 + t1_ <= 1: istop = 4
        if test3 <= ctol: istop = 3
        if test2 <= atol: istop = 2
        if test1 <= rtol: istop = 1
        if istop > 0:
            break
    return dict(x=x, istop=istop, itn=itn, nMatvec=2 * itn, r1norm=r1norm, r2norm=r2norm, residNorm=r2norm,
                Anorm=Anorm, Acond=Acond, Arnorm=Arnorm, xnorm=xnorm, optimal=istop in (1, 2, 4, 5, 8),
                dir_errors_window=np.array(dir_errors), trace=np.array(red.trace))


def _sign(a):
    return -1 if a < 0 else 1


def sym_ortho(a, b):
    """Stable Givens rotation (lsmr.py:500-519)."""
    if b == 0:
        return _sign(a), 0, abs(a)
    elif a == 0:
        return 0, _sign(b), abs(b)
    elif abs(b) > abs(a):
        tau = a / b
        s = _sign(b) / sqrt(1 + tau * tau)
        c = s * tau
        r = b / s
    else:
        tau = b / a
        c = _sign(a) / sqrt(1 + tau * tau)
        s = c * tau
        r = a / c
    return c, s, r


# --------------------------------------------------------------------------- #
# LSMR   -- reference pykrylov/lls/lsmr.py:64-492
# --------------------------------------------------------------------------- #
def lsmr(A, At, shape, b, damp=0.0, atol=1e-9, btol=1e-9, conlim=1e8, itnlim=None, etol=1.0e-6, window=5,
         red=None, M=None, N=None):
    red = red or Reductions()
    m, n = shape
    if itnlim is None:
        itnlim = min(m, n)
    met = _Metric(M, N)
    u, v, alpha, beta = _gk_start(A, At, b, red, "lsmr", met)
    if v is None:
        v = np.zeros(n)
    itn = 0
    zetabar = alpha * beta                                       # lsmr.py:232-238
    alphabar = alpha
    rho = 1
    rhobar = 1
    cbar = 1
    sbar = 0
    h = v.copy()
    hbar = np.zeros(n)
    x = np.zeros(n)
    betadd = beta                                                # lsmr.py:248-254
    betad = 0
    rhodold = 1
    tautildeold = 0
    thetatilde = 0
    zeta = 0
    d = 0
    normA2 = alpha * alpha                                       # lsmr.py:258-266
    maxrbar = 0
    minrbar = 1e+100
    normA = sqrt(normA2)
    condA = 1
    normx = 0
    x_nrg2 = 0
    d_err = np.zeros(window)
    dir_errors = []
    normb = beta
    istop = 0
    ctol = 0
    if conlim > 0:
        ctol = 1 / conlim
    normr = beta
    normar = alpha * beta
    if normar == 0:
        return dict(x=x, istop=istop, itn=itn, normr=normr, normar=normar, normA=normA, condA=condA, normx=normx,
                    dir_errors_window=np.array(dir_errors), trace=np.array(red.trace))
    while itn < itnlim:
        itn += 1
        u, v, alpha, beta = _gk_step(A, At, u, v, alpha, red, "lsmr", met)
        chat, shat, alphahat = sym_ortho(alphabar, damp)         # lsmr.py:338
        rhoold = rho                                             # lsmr.py:342-345
        c, s, rho = sym_ortho(alphahat, beta)
        thetanew = s * alpha
        alphabar = c * alpha
        rhobarold = rhobar                                       # lsmr.py:349-355
        zetaold = zeta
        thetabar = sbar * rho
        rhotemp = cbar * rho
        cbar, sbar, rhobar = sym_ortho(cbar * rho, thetanew)
        zeta = cbar * zetabar
        zetabar = -sbar * zetabar
        hbar = h - (thetabar * rho / (rhoold * rhobarold)) * hbar   # lsmr.py:359-361
        x = x + (zeta / (rho * rhobar)) * hbar
        h = v - (thetanew / rho) * h
        x_nrg2 += zeta * zeta                                    # lsmr.py:366-373
        d_err[itn % window] = zeta
        if itn > window:
            trnc = np.linalg.norm(d_err)
            red.trace.append(float(trnc))
            x_nrg = sqrt(x_nrg2)
            dir_errors.append(trnc / x_nrg)
            if trnc < etol * x_nrg:
                istop = 8
        betaacute = chat * betadd                                # lsmr.py:378-398
        betacheck = -shat * betadd
        betahat = c * betaacute
        betadd = -s * betaacute
        thetatildeold = thetatilde
        ctildeold, stildeold, rhotildeold = sym_ortho(rhodold, thetabar)
        thetatilde = stildeold * rhobar
        rhodold = ctildeold * rhobar
        betad = -stildeold * betad + ctildeold * betahat
        tautildeold = (zetaold - thetatildeold * tautildeold) / rhotildeold
        taud = (zeta - thetatilde * tautildeold) / rhodold
        d = d + betacheck * betacheck
        normr = sqrt(d + _sq(betad - taud) + betadd * betadd)
        normA2 = normA2 + beta * beta                            # lsmr.py:401-403
        normA = sqrt(normA2)
        normA2 = normA2 + alpha * alpha
        maxrbar = max(maxrbar, rhobarold)                        # lsmr.py:406-409
        if itn > 1:
            minrbar = min(minrbar, rhobarold)
        condA = max(maxrbar, rhotemp) / min(minrbar, rhotemp)
        normar = abs(zetabar)                                    # lsmr.py:414-425
        normx = red.norm(x, "lsmr.normx")
        test1 = normr / normb
        test2 = normar / (normA * normr)
        test3 = 1 / condA
        t1 = test1 / (1 + normA * normx / normb)
        rtol = btol + atol * normA * normx / normb
        if itn >= itnlim: istop = 7                              # lsmr.py:438-447
        if 1 + test3 <= 1: istop = 6
        if 1 + test2 <= 1: istop = 5
        if 1 + t1 <= 1: istop = 4
        if test3 <= ctol: istop = 3
        if test2 <= atol: istop = 2
        if test1 <= rtol: istop = 1
        if istop > 0:
            break
    return dict(x=x, istop=istop, itn=itn, normr=normr, normar=normar, normA=normA, condA=condA, normx=normx,
                dir_errors_window=np.array(dir_errors), trace=np.array(red.trace))


# --------------------------------------------------------------------------- #
# CRAIG   -- reference pykrylov/lls/craig.py:104-520
# --------------------------------------------------------------------------- #
def craig(A, At, shape, rhs, itnlim=0, atol=1.0e-9, btol=1.0e-9, etol=1.0e-6, window=5, red=None, M=None, N=None):
    red = red or Reductions()
    m, n = shape
    if itnlim == 0:
        itnlim = 3 * n
    itn = istop = 0
    r_nrg2 = 0.0
    x_nrg2 = 0.0
    d_err = np.zeros(window)
    dir_errors = []
    met = _Metric(M, N)
    u, v, alpha, beta = _gk_start(A, At, rhs[:m], red, "craig", met)
    if v is None:
        v = np.zeros(n)          # (the reference would fail on an unbound v for b = 0; irrelevant: the loop is skipped)
    x_is_zero = False
    if beta == 0.0:
        x_is_zero = True
        istop = 0
    bnorm = beta
    rho = _hyp(alpha, 1)                                         # craig.py:246-262
    d = u / rho
    tau = beta / rho
    r = tau * d
    rnorm = tau * tau
    c = alpha / rho
    s = 1 / rho
    zeta = s * beta
    eta = c * zeta
    xi = s * zeta
    w = c * v
    wbar = s * v
    x = zeta * w
    xnorm = eta * eta
    r1norm = xi * xi
    r2norm = rnorm
    Arnorm = 0.0
    while itn < itnlim and not x_is_zero:
        itn += 1
        Mu = met.Mu if M is not None else u                      # craig.py:302-331
        Nv = met.Nv if N is not None else v
        Mu_new = A(v) - alpha * Mu
        u_new = M(Mu_new) if M is not None else Mu_new
        beta = sqrt(red.dot(u_new, Mu_new, "craig.beta"))
        Arnorm = abs(alpha * beta * s * zeta)
        if beta > 0:
            u_new /= beta
            if M is not None:
                Mu_new /= beta
            Nv_new = At(u_new) - beta * Nv
            v_new = N(Nv_new) if N is not None else Nv_new
            alpha = sqrt(red.dot(v_new, Nv_new, "craig.alpha"))
            if alpha > 0:
                v_new /= alpha
                if N is not None:
                    Nv_new /= alpha
            v = v_new
            met.Nv = Nv_new
        met.Mu = Mu_new
        u = u_new
        beta_hat = c * beta                                      # craig.py:333-342
        gamma = s * beta
        delta = _hyp(gamma, 1)
        c2 = -1 / delta
        s2 = gamma / delta
        alpha_hat = _hyp(alpha, delta)
        c = alpha / alpha_hat
        s = delta / alpha_hat
        d = (u - beta_hat * d) / alpha_hat                       # craig.py:345-347
        tau = -beta_hat * tau / alpha_hat
        r += tau * d
        zeta = -beta_hat * zeta / alpha_hat                      # craig.py:350-352
        eta = c * zeta
        xi = s * zeta
        wbar *= s2                                               # craig.py:355-359
        w = c * v + s * wbar
        wbar *= -c
        wbar += s * v
        x += zeta * w
        r_nrg2 += tau * tau                                      # craig.py:367-375
        x_nrg2 += zeta * zeta
        d_err[itn % window] = tau
        if itn > window:
            trnc = np.linalg.norm(d_err)
            red.trace.append(float(trnc))
            r_nrg = sqrt(r_nrg2)
            dir_errors.append(trnc / r_nrg)
            if trnc < etol * r_nrg:
                istop = 8
        rnorm += tau * tau                                       # craig.py:384-392
        xnorm += eta * eta
        r1norm += xi * xi
        r2norm = rnorm
        test1 = sqrt(rnorm) / bnorm
        t1 = test1
        rtol = btol
        if itn >= itnlim: istop = 7                              # craig.py:406-413
        if 1 + t1 <= 1: istop = 4
        if test1 <= rtol: istop = 1
        if istop > 0:
            break
    return dict(x=x, r=r, istop=istop, itn=itn, nMatvec=2 * itn, r1norm=sqrt(r1norm), r2norm=sqrt(r2norm),
                Arnorm=Arnorm, xnorm=xnorm, optimal=istop in (1, 2, 4, 5, 8),
                dir_errors_d_window=np.array(dir_errors), trace=np.array(red.trace))


# --------------------------------------------------------------------------- #
# CRAIG-MR   -- reference pykrylov/lls/craigmr.py:51-241  (its per-iteration print at :190 is dropped)
# --------------------------------------------------------------------------- #
def craigmr(A, At, shape, b, itnlim=None, etol=1.0e-6, window=5, red=None, M=None, N=None):
    red = red or Reductions()
    m, n = shape
    if itnlim is None:
        itnlim = min(m, n)
    met = _Metric(M, N)
    u, v, alpha, beta = _gk_start(A, At, b, red, "craigmr", met)
    if v is None:
        v = np.zeros(n)
    itn = 0
    alpha_hat = sqrt(_sq(alpha) + 1)                             # craigmr.py:100-110
    c = alpha / alpha_hat
    s = 1. / alpha_hat
    zeta_hat = beta
    alpha_tilde = alpha_hat
    theta = 0.
    d = u / alpha_hat
    dbar = np.zeros(m)
    x = np.zeros(m)
    x_nrg2 = 0.
    d_err = np.zeros(window)
    dir_errors = []
    istop = 0
    while itn < itnlim:
        itn += 1
        u, v, alpha, beta = _gk_step(A, At, u, v, alpha, red, "craigmr", met)
        beta_hat = c * beta                                      # craigmr.py:161-170
        gamma = s * beta
        delta = sqrt(_sq(gamma) + 1)
        alpha_hat = sqrt(_sq(alpha) + _sq(delta))
        c = alpha / alpha_hat
        s = delta / alpha_hat
        rho = sqrt(_sq(alpha_tilde) + _sq(beta_hat))             # craigmr.py:173-175
        c_hat = alpha_tilde / rho
        s_hat = beta_hat / rho
        dbar = (d - theta * dbar) / rho                          # craigmr.py:178
        theta = s_hat * alpha_hat                                # craigmr.py:181-185
        alpha_tilde = -c_hat * alpha_hat
        zeta = c_hat * zeta_hat
        zeta_hat = s_hat * zeta_hat
        x_nrg2 += zeta * zeta
        d = (u - beta_hat * d) / alpha_hat                       # craigmr.py:192-194
        x += zeta * dbar
        d_err[itn % window] = zeta                               # craigmr.py:203-210
        if itn > window:
            trnc = np.linalg.norm(d_err)
            red.trace.append(float(trnc))
            x_nrg = sqrt(x_nrg2)
            dir_errors.append(trnc / x_nrg)
            if trnc < etol * x_nrg:
                istop = 8
        if itn >= itnlim: istop = 7
        if istop > 0:
            break
    return dict(x=x, istop=istop, itn=itn, nMatvec=2 * itn, optimal=istop in (1, 2, 4, 5, 8),
                dir_errors_window=np.array(dir_errors), trace=np.array(red.trace))
