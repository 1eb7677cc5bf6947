// mk_lls.hip -- LSQR, LSMR, CRAIG, CRAIG-MR, device resident; M and N may be diagonal preconditioners.
// Reference: pykrylov/lls/lsqr.py:86-453, lsmr.py:64-492, craig.py:104-520, craigmr.py:51-241.
//
// All four run on the Golub-Kahan bidiagonalisation (lsqr.py:252-271): with u, v normalised in place,
//   G1  u <- A v - alpha u          (SpMV on A, row epilogue) ; partial <u,u>
//   G2  beta = sqrt(<u,u>) ; u <- u / beta
//   G3  v <- A' u - beta v          (SpMV on the transposed CSR, row epilogue) ; partial <v,v>   [skipped if beta = 0]
//   G4  alpha = sqrt(<v,v>) ; v <- v / alpha ; the solver's scalar recurrence ; its n-space vector updates
//   G5  (CRAIG, CRAIG-MR) the m-space vector updates
// Stopping tests that need a reduction produced by G4 (LSQR: ||dk||, LSMR: ||x||) are completed by the gate
// of the next pass's G1, every workgroup evaluating them identically.
#include "mk_solver.h"

namespace {

constexpr int MAXWIN = 16;
enum { S_BETA = 0, S_BNORM = 1, S_ISTOP = 2, S_DD0 = 3, S_DD1 = 4, S_OUT = 8, S_BLK = 32, BLK = 64 };
enum { SLOT_UU = 0, SLOT_VV = 1, SLOT_X = 2 };
// output slots (scal[S_OUT + k])
enum { O_R1NORM = 0, O_R2NORM, O_ANORM, O_ACOND, O_ARNORM, O_XNORM, O_NORMR, O_NORMAR };
// state block: B_ALPHA first for every solver, the rest is solver specific (see each prologue)
enum { B_ALPHA = 0, B_ISTOP = 1, B_XNRG2 = 2, B_DERR = 3, B_S = B_DERR + MAXWIN };   // B_S .. : solver state

__device__ __forceinline__ double hyp(double a, double b) { return __dsqrt_rn(a * a + b * b); }
__device__ __forceinline__ double pmax(double a, double b) { return (b > a) ? b : a; }   // Python max(a, b)
__device__ __forceinline__ double pmin(double a, double b) { return (b < a) ? b : a; }   // Python min(a, b)

// truncated direct-error window (lsqr.py:310-318): returns trnc / nrg (NaN while itn <= window), may set istop = 8
__device__ __forceinline__ double window_test(const double *bi, double *bo, bool lead, int64_t itn, int window, double val,
                                              double nrg2, double etol, int *istop) {
    const int slot = (int)(itn % window);
    double ratio = __builtin_nan("");
    double dw[MAXWIN];                         // the whole window in one batch of loads (not one round trip each)
#pragma unroll
    for (int k = 0; k < MAXWIN; ++k) dw[k] = bi[B_DERR + k];
    if (itn > window) {
        double ss = 0.0;
#pragma unroll
        for (int k = 0; k < MAXWIN; ++k) {
            if (k < window) {
                const double e = (k == slot) ? val : dw[k];
                ss += e * e;
            }
        }
        const double trnc = __dsqrt_rn(ss), nrg = __dsqrt_rn(nrg2);
        ratio = trnc / nrg;
        if (trnc < etol * nrg) *istop = 8;
    }
    if (lead) {
#pragma unroll
        for (int k = 0; k < MAXWIN; ++k)
            if (k < window) bo[B_DERR + k] = (k == slot) ? val : dw[k];
    }
    return ratio;
}

// ------------------------------------------------------------------ bidiagonalisation kernels
// With a diagonal M (u = M(Mu), lsqr.py:253-254) the recurrence runs on Mu and u = dm * Mu is written beside it;
// without, u and Mu are one vector, as in the reference where the names alias.  Same for N, v and Nv.
struct EpiU {        // Mu <- A v - alpha Mu ; u = M(Mu) ; <u, Mu>
    static constexpr int NACC = 1, SLOT0 = SLOT_UU;
    static constexpr bool NO_MARCH = true;         // (rectangular operators: mk_device.h MkNoMarch)
    const double *blk;
    double *u;
    const double *dm;
    double *Mu;
    double alpha;
    int nt;              // u (and Mu) go past the caches (vectors beyond the Infinity Cache; mk_store_stream, mk_solver.h)
    __device__ void prologue(double *) { alpha = blk[B_ALPHA]; }
    __device__ double xin(double x) const { return x; }
    __device__ void row(int64_t i, double sum, double *acc) {
        if (dm) {
            const double t = sum - alpha * Mu[i];                 // lsqr.py:252
            mk_store_stream(Mu + i, t, nt);
            const double uu = dm[i] * t;                          // lsqr.py:254
            mk_store_stream(u + i, uu, nt);
            acc[0] += uu * t;                                     // lsqr.py:257
        } else {
            const double t = sum - alpha * u[i];                  // lsqr.py:252
            mk_store_stream(u + i, t, nt);
            acc[0] += t * t;                                      // lsqr.py:257
        }
    }
};

struct OpNormU {     // beta ; u /= beta
    static constexpr int NACC = 0, SLOT0 = 0;
    static constexpr bool NO_MARCH = true;         // (rectangular operators: mk_device.h MkNoMarch)
    const double *part;
    int np;
    double *scal;
    double *u;
    double *Mu;          // scaled as well when M is given (lsqr.py:260), else null
    double beta;
    __device__ bool prologue(double *s4, bool lead) {
        beta = __dsqrt_rn(mk_total(part + SLOT_UU * MK_MAXP, np, s4));
        if (lead) scal[S_BETA] = beta;
        return false;
    }
    __device__ bool skip() const { return !(beta > 0); }          // lsqr.py:258
    __device__ void pair(int64_t i, double *) {
        double2 v = mk_ld2(u, i);
        v.x = v.x / beta;                                         // lsqr.py:259
        v.y = v.y / beta;
        mk_st2(u, i, v);
        if (Mu) {
            double2 w = mk_ld2(Mu, i);
            w.x = w.x / beta;                                     // lsqr.py:260
            w.y = w.y / beta;
            mk_st2(Mu, i, w);
        }
    }
    __device__ void one(int64_t i, double *) {
        u[i] = u[i] / beta;
        if (Mu) Mu[i] = Mu[i] / beta;
    }
};

struct GateV {       // the A' product happens only if beta > 0 (lsqr.py:258)
    const double *scal;
    __device__ bool open(double *, bool, bool *) { return scal[S_BETA] > 0; }
};

struct EpiV {        // Nv <- A' u - beta Nv ; v = N(Nv) ; <v, Nv>
    static constexpr int NACC = 1, SLOT0 = SLOT_VV;
    static constexpr bool NO_MARCH = true;         // (rectangular operators: mk_device.h MkNoMarch)
    const double *scal;
    double *v;
    const double *dn;
    double *Nv;
    double beta;
    __device__ void prologue(double *) { beta = scal[S_BETA]; }
    __device__ double xin(double x) const { return x; }
    __device__ void row(int64_t j, double sum, double *acc) {
        if (dn) {
            const double t = sum - beta * Nv[j];                  // lsqr.py:264
            Nv[j] = t;
            const double vv = dn[j] * t;                          // lsqr.py:266
            v[j] = vv;
            acc[0] += vv * t;                                     // lsqr.py:269
        } else {
            const double t = sum - beta * v[j];                   // lsqr.py:264
            v[j] = t;
            acc[0] += t * t;                                      // lsqr.py:269
        }
    }
};

// Row-block partition (mk_csr_set_row_block): A' u arrives as t = sum over ranks of the local blocks' products; this
// is EpiV's row step on it, run identically by every rank (v is replicated)
struct OpVt {        // Nv <- t - beta Nv ; v = N(Nv) ; <v, Nv>
    static constexpr int NACC = 1, SLOT0 = SLOT_VV;
    static constexpr bool NO_MARCH = true;         // (rectangular operators: mk_device.h MkNoMarch)
    const double *scal;
    const double *t;
    double *v;
    const double *dn;
    double *Nv;
    double beta;
    __device__ bool prologue(double *, bool) {
        beta = scal[S_BETA];
        return false;
    }
    __device__ bool skip() const { return !(beta > 0); }          // lsqr.py:258 (the gate of the fused product)
    __device__ void one(int64_t j, double *acc) {
        if (dn) {
            const double w = t[j] - beta * Nv[j];                 // lsqr.py:264
            Nv[j] = w;
            const double vv = dn[j] * w;                          // lsqr.py:266
            v[j] = vv;
            acc[0] += vv * w;                                     // lsqr.py:269
        } else {
            const double w = t[j] - beta * v[j];                  // lsqr.py:264
            v[j] = w;
            acc[0] += w * w;                                      // lsqr.py:269
        }
    }
    __device__ void pair(int64_t j, double *acc) {
        one(j, acc);
        one(j + 1, acc);
    }
};

struct OpScaleNv {   // Nv /= alpha where the solver's G4 did v /= alpha (lsqr.py:272): only with a preconditioner N
    static constexpr int NACC = 0, SLOT0 = 0;
    static constexpr bool NO_MARCH = true;         // (rectangular operators: mk_device.h MkNoMarch)
    const double *scal;
    const double *blk;   // the state block that holds the new alpha
    int need_beta;       // loop: the normalisation happens inside `if beta > 0` (lsqr.py:258-272)
    double *Nv;
    double alpha;
    bool on;
    __device__ bool prologue(double *, bool) {
        alpha = blk[B_ALPHA];
        on = (alpha > 0) && (!need_beta || scal[S_BETA] > 0);
        return false;
    }
    __device__ bool skip() const { return !on; }
    __device__ void pair(int64_t i, double *) {
        double2 w = mk_ld2(Nv, i);
        w.x = w.x / alpha;
        w.y = w.y / alpha;
        mk_st2(Nv, i, w);
    }
    __device__ void one(int64_t i, double *) { Nv[i] = Nv[i] / alpha; }
};

// new alpha exactly as the reference leaves it: unchanged when beta == 0
__device__ __forceinline__ double next_alpha(const double *part, int np, double beta, double alpha_old, double *s4) {
    const double vv = mk_total(part + SLOT_VV * MK_MAXP, np, s4);
    return (beta > 0) ? __dsqrt_rn(vv) : alpha_old;
}

// ================================================================== LSQR
namespace lsqr {
enum { RHOBAR = B_S, PHIBAR, ANORM, Z, XXNORM, RES2, CS2, SN2, TEST1, TEST2, T1, RTOL, R1NORM, R2NORM, ARNORM, XNORM };

struct Gate {        // finishes the previous pass: ddnorm, Acond, test3, stopping rules (lsqr.py:338-398)
    const double *part;
    int np;
    double *scal;
    MkStatus *st;
    int64_t it, itnlim;
    double atol;
    __device__ bool open(double *s4, bool lead, bool *stop) {
        if (it == 0) {
            if (lead) st->itn = 1;
            return true;
        }
        const int par = (int)(it & 1);
        const double *b = scal + S_BLK + par * BLK;
        const double nk = __dsqrt_rn(mk_total(part + SLOT_X * MK_MAXP, np, s4));   // norm(dk), lsqr.py:303
        const double ddnorm = scal[S_DD0 + par] + nk * nk;
        const double Acond = b[ANORM] * __dsqrt_rn(ddnorm);                         // lsqr.py:338
        const double test3 = (Acond == 0.0) ? __builtin_inf() : 1.0 / Acond;
        int istop = (int)b[B_ISTOP];
        if (it >= itnlim) istop = 7;                                                // lsqr.py:383-392 (itn == it here)
        if (1 + test3 <= 1) istop = 6;
        if (1 + b[TEST2] <= 1) istop = 5;
        if (1 + b[T1] <= 1) istop = 4;
        if (test3 <= 0.0) istop = 3;                                                // ctol stays 0 (lsqr.py:161-163)
        if (b[TEST2] <= atol) istop = 2;
        if (b[TEST1] <= b[RTOL]) istop = 1;
        const bool fin = (istop > 0) || (it >= itnlim);
        if (lead) {
            scal[S_DD0 + (par ^ 1)] = ddnorm;
            scal[S_OUT + O_ACOND] = Acond;
            scal[S_ISTOP] = (double)istop;
            if (!fin) st->itn = it + 1;
        }
        if (fin) {
            *stop = true;
            return false;
        }
        return true;
    }
};

struct OpN {         // G4
    static constexpr int NACC = 1, SLOT0 = SLOT_X;
    static constexpr bool NO_MARCH = true;         // (rectangular operators: mk_device.h MkNoMarch)
    const double *part;
    int np;
    double *scal;
    MkStatus *st;
    double *hist;
    int par;
    int64_t itn;
    int window;
    double damp, atol, btol, etol;
    double *v, *w, *x;
    double alpha, t1, t2, cdk;
    bool norm_v;
    __device__ bool prologue(double *s4, bool lead) {
        const double *bi = scal + S_BLK + par * BLK;
        double *bo = scal + S_BLK + (par ^ 1) * BLK;
        const double beta = scal[S_BETA], bnorm = scal[S_BNORM];
        const double alpha_old = bi[B_ALPHA];
        double Anorm = bi[ANORM];
        if (beta > 0) Anorm = __dsqrt_rn(Anorm * Anorm + alpha_old * alpha_old + beta * beta + damp * damp);   // lsqr.py:262
        alpha = next_alpha(part, np, beta, alpha_old, s4);
        norm_v = (beta > 0) && (alpha > 0);
        const double rhobar1 = hyp(bi[RHOBAR], damp);                                // lsqr.py:277-281
        const double cs1 = bi[RHOBAR] / rhobar1, sn1 = damp / rhobar1;
        const double psi = sn1 * bi[PHIBAR];
        double phibar = cs1 * bi[PHIBAR];
        const double rho = hyp(rhobar1, beta);                                       // lsqr.py:286-293
        const double cs = rhobar1 / rho, sn = beta / rho;
        const double theta = sn * alpha;
        const double rhobar = -cs * alpha;
        const double phi = cs * phibar;
        phibar = sn * phibar;
        const double tau = sn * phi;
        t1 = phi / rho;                                                              // lsqr.py:297-299
        t2 = -theta / rho;
        cdk = 1.0 / rho;
        int istop = 0;
        const double xnrg2 = bi[B_XNRG2] + phi * phi;                                // lsqr.py:310-318
        const double ratio = window_test(bi, bo, lead, itn, window, phi, xnrg2, etol, &istop);
        const double delta = bi[SN2] * rho;                                          // lsqr.py:324-332
        const double gambar = -bi[CS2] * rho;
        const double rhs = phi - delta * bi[Z];
        const double zbar = rhs / gambar;
        const double xnorm = __dsqrt_rn(bi[XXNORM] + zbar * zbar);
        const double gamma = hyp(gambar, theta);
        const double cs2 = gambar / gamma, sn2 = theta / gamma;
        const double z = rhs / gamma;
        const double xxnorm = bi[XXNORM] + z * z;
        const double res1 = phibar * phibar;                                         // lsqr.py:339-342
        const double res2 = bi[RES2] + psi * psi;
        const double rnorm = __dsqrt_rn(res1 + res2);
        const double Arnorm = alpha * fabs(tau);
        const double r1sq = rnorm * rnorm - damp * damp * xxnorm;                    // lsqr.py:353-356
        double r1norm = __dsqrt_rn(fabs(r1sq));
        if (r1sq < 0) r1norm = -r1norm;
        const double test1 = rnorm / bnorm;                                          // lsqr.py:361-371
        const double test2 = (Anorm == 0. || rnorm == 0.) ? __builtin_inf() : Arnorm / (Anorm * rnorm);
        const double t1_ = test1 / (1 + Anorm * xnorm / bnorm);
        const double rtol = btol + atol * Anorm * xnorm / bnorm;
        if (lead) {
            bo[B_ALPHA] = alpha;
            bo[B_ISTOP] = (double)istop;
            bo[B_XNRG2] = xnrg2;
            bo[RHOBAR] = rhobar;
            bo[PHIBAR] = phibar;
            bo[ANORM] = Anorm;
            bo[Z] = z;
            bo[XXNORM] = xxnorm;
            bo[RES2] = res2;
            bo[CS2] = cs2;
            bo[SN2] = sn2;
            bo[TEST1] = test1;
            bo[TEST2] = test2;
            bo[T1] = t1_;
            bo[RTOL] = rtol;
            scal[S_OUT + O_R1NORM] = r1norm;
            scal[S_OUT + O_R2NORM] = rnorm;
            scal[S_OUT + O_ANORM] = Anorm;
            scal[S_OUT + O_ARNORM] = Arnorm;
            scal[S_OUT + O_XNORM] = xnorm;
            const int64_t h = st->hist_len % MK_HIST_RING;
            hist[h] = rnorm;
            hist[MK_HIST_RING + h] = ratio;
            st->hist_len += 1;
            st->nMatvec = 2 * itn;
        }
        return false;
    }
    __device__ bool skip() const { return false; }
    __device__ void elem(double &vv, double &wv, double &xv, double *acc) {
        if (norm_v) vv = vv / alpha;                                                 // lsqr.py:271
        const double dk = cdk * wv;                                                  // lsqr.py:299
        acc[0] += dk * dk;                                                           // lsqr.py:303
        xv = xv + t1 * wv;                                                           // lsqr.py:301
        wv = wv * t2;                                                                // lsqr.py:302
        wv = wv + vv;
    }
    __device__ void pair(int64_t i, double *acc) {
        double2 vv = mk_ld2(v, i), wv = mk_ld2(w, i), xv = mk_ld2(x, i);
        elem(vv.x, wv.x, xv.x, acc);
        elem(vv.y, wv.y, xv.y, acc);
        if (norm_v) mk_st2(v, i, vv);
        mk_st2(w, i, wv);
        mk_st2(x, i, xv);
    }
    __device__ void one(int64_t i, double *acc) {
        double vv = v[i], wv = w[i], xv = x[i];
        elem(vv, wv, xv, acc);
        if (norm_v) v[i] = vv;
        w[i] = wv;
        x[i] = xv;
    }
};
}  // namespace lsqr

// ================================================================== LSMR
namespace lsmr {
enum { ALPHABAR = B_S, RHO, RHOBAR, CBAR, SBAR, ZETA, ZETABAR, BETADD, BETAD, RHODOLD, TAUTILDEOLD, THETATILDE, D,
       NORMA2, MAXRBAR, MINRBAR, NORMR, NORMAR, NORMA, CONDA };

__device__ __forceinline__ void sym_ortho(double a, double b, double *c, double *s, double *r) {   // lsmr.py:500-519
    if (b == 0) {
        *c = (a < 0) ? -1.0 : 1.0;
        *s = 0.0;
        *r = fabs(a);
    } else if (a == 0) {
        *c = 0.0;
        *s = (b < 0) ? -1.0 : 1.0;
        *r = fabs(b);
    } else if (fabs(b) > fabs(a)) {
        const double tau = a / b;
        *s = ((b < 0) ? -1.0 : 1.0) / __dsqrt_rn(1 + tau * tau);
        *c = *s * tau;
        *r = b / *s;
    } else {
        const double tau = b / a;
        *c = ((a < 0) ? -1.0 : 1.0) / __dsqrt_rn(1 + tau * tau);
        *s = *c * tau;
        *r = a / *c;
    }
}

struct Gate {        // finishes the previous pass: normx and the stopping rules (lsmr.py:414-447)
    const double *part;
    int np;
    double *scal;
    MkStatus *st;
    int64_t it, itnlim;
    double atol, btol, ctol;
    __device__ bool open(double *s4, bool lead, bool *stop) {
        if (it == 0) {
            if (lead) st->itn = 1;
            return true;
        }
        const double *b = scal + S_BLK + (int)(it & 1) * BLK;
        const double normb = scal[S_BNORM];
        const double normx = __dsqrt_rn(mk_total(part + SLOT_X * MK_MAXP, np, s4));    // lsmr.py:415
        const double normr = b[NORMR], normar = b[NORMAR], normA = b[NORMA], condA = b[CONDA];
        const double test1 = normr / normb;
        const double test2 = normar / (normA * normr);
        const double test3 = 1 / condA;
        const double t1 = test1 / (1 + normA * normx / normb);
        const double rtol = btol + atol * normA * normx / normb;
        int istop = (int)b[B_ISTOP];
        if (it >= itnlim) istop = 7;                                                  // lsmr.py:438-447
        if (1 + test3 <= 1) istop = 6;
        if (1 + test2 <= 1) istop = 5;
        if (1 + t1 <= 1) istop = 4;
        if (test3 <= ctol) istop = 3;
        if (test2 <= atol) istop = 2;
        if (test1 <= rtol) istop = 1;
        const bool fin = (istop > 0) || (it >= itnlim);
        if (lead) {
            scal[S_OUT + O_XNORM] = normx;
            scal[S_ISTOP] = (double)istop;
            if (!fin) st->itn = it + 1;
        }
        if (fin) {
            *stop = true;
            return false;
        }
        return true;
    }
};

struct OpN {
    static constexpr int NACC = 1, SLOT0 = SLOT_X;
    static constexpr bool NO_MARCH = true;         // (rectangular operators: mk_device.h MkNoMarch)
    const double *part;
    int np;
    double *scal;
    MkStatus *st;
    double *hist;
    int par;
    int64_t itn;
    int window;
    double damp, etol;
    double *v, *h, *hbar, *x;
    double alpha, c_hbar, c_x, c_h;
    bool norm_v;
    __device__ bool prologue(double *s4, bool lead) {
        const double *bi = scal + S_BLK + par * BLK;
        double *bo = scal + S_BLK + (par ^ 1) * BLK;
        const double beta = scal[S_BETA];
        alpha = next_alpha(part, np, beta, bi[B_ALPHA], s4);
        norm_v = (beta > 0) && (alpha > 0);
        double chat, shat, alphahat, c, s, rho;
        sym_ortho(bi[ALPHABAR], damp, &chat, &shat, &alphahat);                       // lsmr.py:338
        const double rhoold = bi[RHO];                                                // lsmr.py:342-345
        sym_ortho(alphahat, beta, &c, &s, &rho);
        const double thetanew = s * alpha;
        const double alphabar = c * alpha;
        const double rhobarold = bi[RHOBAR];                                          // lsmr.py:349-355
        const double zetaold = bi[ZETA];
        const double thetabar = bi[SBAR] * rho;
        const double rhotemp = bi[CBAR] * rho;
        double cbar, sbar, rhobar;
        sym_ortho(bi[CBAR] * rho, thetanew, &cbar, &sbar, &rhobar);
        const double zeta = cbar * bi[ZETABAR];
        const double zetabar = -sbar * bi[ZETABAR];
        c_hbar = thetabar * rho / (rhoold * rhobarold);                               // lsmr.py:359-361
        c_x = zeta / (rho * rhobar);
        c_h = thetanew / rho;
        int istop = 0;
        const double xnrg2 = bi[B_XNRG2] + zeta * zeta;                               // lsmr.py:366-373
        const double ratio = window_test(bi, bo, lead, itn, window, zeta, xnrg2, etol, &istop);
        const double betaacute = chat * bi[BETADD];                                   // lsmr.py:378-398
        const double betacheck = -shat * bi[BETADD];
        const double betahat = c * betaacute;
        const double betadd = -s * betaacute;
        const double thetatildeold = bi[THETATILDE];
        double ctildeold, stildeold, rhotildeold;
        sym_ortho(bi[RHODOLD], thetabar, &ctildeold, &stildeold, &rhotildeold);
        const double thetatilde = stildeold * rhobar;
        const double rhodold = ctildeold * rhobar;
        const double betad = -stildeold * bi[BETAD] + ctildeold * betahat;
        const double tautildeold = (zetaold - thetatildeold * bi[TAUTILDEOLD]) / rhotildeold;
        const double taud = (zeta - thetatilde * tautildeold) / rhodold;
        const double d = bi[D] + betacheck * betacheck;
        const double bt = betad - taud;
        const double normr = __dsqrt_rn(d + bt * bt + betadd * betadd);
        double normA2 = bi[NORMA2] + beta * beta;                                     // lsmr.py:401-403
        const double normA = __dsqrt_rn(normA2);
        normA2 = normA2 + alpha * alpha;
        const double maxrbar = pmax(bi[MAXRBAR], rhobarold);                          // lsmr.py:406-409
        double minrbar = bi[MINRBAR];
        if (itn > 1) minrbar = pmin(minrbar, rhobarold);
        const double condA = pmax(maxrbar, rhotemp) / pmin(minrbar, rhotemp);
        const double normar = fabs(zetabar);                                          // lsmr.py:414
        if (lead) {
            bo[B_ALPHA] = alpha;
            bo[B_ISTOP] = (double)istop;
            bo[B_XNRG2] = xnrg2;
            bo[ALPHABAR] = alphabar;
            bo[RHO] = rho;
            bo[RHOBAR] = rhobar;
            bo[CBAR] = cbar;
            bo[SBAR] = sbar;
            bo[ZETA] = zeta;
            bo[ZETABAR] = zetabar;
            bo[BETADD] = betadd;
            bo[BETAD] = betad;
            bo[RHODOLD] = rhodold;
            bo[TAUTILDEOLD] = tautildeold;
            bo[THETATILDE] = thetatilde;
            bo[D] = d;
            bo[NORMA2] = normA2;
            bo[MAXRBAR] = maxrbar;
            bo[MINRBAR] = minrbar;
            bo[NORMR] = normr;
            bo[NORMAR] = normar;
            bo[NORMA] = normA;
            bo[CONDA] = condA;
            scal[S_OUT + O_NORMR] = normr;
            scal[S_OUT + O_NORMAR] = normar;
            scal[S_OUT + O_ANORM] = normA;
            scal[S_OUT + O_ACOND] = condA;
            const int64_t hh = st->hist_len % MK_HIST_RING;
            hist[hh] = normr;
            hist[MK_HIST_RING + hh] = ratio;
            st->hist_len += 1;
            st->nMatvec = 2 * itn;
        }
        return false;
    }
    __device__ bool skip() const { return false; }
    __device__ void elem(double &vv, double &hv, double &hb, double &xv, double *acc) {
        if (norm_v) vv = vv / alpha;                                                  // lsmr.py:332
        hb = hv - c_hbar * hb;                                                        // lsmr.py:359
        xv = xv + c_x * hb;                                                           // lsmr.py:360
        hv = vv - c_h * hv;                                                           // lsmr.py:361
        acc[0] += xv * xv;                                                            // lsmr.py:415
    }
    __device__ void pair(int64_t i, double *acc) {
        double2 vv = mk_ld2(v, i), hv = mk_ld2(h, i), hb = mk_ld2(hbar, i), xv = mk_ld2(x, i);
        elem(vv.x, hv.x, hb.x, xv.x, acc);
        elem(vv.y, hv.y, hb.y, xv.y, acc);
        if (norm_v) mk_st2(v, i, vv);
        mk_st2(h, i, hv);
        mk_st2(hbar, i, hb);
        mk_st2(x, i, xv);
    }
    __device__ void one(int64_t i, double *acc) {
        double vv = v[i], hv = h[i], hb = hbar[i], xv = x[i];
        elem(vv, hv, hb, xv, acc);
        if (norm_v) v[i] = vv;
        h[i] = hv;
        hbar[i] = hb;
        x[i] = xv;
    }
};
}  // namespace lsmr

// ================================================================== CRAIG and CRAIG-MR
namespace craig {
// persistent state, then per-pass temporaries handed from G4 to G5
enum { C = B_S, S, TAU, ZETA, RNORM, XNORM, R1NORM, RNRG2, ARNORM, T_BETAHAT, T_ALPHAHAT, T_TAU, T_STOP,
       // CRAIG-MR
       ZETA_HAT, ALPHA_TILDE, THETA, T_RHO, T_THETA_OLD, T_ZETA };

struct CountGate {   // loop header `while itn < itnlim` (craig.py:296, craigmr.py:128)
    MkStatus *st;
    int64_t it, itnlim;
    __device__ bool open(double *, bool lead, bool *stop) {
        if (it >= itnlim) {
            *stop = true;
            return false;
        }
        if (lead) st->itn = it + 1;
        return true;
    }
};

struct OpN {         // CRAIG G4: alpha, rotations, w / wbar / x, stopping tests
    static constexpr int NACC = 0, SLOT0 = 0;
    static constexpr bool NO_MARCH = true;         // (rectangular operators: mk_device.h MkNoMarch)
    const double *part;
    int np;
    double *scal;
    MkStatus *st;
    double *hist;
    int par;
    int64_t itn, itnlim;
    int window;
    double btol, etol;
    double *v, *w, *wbar, *x;
    double alpha, c, s, s2, zeta;
    bool norm_v;
    __device__ bool prologue(double *s4, bool lead) {
        const double *bi = scal + S_BLK + par * BLK;
        double *bo = scal + S_BLK + (par ^ 1) * BLK;
        const double beta = scal[S_BETA], bnorm = scal[S_BNORM];
        const double alpha_old = bi[B_ALPHA];
        const double Arnorm = fabs(alpha_old * beta * bi[S] * bi[ZETA]);              // craig.py:312
        alpha = next_alpha(part, np, beta, alpha_old, s4);
        norm_v = (beta > 0) && (alpha > 0);
        const double beta_hat = bi[C] * beta;                                         // craig.py:333-342
        const double gamma = bi[S] * beta;
        const double delta = hyp(gamma, 1);
        s2 = gamma / delta;
        const double alpha_hat = hyp(alpha, delta);
        c = alpha / alpha_hat;
        s = delta / alpha_hat;
        const double tau = -beta_hat * bi[TAU] / alpha_hat;                           // craig.py:346
        zeta = -beta_hat * bi[ZETA] / alpha_hat;                                      // craig.py:350-352
        const double eta = c * zeta, xi = s * zeta;
        int istop = 0;
        const double rnrg2 = bi[RNRG2] + tau * tau;                                   // craig.py:367-375
        const double xnrg2 = bi[B_XNRG2] + zeta * zeta;
        const double ratio = window_test(bi, bo, lead, itn, window, tau, rnrg2, etol, &istop);
        const double rnorm = bi[RNORM] + tau * tau;                                   // craig.py:384-392
        const double xnorm = bi[XNORM] + eta * eta;
        const double r1norm = bi[R1NORM] + xi * xi;
        const double test1 = __dsqrt_rn(rnorm) / bnorm;
        if (itn >= itnlim) istop = 7;                                                 // craig.py:406-413
        if (1 + test1 <= 1) istop = 4;
        if (test1 <= btol) istop = 1;
        if (lead) {
            bo[B_ALPHA] = alpha;
            bo[B_ISTOP] = (double)istop;
            bo[B_XNRG2] = xnrg2;
            bo[C] = c;
            bo[S] = s;
            bo[TAU] = tau;
            bo[ZETA] = zeta;
            bo[RNORM] = rnorm;
            bo[XNORM] = xnorm;
            bo[R1NORM] = r1norm;
            bo[RNRG2] = rnrg2;
            bo[ARNORM] = Arnorm;
            bo[T_BETAHAT] = beta_hat;
            bo[T_ALPHAHAT] = alpha_hat;
            bo[T_TAU] = tau;
            bo[T_STOP] = (istop > 0) ? 1.0 : 0.0;
            scal[S_ISTOP] = (double)istop;
            scal[S_OUT + O_R1NORM] = __dsqrt_rn(r1norm);
            scal[S_OUT + O_R2NORM] = __dsqrt_rn(rnorm);
            scal[S_OUT + O_ARNORM] = Arnorm;
            scal[S_OUT + O_XNORM] = xnorm;
            const int64_t hh = st->hist_len % MK_HIST_RING;
            hist[hh] = rnorm;
            hist[MK_HIST_RING + hh] = ratio;
            st->hist_len += 1;
            st->nMatvec = 2 * itn;
        }
        return false;
    }
    __device__ bool skip() const { return false; }
    __device__ void elem(double &vv, double &wv, double &wb, double &xv) {
        if (norm_v) vv = vv / alpha;                                                  // craig.py:329
        wb = wb * s2;                                                                 // craig.py:355
        wv = c * vv + s * wb;                                                         // craig.py:356
        wb = wb * (-c);                                                               // craig.py:357
        wb = wb + s * vv;                                                             // craig.py:358
        xv = xv + zeta * wv;                                                          // craig.py:359
    }
    __device__ void pair(int64_t i, double *) {
        double2 vv = mk_ld2(v, i), wv = mk_ld2(w, i), wb = mk_ld2(wbar, i), xv = mk_ld2(x, i);
        elem(vv.x, wv.x, wb.x, xv.x);
        elem(vv.y, wv.y, wb.y, xv.y);
        if (norm_v) mk_st2(v, i, vv);
        mk_st2(w, i, wv);
        mk_st2(wbar, i, wb);
        mk_st2(x, i, xv);
    }
    __device__ void one(int64_t i, double *) {
        double vv = v[i], wv = w[i], wb = wbar[i], xv = x[i];
        elem(vv, wv, wb, xv);
        if (norm_v) v[i] = vv;
        w[i] = wv;
        wbar[i] = wb;
        x[i] = xv;
    }
};

struct OpM {         // CRAIG G5: d = (u - beta_hat d) / alpha_hat ; r += tau d ; halts the loop if G4 said so
    static constexpr int NACC = 0, SLOT0 = 0;
    static constexpr bool NO_MARCH = true;         // (rectangular operators: mk_device.h MkNoMarch)
    const double *blk;      // block written by this pass's G4
    const double *u;
    double *d, *r;
    double beta_hat, alpha_hat, tau;
    __device__ bool prologue(double *, bool) {
        beta_hat = blk[T_BETAHAT];
        alpha_hat = blk[T_ALPHAHAT];
        tau = blk[T_TAU];
        return blk[T_STOP] != 0.0;
    }
    __device__ bool skip() const { return false; }
    __device__ void elem(double uv, double &dv, double &rv) {
        dv = (uv - beta_hat * dv) / alpha_hat;                                        // craig.py:345
        rv = rv + tau * dv;                                                           // craig.py:347
    }
    __device__ void pair(int64_t i, double *) {
        const double2 uv = mk_ld2(u, i);
        double2 dv = mk_ld2(d, i), rv = mk_ld2(r, i);
        elem(uv.x, dv.x, rv.x);
        elem(uv.y, dv.y, rv.y);
        mk_st2(d, i, dv);
        mk_st2(r, i, rv);
    }
    __device__ void one(int64_t i, double *) {
        double dv = d[i], rv = r[i];
        elem(u[i], dv, rv);
        d[i] = dv;
        r[i] = rv;
    }
};

struct OpNmr {       // CRAIG-MR G4: alpha, v normalisation, all scalars (no n-space vectors besides v)
    static constexpr int NACC = 0, SLOT0 = 0;
    static constexpr bool NO_MARCH = true;         // (rectangular operators: mk_device.h MkNoMarch)
    const double *part;
    int np;
    double *scal;
    MkStatus *st;
    double *hist;
    int par;
    int64_t itn, itnlim;
    int window;
    double etol;
    double *v;
    double alpha;
    bool norm_v;
    __device__ bool prologue(double *s4, bool lead) {
        const double *bi = scal + S_BLK + par * BLK;
        double *bo = scal + S_BLK + (par ^ 1) * BLK;
        const double beta = scal[S_BETA];
        alpha = next_alpha(part, np, beta, bi[B_ALPHA], s4);
        norm_v = (beta > 0) && (alpha > 0);
        const double beta_hat = bi[C] * beta;                                         // craigmr.py:161-170
        const double gamma = bi[S] * beta;
        const double delta = __dsqrt_rn(gamma * gamma + 1);
        const double alpha_hat = __dsqrt_rn(alpha * alpha + delta * delta);
        const double c = alpha / alpha_hat, s = delta / alpha_hat;
        const double at = bi[ALPHA_TILDE];
        const double rho = __dsqrt_rn(at * at + beta_hat * beta_hat);                  // craigmr.py:173-175
        const double c_hat = at / rho, s_hat = beta_hat / rho;
        const double theta = s_hat * alpha_hat;                                       // craigmr.py:181-185
        const double alpha_tilde = -c_hat * alpha_hat;
        const double zeta = c_hat * bi[ZETA_HAT];
        const double zeta_hat = s_hat * bi[ZETA_HAT];
        int istop = 0;
        const double xnrg2 = bi[B_XNRG2] + zeta * zeta;
        const double ratio = window_test(bi, bo, lead, itn, window, zeta, xnrg2, etol, &istop);
        if (itn >= itnlim) istop = 7;                                                 // craigmr.py:212
        if (lead) {
            bo[B_ALPHA] = alpha;
            bo[B_ISTOP] = (double)istop;
            bo[B_XNRG2] = xnrg2;
            bo[C] = c;
            bo[S] = s;
            bo[ZETA_HAT] = zeta_hat;
            bo[ALPHA_TILDE] = alpha_tilde;
            bo[THETA] = theta;
            bo[T_BETAHAT] = beta_hat;
            bo[T_ALPHAHAT] = alpha_hat;
            bo[T_RHO] = rho;
            bo[T_THETA_OLD] = bi[THETA];
            bo[T_ZETA] = zeta;
            bo[T_STOP] = (istop > 0) ? 1.0 : 0.0;
            scal[S_ISTOP] = (double)istop;
            const int64_t hh = st->hist_len % MK_HIST_RING;
            hist[hh] = xnrg2;
            hist[MK_HIST_RING + hh] = ratio;
            st->hist_len += 1;
            st->nMatvec = 2 * itn;
        }
        return false;
    }
    __device__ bool skip() const { return !norm_v; }
    __device__ void pair(int64_t i, double *) {
        double2 vv = mk_ld2(v, i);
        vv.x = vv.x / alpha;                                                          // craigmr.py:157
        vv.y = vv.y / alpha;
        mk_st2(v, i, vv);
    }
    __device__ void one(int64_t i, double *) { v[i] = v[i] / alpha; }
};

struct OpMmr {       // CRAIG-MR G5: dbar, d, x (all of length m)
    static constexpr int NACC = 0, SLOT0 = 0;
    static constexpr bool NO_MARCH = true;         // (rectangular operators: mk_device.h MkNoMarch)
    const double *blk;
    const double *u;
    double *d, *dbar, *x;
    double beta_hat, alpha_hat, rho, theta_old, zeta;
    __device__ bool prologue(double *, bool) {
        beta_hat = blk[T_BETAHAT];
        alpha_hat = blk[T_ALPHAHAT];
        rho = blk[T_RHO];
        theta_old = blk[T_THETA_OLD];
        zeta = blk[T_ZETA];
        return blk[T_STOP] != 0.0;
    }
    __device__ bool skip() const { return false; }
    __device__ void elem(double uv, double &dv, double &db, double &xv) {
        db = (dv - theta_old * db) / rho;                                             // craigmr.py:178
        dv = (uv - beta_hat * dv) / alpha_hat;                                        // craigmr.py:192
        xv = xv + zeta * db;                                                          // craigmr.py:194
    }
    __device__ void pair(int64_t i, double *) {
        const double2 uv = mk_ld2(u, i);
        double2 dv = mk_ld2(d, i), db = mk_ld2(dbar, i), xv = mk_ld2(x, i);
        elem(uv.x, dv.x, db.x, xv.x);
        elem(uv.y, dv.y, db.y, xv.y);
        mk_st2(d, i, dv);
        mk_st2(dbar, i, db);
        mk_st2(x, i, xv);
    }
    __device__ void one(int64_t i, double *) {
        double dv = d[i], db = dbar[i], xv = x[i];
        elem(u[i], dv, db, xv);
        d[i] = dv;
        dbar[i] = db;
        x[i] = xv;
    }
};
}  // namespace craig

// ================================================================== setup kernels
// scalar initialisation of all four solvers (lsqr.py:211-226, lsmr.py:232-283, craig.py:236-262, craigmr.py:98-112)
__global__ __launch_bounds__(MK_BLOCK) void lls_init_kernel(const double *part, int np, double *scal, MkStatus *st,
                                                            MkHalt halt, int kind, int64_t itnlim) {
    __shared__ double s4[4];
    const double beta = scal[S_BETA];
    const double alpha = (beta > 0) ? __dsqrt_rn(mk_total(part + SLOT_VV * MK_MAXP, np, s4)) : 0.0;
    if (threadIdx.x != 0) return;
    bool stop = false;
    int istop = 0;
    scal[S_BNORM] = beta;
    scal[S_DD0] = scal[S_DD1] = 0.0;
    for (int p = 0; p < 2; ++p) {
        double *b = scal + S_BLK + p * BLK;
        for (int k = 0; k < BLK; ++k) b[k] = 0.0;
        b[B_ALPHA] = alpha;
        if (kind == MK_LSQR) {
            b[lsqr::RHOBAR] = alpha;
            b[lsqr::PHIBAR] = beta;
            b[lsqr::CS2] = -1.0;
        } else if (kind == MK_LSMR) {
            b[lsmr::ALPHABAR] = alpha;
            b[lsmr::RHO] = 1;
            b[lsmr::RHOBAR] = 1;
            b[lsmr::CBAR] = 1;
            b[lsmr::ZETABAR] = alpha * beta;
            b[lsmr::BETADD] = beta;
            b[lsmr::RHODOLD] = 1;
            b[lsmr::NORMA2] = alpha * alpha;
            b[lsmr::MINRBAR] = 1e+100;
        } else if (kind == MK_CRAIG) {
            const double rho = hyp(alpha, 1);
            const double tau = beta / rho, c = alpha / rho, s = 1 / rho;
            const double zeta = s * beta, eta = c * zeta, xi = s * zeta;
            b[craig::C] = c;
            b[craig::S] = s;
            b[craig::TAU] = tau;
            b[craig::ZETA] = zeta;
            b[craig::RNORM] = tau * tau;
            b[craig::XNORM] = eta * eta;
            b[craig::R1NORM] = xi * xi;
        } else {
            const double alpha_hat = __dsqrt_rn(alpha * alpha + 1);
            b[craig::C] = alpha / alpha_hat;
            b[craig::S] = 1. / alpha_hat;
            b[craig::ZETA_HAT] = beta;
            b[craig::ALPHA_TILDE] = alpha_hat;
        }
    }
    if (kind == MK_LSQR) {
        scal[S_OUT + O_R1NORM] = scal[S_OUT + O_R2NORM] = beta;
        scal[S_OUT + O_ARNORM] = alpha * beta;
        if (alpha * beta == 0.0) stop = true;                                         // x_is_zero, lsqr.py:213-217
    } else if (kind == MK_LSMR) {
        scal[S_OUT + O_NORMR] = beta;
        scal[S_OUT + O_NORMAR] = alpha * beta;
        scal[S_OUT + O_ANORM] = __dsqrt_rn(alpha * alpha);
        scal[S_OUT + O_ACOND] = 1;
        if (alpha * beta == 0) stop = true;                                           // lsmr.py:279-283
    } else if (kind == MK_CRAIG) {
        const double *b = scal + S_BLK;
        scal[S_OUT + O_R1NORM] = __dsqrt_rn(b[craig::R1NORM]);
        scal[S_OUT + O_R2NORM] = __dsqrt_rn(b[craig::RNORM]);
        scal[S_OUT + O_XNORM] = b[craig::XNORM];
        if (beta == 0.0) stop = true;                                                 // craig.py:238-240
    }
    scal[S_ISTOP] = (double)istop;
    st->itn = 0;
    st->nMatvec = 0;
    halt.out(stop || itnlim <= 0);
}

struct OpInitN {     // v /= alpha and the solver's n-space start vectors
    static constexpr int NACC = 0, SLOT0 = 0;
    static constexpr bool NO_MARCH = true;         // (rectangular operators: mk_device.h MkNoMarch)
    const double *scal;
    int kind;
    double *v, *a, *b, *x;     // LSQR: a = w ; LSMR: a = h ; CRAIG: a = w, b = wbar, x
    double alpha, c, s, zeta;
    __device__ bool prologue(double *, bool) {
        const double *blk = scal + S_BLK;
        alpha = blk[B_ALPHA];
        c = s = zeta = 0.0;
        if (kind == MK_CRAIG) {
            c = blk[craig::C];
            s = blk[craig::S];
            zeta = blk[craig::ZETA];
        }
        return false;
    }
    __device__ bool skip() const { return false; }
    __device__ void elem(int64_t i) {
        double vv = v[i];
        if (alpha > 0) vv = vv / alpha;                                               // lsqr.py:207
        v[i] = vv;
        if (kind == MK_LSQR || kind == MK_LSMR) {
            if (kind == MK_LSMR || alpha > 0) a[i] = vv;                              // lsqr.py:209 / lsmr.py:240
        } else if (kind == MK_CRAIG) {
            const double wv = c * vv;                                                 // craig.py:255-257
            a[i] = wv;
            b[i] = s * vv;
            x[i] = zeta * wv;
        }
    }
    __device__ void pair(int64_t i, double *) {
        elem(i);
        elem(i + 1);
    }
    __device__ void one(int64_t i, double *) { elem(i); }
};

struct OpInitM {     // CRAIG: d = u / rho ; r = tau d.  CRAIG-MR: d = u / alpha_hat
    static constexpr int NACC = 0, SLOT0 = 0;
    static constexpr bool NO_MARCH = true;         // (rectangular operators: mk_device.h MkNoMarch)
    const double *scal;
    int kind;
    const double *u;
    double *d, *r;
    double div, tau;
    __device__ bool prologue(double *, bool) {
        const double *blk = scal + S_BLK;
        const double alpha = blk[B_ALPHA];
        div = (kind == MK_CRAIG) ? hyp(alpha, 1) : __dsqrt_rn(alpha * alpha + 1);      // craig.py:246 / craigmr.py:101
        tau = (kind == MK_CRAIG) ? blk[craig::TAU] : 0.0;
        return false;
    }
    __device__ bool skip() const { return false; }
    __device__ void elem(int64_t i) {
        const double dv = u[i] / div;                                                 // craig.py:247 / craigmr.py:108
        d[i] = dv;
        if (kind == MK_CRAIG) r[i] = tau * dv;                                        // craig.py:249
    }
    __device__ void pair(int64_t i, double *) {
        elem(i);
        elem(i + 1);
    }
    __device__ void one(int64_t i, double *) { elem(i); }
};

__global__ __launch_bounds__(MK_BLOCK) void lls_fill_kernel(double *v, int64_t n, double a) {
    for (int64_t i = (int64_t)blockIdx.x * MK_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * MK_BLOCK) v[i] = a;
}

struct LlsSolver : mk_solver {
    int kind;
    int64_t m = 0, nn = 0, nF = 0;
    double *d_u = nullptr, *d_v = nullptr, *d_x = nullptr;
    double *d_a = nullptr, *d_b = nullptr;      // LSQR: w ; LSMR: h, hbar ; CRAIG: w, wbar
    double *d_d = nullptr, *d_r = nullptr, *d_dbar = nullptr;
    double *d_Mu = nullptr, *d_Nv = nullptr;     // only with preconditioners
    const double *d_dm = nullptr, *d_dn = nullptr; // diagonals of M (m entries) and N (n entries), borrowed
    int np_A = 1, np_At = 1, np_n = 1, np_m = 1;
    int64_t itnlim = 0;
    // M / N as host callbacks (mk_solver_set_lls_precon_callback): the kernels run with a diagonal of ones and the
    // vector `u = M(Mu)` / `v = N(Nv)` is replaced by the callback's result right after the kernel that formed it;
    // <u, Mu> / <v, Nv> are then re-formed by a dot kernel (same scheme as mk_solver::host_precon)
    mk_precon_fn fn_m = nullptr, fn_n = nullptr;
    void *user_m = nullptr, *user_n = nullptr;
    double *d_ones_m = nullptr, *d_ones_n = nullptr, *h_cb_in = nullptr, *h_cb_out = nullptr;
    int64_t cb_cap = 0;

    ~LlsSolver() override {
        hipFree(d_ones_m);
        hipFree(d_ones_n);
        if (h_cb_in) hipHostFree(h_cb_in);
        if (h_cb_out) hipHostFree(h_cb_out);
    }

    // out = fn(in) on the host.  Skipped once the loop has halted (unless `force`: the setup calls) and when the
    // product that formed `in` did not run (`need_beta`: lsqr.py:258, everything about v happens `if beta > 0`).
    int host_apply(mk_precon_fn fn, void *user, const double *in_dev, double *out_dev, int64_t len, bool force,
                   bool need_beta) {
        int h = 0;
        double beta = 1.0;
        MK_HIP(hipMemcpyAsync(&h, d_halt + (q & 1), sizeof(int), hipMemcpyDeviceToHost, stream));
        if (need_beta) MK_HIP(hipMemcpyAsync(&beta, d_scal + S_BETA, sizeof(double), hipMemcpyDeviceToHost, stream));
        if (len > 0) MK_HIP(hipMemcpyAsync(h_cb_in, in_dev, sizeof(double) * (size_t)len, hipMemcpyDeviceToHost, stream));
        MK_HIP(hipStreamSynchronize(stream));
        if ((h && !force) || !(beta > 0)) return MK_OK;
        if (fn(user, h_cb_in, h_cb_out) != 0) {
            const int rc = mk_fail(MK_ERR_STATE, "the host preconditioner callback (M or N) reported a failure");
            if (mk_ctx().pending_rc == MK_OK) mk_ctx().pending_rc = rc;
            return rc;
        }
        if (len > 0) MK_HIP(hipMemcpyAsync(out_dev, h_cb_out, sizeof(double) * (size_t)len, hipMemcpyHostToDevice, stream));
        return MK_OK;
    }
    // <v, Nv> comes from the A' product's epilogue, or from the dot kernel that follows an N callback
    int np_vv() const { return fn_n ? np_n : np_At; }
    int apply_M(bool force) {                    // u = M(Mu) ; <u, Mu>
        if (!fn_m) return MK_OK;
        int rc = host_apply(fn_m, user_m, d_Mu, d_u, m, force, false);
        if (rc != MK_OK) return rc;
        mk_launch_stream(this, MkOpDot<SLOT_UU>{d_u, d_Mu}, m);
        return MK_OK;
    }
    int apply_N(bool force) {                    // v = N(Nv) ; <v, Nv>
        if (!fn_n) return MK_OK;
        int rc = host_apply(fn_n, user_n, d_Nv, d_v, nn, force, true);
        if (rc != MK_OK) return rc;
        mk_launch_stream(this, MkOpDot<SLOT_VV>{d_v, d_Nv}, nn);
        return MK_OK;
    }
    int set_callbacks(mk_precon_fn fm, void *um, mk_precon_fn fn, void *un) {
        const int64_t mm = A->nrows, nc = A->ncols, cap = mm > nc ? mm : nc;
        if ((fm || fn) && cap > cb_cap) {
            if (h_cb_in) hipHostFree(h_cb_in);
            if (h_cb_out) hipHostFree(h_cb_out);
            h_cb_in = h_cb_out = nullptr;
            MK_HIP(hipHostMalloc((void **)&h_cb_in, sizeof(double) * (size_t)(cap > 0 ? cap : 1), hipHostMallocDefault));
            MK_HIP(hipHostMalloc((void **)&h_cb_out, sizeof(double) * (size_t)(cap > 0 ? cap : 1), hipHostMallocDefault));
            cb_cap = cap;
        }
        auto ones = [&](double **p, int64_t len) -> int {
            if (*p) return MK_OK;
            MK_HIP(hipMalloc((void **)p, sizeof(double) * (size_t)(len > 0 ? len : 1) + 16));
            hipLaunchKernelGGL(lls_fill_kernel, dim3(512), dim3(MK_BLOCK), 0, stream, *p, len, 1.0);
            MK_HIP(hipGetLastError());
            return MK_OK;
        };
        int rc;
        if (fm) {
            if ((rc = ones(&d_ones_m, mm)) != MK_OK) return rc;
            d_dm = d_ones_m;
        } else if (fn_m) {
            d_dm = nullptr;
        }
        if (fn) {
            if ((rc = ones(&d_ones_n, nc)) != MK_OK) return rc;
            d_dn = d_ones_n;
        } else if (fn_n) {
            d_dn = nullptr;
        }
        fn_m = fm;
        user_m = um;
        fn_n = fn;
        user_n = un;
        return MK_OK;
    }

    // several GPUs, A = this rank's row block (mk_csr_set_row_block): m-space vectors are slices; n-space vectors are
    // either whole and identical on every rank (mode 1: A' u is all-reduced, all n-space work replicated -- fine for
    // m >> n) or SLICED (mode 2): rank r owns entries [r cnt, (r + 1) cnt) of every n-space vector, A' u is
    // reduce-scattered, the n-space updates and their inner products run on the slices (partial sums all-reduced like
    // the m-space ones) and only v -- which the next A v reads whole -- is all-gathered after its update.  Same bytes
    // on the wire as the all-reduce (reduce-scatter + all-gather), 1 / nranks of the n-space arithmetic and memory.
    bool dist = false, sliced = false;
    double *d_t = nullptr;                       // A' u of the local block, then its sum over the ranks
    double *d_tl = nullptr, *d_xfull = nullptr;  // sliced: this rank's block of the sum; x gathered for the caller
    int64_t cnt = 0, off = 0, nl = 0;            // sliced: entries per rank, this rank's offset, its real entries
    // n-space operands as the kernels see them: the whole vector, or this rank's slice
    double *vL() const { return d_v + off; }
    const double *dnL() const { return d_dn ? d_dn + off : nullptr; }
    int gather_v() { return sliced ? mk_comm_allgather(vL(), d_v, cnt, stream) : (int)MK_OK; }
    // (only LSQR's ||dk||^2 and LSMR's ||x||^2 live in SLOT_X: the CRAIG kernels produce no such partial sums)
    int sum_x() { return (sliced && (kind == MK_LSQR || kind == MK_LSMR)) ? allreduce(SLOT_X, 1) : (int)MK_OK; }

    // <u, Mu> is an m-space inner product: its partial sums are added across the ranks
    int sum_uu() { return dist ? allreduce(SLOT_UU, 1) : (int)MK_OK; }
    // G3: v <- A' u - beta v with <v, v>
    int product_At(bool in_setup) {
        if (!dist) {
            mk_launch_spmv_on(this, At, d_u, EpiV{d_scal, d_v, d_dn, d_Nv, 0.0}, GateV{d_scal});
            return apply_N(in_setup);
        }
        mk_launch_spmv_on(this, At, d_u, MkPlainEpi{d_t}, GateV{d_scal});
        if (sliced) {
            int rc = mk_comm_reduce_scatter_sum(d_t, d_tl, cnt, stream);
            if (rc != MK_OK) return rc;
            mk_launch_stream(this, OpVt{d_scal, d_tl, vL(), dnL(), d_Nv, 0.0}, nl);
            return allreduce(SLOT_VV, 1);
        }
        int rc = mk_comm_allreduce_sum(d_t, nn, stream);   // (a skipped product leaves old data: OpVt skips as well)
        if (rc != MK_OK) return rc;
        mk_launch_stream(this, OpVt{d_scal, d_t, d_v, d_dn, d_Nv, 0.0}, nn);
        return MK_OK;
    }

    explicit LlsSolver(int k) : kind(k) {}

    int setup(const double *rhs, const double *guess) override {
        if (guess) return mk_fail(MK_ERR_UNSUPPORTED, "the least-squares solvers start from x = 0");
        if (!At) return mk_fail(MK_ERR_STATE, "least-squares solver: call mk_solver_set_transpose first");
        if (A->ex.mode >= 0 || At->ex.mode >= 0)
            return mk_fail(MK_ERR_UNSUPPORTED, "least-squares solvers: partition A by row blocks (mk_csr_set_row_block), "
                           "not with a halo / all-gather exchange plan");
        dist = A->row_block && mk_comm_active();
        sliced = dist && A->row_block == 2;
        if (dist && (A->host_fn || At->host_fn || fn_m || fn_n))
            return mk_fail(MK_ERR_UNSUPPORTED, "least-squares solvers: matrix-free operators and M / N callbacks are "
                           "single-GPU");
        if (prm.window < 1 || prm.window > MAXWIN) return mk_fail(MK_ERR_ARG, "window must be in 1..%d", MAXWIN);
        use_hist2 = true;
        m = A->nrows;
        nn = A->ncols;
        itnlim = prm.itnlim;
        np_A = mk_grid_spmv_for(A);
        np_At = mk_grid_spmv_for(At);
        np_n = mk_grid_stream(nn);
        np_m = mk_grid_stream(m);
        // (row blocks: all-reduced slots mean the same on every rank only if all MK_MAXP entries are added; the
        //  producers clear what their grids leave unused, MkHalt::clear_tail)
        if (dist) np_A = np_At = np_n = np_m = MK_MAXP;
        // n-space lengths: nA = what this rank holds of a, b, x, Nv; nF = v and A' u, which the products see whole
        cnt = nn;
        off = 0;
        nl = nn;
        if (sliced) {
            int P = 1, rk = 0;
            mk_comm_info(&P, &rk);
            cnt = (((nn + P - 1) / P) + 1) & ~(int64_t)1;    // (even: the stream kernels work on 16-byte pairs)
            off = (int64_t)rk * cnt;
            nl = nn - off < 0 ? 0 : (nn - off < cnt ? nn - off : cnt);
            nF = cnt * P;
        } else {
            nF = nn;
        }
        const int64_t nA = sliced ? cnt : nn;
        if (!d_u) {
            int rc;
            if ((rc = alloc_vec(&d_u, m)) || (rc = alloc_vec(&d_v, nF)) ||
                (rc = alloc_vec(&d_x, kind == MK_CRAIGMR ? m : nA)) || (rc = alloc_vec(&d_a, nA)) ||
                (rc = alloc_vec(&d_b, nA)) || (rc = alloc_vec(&d_d, m)) || (rc = alloc_vec(&d_r, m)) ||
                (rc = alloc_vec(&d_dbar, m)))
                return rc;
        }
        MK_HIP(hipMemsetAsync(d_v, 0, sizeof(double) * (size_t)nF, stream));
        for (double *p : {d_a, d_b}) MK_HIP(hipMemsetAsync(p, 0, sizeof(double) * (size_t)nA, stream));
        for (double *p : {d_d, d_r, d_dbar}) MK_HIP(hipMemsetAsync(p, 0, sizeof(double) * (size_t)m, stream));
        MK_HIP(hipMemsetAsync(d_x, 0, sizeof(double) * (size_t)(kind == MK_CRAIGMR ? m : nA), stream));
        int rc2;
        if (d_dm && !d_Mu && (rc2 = alloc_vec(&d_Mu, m))) return rc2;
        if (d_dn && !d_Nv && (rc2 = alloc_vec(&d_Nv, nA))) return rc2;
        if (dist && !d_t && (rc2 = alloc_vec(&d_t, nF))) return rc2;          // (entries past nn stay zero)
        if (sliced && !d_tl && ((rc2 = alloc_vec(&d_tl, cnt)) || (rc2 = alloc_vec(&d_xfull, nF)))) return rc2;
        if (d_Nv) MK_HIP(hipMemsetAsync(d_Nv, 0, sizeof(double) * (size_t)nA, stream));
        if (d_dm) {
            mk_launch_stream(this, MkOpCopy{rhs, d_Mu}, m);                            // Mu = rhs.copy()   lsqr.py:188
            mk_launch_stream(this, MkOpMul{d_dm, d_Mu, d_u}, m);                       // u = M(Mu)         lsqr.py:190
            mk_launch_stream(this, MkOpDot<SLOT_UU>{d_u, d_Mu}, m);                    // <u, Mu>           lsqr.py:195
            if ((rc2 = apply_M(true)) != MK_OK) return rc2;
        } else {
            mk_launch_stream(this, MkOpCopy{rhs, d_u}, m);                             // Mu = rhs.copy()   lsqr.py:188
            mk_launch_stream(this, MkOpDot<SLOT_UU>{d_u, d_u}, m);                     // lsqr.py:195
        }
        if ((rc2 = sum_uu()) != MK_OK) return rc2;
        mk_launch_stream(this, OpNormU{d_part, np_m, d_scal, d_u, d_dm ? d_Mu : nullptr, 0.0}, m);   // lsqr.py:197-198
        // Nv = A' u (Nv is zero: the epilogue's "- beta Nv" term vanishes exactly)     lsqr.py:200
        if ((rc2 = product_At(true)) != MK_OK) return rc2;
        hipLaunchKernelGGL(lls_init_kernel, dim3(1), dim3(MK_BLOCK), 0, stream, d_part, np_vv(), d_scal, d_status,
                           next_halt(), kind, itnlim);
        mk_launch_stream(this, OpInitN{d_scal, kind, vL(), d_a, d_b, d_x, 0, 0, 0, 0}, nl);
        if ((rc2 = gather_v()) != MK_OK) return rc2;
        if (d_dn) mk_launch_stream(this, OpScaleNv{d_scal, d_scal + S_BLK, 0, d_Nv, 0.0, false}, nl);   // lsqr.py:209
        if (kind == MK_CRAIG || kind == MK_CRAIGMR)
            mk_launch_stream(this, OpInitM{d_scal, kind, d_u, d_d, d_r, 0, 0}, m);
        return MK_OK;
    }

    int enqueue_spmv_only(int which) override {            // (timing aid: a product's kernel without its gate; one GPU)
        if (dist) return mk_fail(MK_ERR_UNSUPPORTED, "product timing of the least-squares solvers is single-GPU");
        const double *blk = d_scal + S_BLK + (int)(it & 1) * BLK;
        if (which == 0) mk_launch_spmv_on(this, A, d_v, EpiU{blk, d_u, d_dm, d_Mu, 0.0, mk_store_nt(A)});
        else if (which == 1) mk_launch_spmv_on(this, At, d_u, EpiV{d_scal, d_v, d_dn, d_Nv, 0.0});
        else return mk_fail(MK_ERR_ARG, "the least-squares solvers have two products per pass (A v, A' u)");
        return MK_OK;
    }

    int enqueue_pass() override {
        const int par = (int)(it & 1);
        const double *blk = d_scal + S_BLK + par * BLK;
        const double *blk_next = d_scal + S_BLK + (par ^ 1) * BLK;
        const int64_t itn = it + 1;
        // G1: u <- A v - alpha u, gated by what is left of the previous pass
        if (kind == MK_LSQR)
            mk_launch_spmv_on(this, A, d_v, EpiU{blk, d_u, d_dm, d_Mu, 0.0, mk_store_nt(A)},
                              lsqr::Gate{d_part, np_n, d_scal, d_status, it, itnlim, prm.atol});
        else if (kind == MK_LSMR)
            mk_launch_spmv_on(this, A, d_v, EpiU{blk, d_u, d_dm, d_Mu, 0.0, mk_store_nt(A)},
                              lsmr::Gate{d_part, np_n, d_scal, d_status, it, itnlim, prm.atol, prm.btol,
                                         prm.conlim > 0 ? 1.0 / prm.conlim : 0.0});
        else
            mk_launch_spmv_on(this, A, d_v, EpiU{blk, d_u, d_dm, d_Mu, 0.0, mk_store_nt(A)}, craig::CountGate{d_status, it, itnlim});
        int rc = apply_M(false);
        if (rc != MK_OK) return rc;
        if ((rc = sum_uu()) != MK_OK) return rc;
        mk_launch_stream(this, OpNormU{d_part, fn_m ? np_m : np_A, d_scal, d_u, d_dm ? d_Mu : nullptr, 0.0}, m);   // G2
        if ((rc = product_At(false)) != MK_OK) return rc;                                                // G3
        if (kind == MK_LSQR) {
            mk_launch_stream(this, lsqr::OpN{d_part, np_vv(), d_scal, d_status, d_hist, par, itn, prm.window, prm.damp,
                                             prm.atol, prm.btol, prm.etol, vL(), d_a, d_x, 0, 0, 0, 0, false}, nl);
        } else if (kind == MK_LSMR) {
            mk_launch_stream(this, lsmr::OpN{d_part, np_vv(), d_scal, d_status, d_hist, par, itn, prm.window, prm.damp,
                                             prm.etol, vL(), d_a, d_b, d_x, 0, 0, 0, 0, false}, nl);
        } else if (kind == MK_CRAIG) {
            mk_launch_stream(this, craig::OpN{d_part, np_vv(), d_scal, d_status, d_hist, par, itn, itnlim, prm.window,
                                              prm.btol, prm.etol, vL(), d_a, d_b, d_x, 0, 0, 0, 0, 0, false}, nl);
            mk_launch_stream(this, craig::OpM{blk_next, d_u, d_d, d_r, 0, 0, 0}, m);
        } else {
            mk_launch_stream(this, craig::OpNmr{d_part, np_vv(), d_scal, d_status, d_hist, par, itn, itnlim, prm.window,
                                                prm.etol, vL(), 0, false}, nl);
            mk_launch_stream(this, craig::OpMmr{blk_next, d_u, d_d, d_dbar, d_x, 0, 0, 0, 0, 0}, m);
        }
        if (d_dn) mk_launch_stream(this, OpScaleNv{d_scal, blk_next, 1, d_Nv, 0.0, false}, nl);          // lsqr.py:272
        if ((rc = sum_x()) != MK_OK) return rc;              // (sliced: ||dk||^2 / ||x||^2 partials of the slices)
        return gather_v();                                   // ... and the next A v reads v whole
    }

    int finish(mk_result *res) override {
        int rc = poll();
        if (rc != MK_OK) return rc;
        fill_result(res);
        const double *o = h_scal + S_OUT;
        res->istop = (int32_t)h_scal[S_ISTOP];
        res->residNorm = o[O_R2NORM];
        res->residNorm0 = h_scal[S_BNORM];
        res->Anorm = o[O_ANORM];
        res->Acond = o[O_ACOND];
        res->Arnorm = o[O_ARNORM];
        res->xnorm = o[O_XNORM];
        res->aux[0] = o[O_R1NORM];
        res->aux[1] = o[O_NORMR];
        res->aux[2] = o[O_NORMAR];
        res->aux[3] = h_scal[S_BLK + (int)(it & 1) * BLK + B_XNRG2];      // xNrgNorm2 of the last pass (lsqr.py:311, the `show` summary)
        const int is = res->istop;
        res->converged = (is == 1 || is == 2 || is == 4 || is == 5 || is == 8) ? 1 : 0;     // `optimal`, lsqr.py:442
        // the gates leave itn one ahead when they admit a pass that the host never enqueued; report completed passes
        res->itn = res->nMatvec / 2;
        if (sliced && kind != MK_CRAIGMR) {                  // the caller reads x whole (CRAIG-MR's x lives in m-space)
            rc = mk_comm_allgather(d_x, d_xfull, cnt, stream);
            if (rc != MK_OK) return rc;
        }
        return MK_OK;
    }

    const double *x() const override { return (sliced && kind != MK_CRAIGMR) ? d_xfull : d_x; }
    const double *vector(int i) const override { return i == 0 ? d_r : (i == 1 ? d_u : (i == 2 ? d_v : nullptr)); }
    int set_metric(const double *dm, const double *dn) {
        if (!fn_m) d_dm = dm;                    // (a side that has a callback keeps its diagonal of ones)
        if (!fn_n) d_dn = dn;
        return MK_OK;
    }
};

}  // namespace

mk_solver *mk_make_lls(int kind) { return new LlsSolver(kind); }

int mk_lls_set_metric(mk_solver *s, const double *dm, const double *dn) {
    return static_cast<LlsSolver *>(s)->set_metric(dm, dn);
}

int mk_lls_set_callbacks(mk_solver *s, mk_precon_fn fn_m, void *user_m, mk_precon_fn fn_n, void *user_n) {
    return static_cast<LlsSolver *>(s)->set_callbacks(fn_m, user_m, fn_n, user_n);
}
