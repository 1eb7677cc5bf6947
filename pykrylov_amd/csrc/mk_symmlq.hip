// mk_symmlq.hip -- SYMMLQ, device resident.   Reference: pykrylov/symmlq/symmlq.py:65-400.
// (The reference crashes at :162, `self.matvec` does not exist; it is read as `self.op * v`.)
//
// Setup (first Lanczos step with local reorthogonalisation, :129-217) is a chain of four kernels whose
// scalars flow through partial sums, so it needs no host round trip either.  One pass of the loop
// (:235-349, unpreconditioned: y and r2 hold the same values) = 3 kernels:
//   K1  v = (1/beta) r2 ; t = A v - shift v - (beta/oldb) r1 ; partial <v, t>                (:300-305)
//   K2  y = t - (alfa/beta) r2  (written over r1, which becomes the new r2) ; partial <y, y>   (:306-311)
//   K3  beta, the plane rotation, x += (z cs) w + (z sn) v ; w = sn w - cs v ; then the loop-top norm
//       estimates and stopping tests of the NEXT pass (:237-297), which decide whether to halt   (:313-349)
// The epilogue (:361-382: transfer to the CG point, step along b, true residual) runs in finish().
// Algorithmic traffic per pass: B_spmv + 24n (K1) + 24n (K2) + 40n (K3).
// With a diagonal preconditioner (symmlq.py:131-132,188-189,308-309,372-373) y = d*r2 is a vector of its own,
// written by the kernels that produce r2 (S4, K2) and read by the scaled gathers (S2, K1); the final step along
// b uses d*b.
#include "mk_solver.h"

namespace {

enum { S_BETA1 = 0, S_ALFA = 1, S_ANORM = 2, S_ACOND = 3, S_CGNORM = 4, S_LQNORM = 5, S_DIAG = 6, S_ISTOP = 7,
       S_ZS = 8, S_LASTBLK = 9, S_BLK = 16, BLK = 16 };
enum { B_OLDB = 0, B_BETA, B_GBAR, B_DBAR, B_RHS1, B_RHS2, B_TNORM, B_YNORM2, B_SNPROD, B_BSTEP, B_GMAX, B_GMIN,
       B_ISTOP };
// a kernel never writes a slot it reads (see mk_bicgstab.hip): setup chain A -> B -> (C, D) -> A
enum { SLOT_A = 0, SLOT_B = 1, SLOT_C = 2, SLOT_D = 3 };

constexpr double EPS = 2.220446049250313e-16;

struct Top {   // values produced at the top of a loop pass (symmlq.py:237-276)
    double anorm, acond, cgnorm, lqnorm, diag;
    int istop;
};

__device__ __forceinline__ Top loop_top(double tnorm, double ynorm2, double gbar, double rhs1, double rhs2,
                                        double snprod, double beta, double beta1, double gmax, double gmin,
                                        double rtol, int istop) {
    Top t;
    t.anorm = __dsqrt_rn(tnorm);
    const double ynorm = __dsqrt_rn(ynorm2);
    const double epsa = t.anorm * EPS;
    const double epsx = t.anorm * ynorm * EPS;
    const double epsr = t.anorm * ynorm * rtol;
    t.diag = gbar;
    if (t.diag == 0) t.diag = epsa;
    t.lqnorm = __dsqrt_rn(rhs1 * rhs1 + rhs2 * rhs2);
    const double qrnorm = snprod * beta1;
    t.cgnorm = qrnorm * beta / fabs(t.diag);
    if (t.lqnorm < t.cgnorm) {                                               // symmlq.py:257-261
        t.acond = gmax / gmin;
    } else {
        const double ad = fabs(t.diag);
        const double denom = (ad < gmin) ? ad : gmin;
        t.acond = gmax / denom;
    }
    if (istop == 0) {                                                        // symmlq.py:271-276
        if (t.acond >= 0.1 / EPS) istop = 4;
        if (epsx >= beta1) istop = 3;
        if (t.cgnorm <= epsx) istop = 2;
        if (t.cgnorm <= epsr) istop = 1;
    }
    t.istop = istop;
    return t;
}

__device__ __forceinline__ void store_top(double *scal, const Top &t) {
    scal[S_ANORM] = t.anorm;
    scal[S_ACOND] = t.acond;
    scal[S_CGNORM] = t.cgnorm;
    scal[S_LQNORM] = t.lqnorm;
    scal[S_DIAG] = t.diag;
    scal[S_ISTOP] = (double)t.istop;
}

// ---------------------------------------------------------------- setup chain
struct GateS2 {      // beta1 tests (symmlq.py:151-158) then the first product (:162)
    const double *part;
    int np;
    double *scal;
    MkStatus *st;
    __device__ bool open(double *s4, bool lead, bool *stop) {
        const double b = mk_total(part + SLOT_A * MK_MAXP, np, s4);           // symmlq.py:134
        if (b < 0 || b == 0) {
            if (lead) {
                scal[S_BETA1] = b;
                scal[S_ISTOP] = (b < 0) ? 8.0 : 0.0;
            }
            *stop = true;
            return false;
        }
        if (lead) {
            scal[S_BETA1] = __dsqrt_rn(b);
            st->nMatvec = 1;
        }
        return true;
    }
};

struct EpiS2 {       // v = y / beta1 ; y = A v - shift v ; <v, y>
    static constexpr int NACC = 1, SLOT0 = SLOT_B;
    const double *part;
    int np;
    const double *y0;
    double *v, *t;
    double shift;
    int has_shift;
    double s;
    __device__ void prologue(double *s4) {
        const double beta1 = __dsqrt_rn(mk_total(part + SLOT_A * MK_MAXP, np, s4));
        s = 1.0 / beta1;                                                      // symmlq.py:159
    }
    __device__ double xin(double yj) const { return s * yj; }
    __device__ void row(int64_t i, double sum, double *acc) {
        const double vv = s * y0[i];                                          // symmlq.py:160
        v[i] = vv;
        double tv = sum;
        if (has_shift) tv = tv - shift * vv;                                  // symmlq.py:177
        t[i] = tv;
        acc[0] += vv * tv;                                                    // symmlq.py:178
    }
};

struct OpS3 {        // y -= (alfa/beta1) r1 ; z = <v,y>, s = <v,v>
    static constexpr int NACC = 2, SLOT0 = SLOT_C;
    const double *part;
    int np;
    double *scal;
    const double *r1, *v;
    double *t;
    double c;
    __device__ bool prologue(double *s4, bool lead) {
        const double alfa = mk_total(part + SLOT_B * MK_MAXP, np, s4);
        c = alfa / scal[S_BETA1];                                             // symmlq.py:179
        if (lead) scal[S_ALFA] = alfa;
        return false;
    }
    __device__ bool skip() const { return false; }
    __device__ void elem(double r1v, double vv, double &tv, double *acc) {
        tv = tv - c * r1v;
        acc[0] += vv * tv;                                                    // symmlq.py:183
        acc[1] += vv * vv;                                                    // symmlq.py:184
    }
    __device__ void pair(int64_t i, double *acc) {
        const double2 a = mk_ld2(r1, i), b = mk_ld2(v, i);
        double2 tv = mk_ld2(t, i);
        elem(a.x, b.x, tv.x, acc);
        elem(a.y, b.y, tv.y, acc);
        mk_st2(t, i, tv);
    }
    __device__ void one(int64_t i, double *acc) {
        double tv = t[i];
        elem(r1[i], v[i], tv, acc);
        t[i] = tv;
    }
};

struct OpS4 {        // y -= (z/s) v ; r2 = y ; <r2, y>
    static constexpr int NACC = 1, SLOT0 = SLOT_A;
    const double *part;
    int np;
    double *scal;
    const double *t, *v;
    double *r2;
    const double *dg;         // preconditioner diagonal or null
    double *yprec;            // y = precon * r2 (only with dg)
    double c;
    __device__ bool prologue(double *s4, bool lead) {
        const double z = mk_total(part + SLOT_C * MK_MAXP, np, s4);
        const double s = mk_total(part + SLOT_D * MK_MAXP, np, s4);
        c = z / s;                                                            // symmlq.py:185
        if (lead) scal[S_ZS] = c;
        return false;
    }
    __device__ bool skip() const { return false; }
    __device__ void pair(int64_t i, double *acc) {
        const double2 tv = mk_ld2(t, i), vv = mk_ld2(v, i);
        double2 y;
        y.x = tv.x - c * vv.x;
        y.y = tv.y - c * vv.y;
        mk_st2(r2, i, y);
        if (dg) {                                                             // symmlq.py:188-189
            const double2 gv = mk_ld2(dg, i);
            double2 pv;
            pv.x = gv.x * y.x;
            pv.y = gv.y * y.y;
            mk_st2(yprec, i, pv);
            acc[0] += y.x * pv.x;                                             // symmlq.py:190
            acc[0] += y.y * pv.y;
        } else {
            acc[0] += y.x * y.x;                                              // symmlq.py:190
            acc[0] += y.y * y.y;
        }
    }
    __device__ void one(int64_t i, double *acc) {
        const double y = t[i] - c * v[i];
        r2[i] = y;
        if (dg) {
            const double pv = dg[i] * y;
            yprec[i] = pv;
            acc[0] += y * pv;
        } else {
            acc[0] += y * y;
        }
    }
};

// symmlq.py:188-217 and the first loop-top evaluation (:235-297)
__global__ __launch_bounds__(MK_BLOCK) void symmlq_init_kernel(const double *part, int np, double *scal, MkStatus *st,
                                                               MkHalt halt, double rtol, int64_t matvec_max) {
    __shared__ double s4[4];
    const bool halted = halt.in();
    double beta = mk_total(part + SLOT_A * MK_MAXP, np, s4);
    if (threadIdx.x != 0) return;
    if (halted) {
        halt.out(true);
        return;
    }
    int istop = 0;
    bool done = false;
    const double beta1 = scal[S_BETA1], alfa = scal[S_ALFA];
    if (beta < 0) {                                                           // symmlq.py:191-193
        istop = 8;
        done = true;
    }
    beta = __dsqrt_rn(beta);
    if (beta <= EPS) istop = -1;                                              // symmlq.py:198-199
    const double gmax = fabs(alfa) + EPS;
    for (int p = 0; p < 2; ++p) {
        double *b = scal + S_BLK + p * BLK;
        b[B_OLDB] = beta1;                                                    // symmlq.py:188, :212-217
        b[B_BETA] = beta;
        b[B_GBAR] = alfa;
        b[B_DBAR] = beta;
        b[B_RHS1] = beta1;
        b[B_RHS2] = 0.0;
        b[B_TNORM] = alfa * alfa + beta * beta;
        b[B_YNORM2] = 0.0;
        b[B_SNPROD] = 1.0;
        b[B_BSTEP] = 0.0;
        b[B_GMAX] = gmax;
        b[B_GMIN] = gmax;
        b[B_ISTOP] = (double)istop;
    }
    scal[S_CGNORM] = beta1;
    scal[S_LQNORM] = 0.0;
    scal[S_DIAG] = 1.0;
    scal[S_ISTOP] = (double)istop;
    bool stop = done;
    if (!done && 1 < matvec_max) {                                            // symmlq.py:235: while nMatvec < matvec_max
        const double *b = scal + S_BLK;
        const Top t = loop_top(b[B_TNORM], 0.0, alfa, beta1, 0.0, 1.0, beta, beta1, gmax, gmax, rtol, istop);
        store_top(scal, t);
        scal[S_BLK + B_ISTOP] = scal[S_BLK + BLK + B_ISTOP] = (double)t.istop;
        st->itn = 1;
        stop = (t.istop != 0);                                                // symmlq.py:292-293
    } else {
        stop = true;
    }
    halt.out(stop);
}

// ---------------------------------------------------------------- loop kernels
struct CountGate {
    MkStatus *st;
    int64_t nmv;
    __device__ bool open(double *, bool lead, bool *) {
        if (lead) st->nMatvec = nmv + 1;                                      // symmlq.py:302
        return true;
    }
};

struct EpiK1 {
    static constexpr int NACC = 1, SLOT0 = SLOT_A;
    const double *blk;
    const double *r2, *r1;
    double *v, *t;
    double shift;
    int has_shift;
    double s, c;
    int nt;                   // v and t go past the caches (mk_store_stream, mk_solver.h)
    __device__ void prologue(double *) {
        s = 1 / blk[B_BETA];                                                  // symmlq.py:300
        c = blk[B_BETA] / blk[B_OLDB];                                        // symmlq.py:304
    }
    __device__ double xin(double yj) const { return s * yj; }
    __device__ void row(int64_t i, double sum, double *acc) {
        const double vv = s * r2[i];                                          // symmlq.py:301
        mk_store_stream(v + i, vv, nt);
        double tv = sum;
        if (has_shift) tv = tv - shift * vv;                                  // symmlq.py:303
        tv = tv - c * r1[i];                                                  // symmlq.py:304
        mk_store_stream(t + i, tv, nt);
        acc[0] += vv * tv;                                                    // symmlq.py:305
    }
    // pipelined kernels (brick march): vv = xin(r2[i]) comes from the kernel's registers, r1[i] arrives as o[0]
    static constexpr int NPF = 1;
    __device__ const double *pf_vec(int) const { return r1; }
    __device__ void row_x_pf(int64_t i, double sum, double vv, const double *o, double *acc) {
        mk_store_stream(v + i, vv, nt);
        double tv = sum;
        if (has_shift) tv = tv - shift * vv;                                  // symmlq.py:303
        tv = tv - c * o[0];                                                   // symmlq.py:304
        mk_store_stream(t + i, tv, nt);
        acc[0] += vv * tv;                                                    // symmlq.py:305
    }
};

struct OpK2 {
    static constexpr int NACC = 1, SLOT0 = SLOT_B;
    const double *part;
    int np;
    double *scal;
    const double *blk;
    const double *r2, *t;
    double *ynew;
    const double *dg;         // preconditioner diagonal or null
    double *yprec;            // y = precon * r2 (only with dg)
    double c;
    __device__ bool prologue(double *s4, bool lead) {
        const double alfa = mk_total(part + SLOT_A * MK_MAXP, np, s4);
        c = alfa / blk[B_BETA];                                               // symmlq.py:306
        if (lead) scal[S_ALFA] = alfa;
        return false;
    }
    __device__ bool skip() const { return false; }
    __device__ void pair(int64_t i, double *acc) {
        const double2 rv = mk_ld2(r2, i), tv = mk_ld2(t, i);
        double2 y;
        y.x = tv.x - c * rv.x;
        y.y = tv.y - c * rv.y;
        mk_st2(ynew, i, y);
        if (dg) {                                                             // symmlq.py:308-309
            const double2 gv = mk_ld2(dg, i);
            double2 pv;
            pv.x = gv.x * y.x;
            pv.y = gv.y * y.y;
            mk_st2(yprec, i, pv);
            acc[0] += y.x * pv.x;                                             // symmlq.py:311
            acc[0] += y.y * pv.y;
        } else {
            acc[0] += y.x * y.x;                                              // symmlq.py:311
            acc[0] += y.y * y.y;
        }
    }
    __device__ void one(int64_t i, double *acc) {
        const double y = t[i] - c * r2[i];
        ynew[i] = y;
        if (dg) {
            const double pv = dg[i] * y;
            yprec[i] = pv;
            acc[0] += y * pv;
        } else {
            acc[0] += y * y;
        }
    }
};

struct OpK3 {
    static constexpr int NACC = 0, SLOT0 = 0;
    const double *part;
    int np;
    double *scal;
    MkStatus *st;
    int par;
    int64_t nmv, matvec_max;          // products done after this pass (host-known)
    double rtol;
    const double *v;
    double *w, *x;
    double s, t, cs, sn;
    bool brk;
    __device__ bool prologue(double *s4, bool lead) {
        const double *bi = scal + S_BLK + par * BLK;
        double *bo = scal + S_BLK + (par ^ 1) * BLK;
        const double beta1 = scal[S_BETA1], alfa = scal[S_ALFA];
        double beta = mk_total(part + SLOT_B * MK_MAXP, np, s4);              // symmlq.py:311
        int istop = (int)bi[B_ISTOP];
        brk = false;
        if (beta < 0) {                                                       // symmlq.py:313-315
            brk = true;
            if (lead) {
                scal[S_ISTOP] = 6.0;
                bo[B_ISTOP] = 6.0;
            }
            return true;
        }
        const double oldb = bi[B_BETA];                                       // symmlq.py:310
        beta = __dsqrt_rn(beta);
        const double tnorm = bi[B_TNORM] + alfa * alfa + oldb * oldb + beta * beta;
        const double gbar0 = bi[B_GBAR], dbar0 = bi[B_DBAR];
        const double gamma = __dsqrt_rn(gbar0 * gbar0 + oldb * oldb);         // symmlq.py:322-328
        cs = gbar0 / gamma;
        sn = oldb / gamma;
        const double delta = cs * dbar0 + sn * alfa;
        const double gbar = sn * dbar0 - cs * alfa;
        const double epsln = sn * beta;
        const double dbar = -cs * beta;
        const double z = bi[B_RHS1] / gamma;                                  // symmlq.py:332-334
        s = z * cs;
        t = z * sn;
        const double snprod0 = bi[B_SNPROD];
        const double bstep = snprod0 * cs * z + bi[B_BSTEP];                  // symmlq.py:343-349
        const double snprod = snprod0 * sn;
        const double gmax = (gamma > bi[B_GMAX]) ? gamma : bi[B_GMAX];
        const double gmin = (gamma < bi[B_GMIN]) ? gamma : bi[B_GMIN];
        const double ynorm2 = z * z + bi[B_YNORM2];
        const double rhs1 = bi[B_RHS2] - delta * z;
        const double rhs2 = -epsln * z;
        // top of the next pass (symmlq.py:235-293), evaluated only if the `while` admits it
        bool stop;
        Top tp;
        tp.istop = istop;
        const bool enter = nmv < matvec_max;
        if (enter) {
            tp = loop_top(tnorm, ynorm2, gbar, rhs1, rhs2, snprod, beta, beta1, gmax, gmin, rtol, istop);
            stop = (tp.istop != 0);
        } else {
            stop = true;
        }
        if (lead) {
            bo[B_OLDB] = oldb;
            bo[B_BETA] = beta;
            bo[B_GBAR] = gbar;
            bo[B_DBAR] = dbar;
            bo[B_RHS1] = rhs1;
            bo[B_RHS2] = rhs2;
            bo[B_TNORM] = tnorm;
            bo[B_YNORM2] = ynorm2;
            bo[B_SNPROD] = snprod;
            bo[B_BSTEP] = bstep;
            bo[B_GMAX] = gmax;
            bo[B_GMIN] = gmin;
            bo[B_ISTOP] = (double)tp.istop;
            scal[S_LASTBLK] = (double)(par ^ 1);
            if (enter) {
                store_top(scal, tp);
                st->itn += 1;
            }
        }
        return stop;
    }
    __device__ bool skip() const { return brk; }
    __device__ void elem(double vv, double &wv, double &xv) {
        xv = xv + (s * wv + t * vv);                                          // symmlq.py:335
        wv = wv * sn;                                                         // symmlq.py:336
        wv = wv - cs * vv;
    }
    __device__ void pair(int64_t i, double *) {
        const double2 vv = mk_ld2(v, i);
        double2 wv = mk_ld2(w, i), xv = mk_ld2(x, i);
        elem(vv.x, wv.x, xv.x);
        elem(vv.y, wv.y, xv.y);
        mk_st2(w, i, wv);
        mk_st2(x, i, xv);
    }
    __device__ void one(int64_t i, double *) {
        double wv = w[i], xv = x[i];
        elem(v[i], wv, xv);
        w[i] = wv;
        x[i] = xv;
    }
};

// ---------------------------------------------------------------- epilogue kernels
struct OpFinX {      // x += zbar w (CG point) ; x += bstep b ; <x,x>        symmlq.py:361-374, :382
    static constexpr int NACC = 1, SLOT0 = SLOT_A;
    const double *w, *b;
    double *x;
    const double *dg;         // preconditioner diagonal or null: the step is along precon * b (symmlq.py:372-373)
    double zbar, bstep;
    int cg_point;
    __device__ bool prologue(double *, bool) { return false; }
    __device__ bool skip() const { return false; }
    __device__ double f(double xv, double wv, double bv) const {
        if (cg_point) xv = xv + zbar * wv;
        return xv + bstep * bv;
    }
    __device__ void pair(int64_t i, double *acc) {
        const double2 wv = mk_ld2(w, i);
        double2 bv = mk_ld2(b, i);
        if (dg) {
            const double2 gv = mk_ld2(dg, i);
            bv.x = gv.x * bv.x;
            bv.y = gv.y * bv.y;
        }
        double2 xv = mk_ld2(x, i);
        xv.x = f(xv.x, wv.x, bv.x);
        xv.y = f(xv.y, wv.y, bv.y);
        mk_st2(x, i, xv);
        acc[0] += xv.x * xv.x;
        acc[0] += xv.y * xv.y;
    }
    __device__ void one(int64_t i, double *acc) {
        const double xv = f(x[i], w[i], dg ? dg[i] * b[i] : b[i]);
        x[i] = xv;
        acc[0] += xv * xv;
    }
};

struct EpiFinR {     // r1 = b - (A x - shift x) ; <r1, r1>                    symmlq.py:378-381
    static constexpr int NACC = 1, SLOT0 = SLOT_B;
    const double *x, *b;
    double shift;
    int has_shift;
    __device__ void prologue(double *) {}
    __device__ double xin(double v) const { return v; }
    __device__ void row(int64_t i, double sum, double *acc) {
        double y = sum;
        if (has_shift) y = y - shift * x[i];
        const double r = b[i] - y;
        acc[0] += r * r;
    }
};

__global__ __launch_bounds__(MK_BLOCK) void fin_norms_kernel(const double *part, int npx, int npr, double *out) {
    __shared__ double s4[4];
    const double xx = mk_total(part + SLOT_A * MK_MAXP, npx, s4);
    const double rr = mk_total(part + SLOT_B * MK_MAXP, npr, s4);
    if (threadIdx.x == 0) {
        out[0] = __dsqrt_rn(xx);
        out[1] = __dsqrt_rn(rr);
    }
}

struct SymmlqSolver : mk_solver {
    double *d_x = nullptr, *d_w = nullptr, *d_v = nullptr, *d_t = nullptr, *d_b = nullptr, *d_out = nullptr;
    double *d_r[2] = {nullptr, nullptr};
    double *d_y = nullptr;            // precon * r2 (only with a preconditioner)
    bool takes_precon() const override { return true; }
    int *d_nohalt = nullptr;
    bool finished = false;
    double rnorm = 0.0, xnorm = 0.0;

    ~SymmlqSolver() override { hipFree(d_nohalt); }

    int setup(const double *rhs, const double *guess) override {
        if (guess) return mk_fail(MK_ERR_UNSUPPORTED, "SYMMLQ always starts from x = 0 (symmlq.py:119)");
        finished = false;
        if (!d_x) {
            int rc;
            if ((rc = alloc_vec(&d_x, nx)) || (rc = alloc_vec(&d_w, n)) || (rc = alloc_vec(&d_v, n)) ||
                (rc = alloc_vec(&d_t, n)) || (rc = alloc_vec(&d_b, n)) || (rc = alloc_vec(&d_out, 2)) ||
                (rc = alloc_vec(&d_r[0], nx)) || (rc = alloc_vec(&d_r[1], nx)))
                return rc;
            MK_HIP(hipMalloc((void **)&d_nohalt, 2 * sizeof(int)));
            MK_HIP(hipMemsetAsync(d_nohalt, 0, 2 * sizeof(int), stream));
        }
        MK_HIP(hipMemsetAsync(d_x, 0, sizeof(double) * (size_t)nx, stream));
        MK_HIP(hipMemsetAsync(d_w, 0, sizeof(double) * (size_t)n, stream));
        MK_HIP(hipMemsetAsync(d_v, 0, sizeof(double) * (size_t)n, stream));
        mk_launch_stream(this, MkOpCopy{rhs, d_b}, n);
        mk_launch_stream(this, MkOpCopy{rhs, d_r[0]}, n);                      // r1 = rhs.copy()   symmlq.py:129
        mk_launch_stream(this, MkOpCopy{rhs, d_r[1]}, n);                      // y = rhs.copy()    symmlq.py:133
        int rc;
        if (d_prec && !d_y && (rc = alloc_vec(&d_y, nx))) return rc;
        double *y0 = d_prec ? d_y : d_r[1];
        if (d_prec) mk_launch_stream(this, MkOpMul{d_prec, d_r[0], d_y}, n);   // y = precon * r1   symmlq.py:131-132
        if (precon_fn && (rc = host_precon(d_r[0], d_y)) != MK_OK) return rc;
        mk_launch_stream(this, MkOpDot<SLOT_A>{d_r[0], y0}, n);                // beta1             symmlq.py:134
        if ((rc = allreduce(SLOT_A, 1)) != MK_OK) return rc;
        if ((rc = exchange(y0)) != MK_OK) return rc;
        mk_launch_spmv(this, y0, EpiS2{d_part, np_stream, y0, d_v, d_t, prm.shift, prm.has_shift, 0.0}, false,
                       GateS2{d_part, np_stream, d_scal, d_status});
        if ((rc = allreduce(SLOT_B, 1)) != MK_OK) return rc;
        mk_launch_stream(this, OpS3{d_part, np_spmv, d_scal, d_r[0], d_v, d_t, 0.0}, n);
        if ((rc = allreduce(SLOT_C, 2)) != MK_OK) return rc;
        mk_launch_stream(this, OpS4{d_part, np_stream, d_scal, d_t, d_v, d_r[1], d_prec, d_y, 0.0}, n);
        if (precon_fn) {                                    // y = precon * r2 ; <r2, y> re-formed   symmlq.py:188-190
            if ((rc = host_precon(d_r[1], d_y)) != MK_OK) return rc;
            mk_launch_stream(this, MkOpDot<SLOT_A>{d_r[1], d_y}, n);
        }
        if ((rc = allreduce(SLOT_A, 1)) != MK_OK) return rc;
        hipLaunchKernelGGL(symmlq_init_kernel, dim3(1), dim3(MK_BLOCK), 0, stream, d_part, np_stream, d_scal, d_status,
                           next_halt(), prm.rtol, prm.matvec_max);
        return MK_OK;
    }

    int enqueue_spmv_only(int which) override {            // (timing aid: the product kernel of a pass without its gate)
        if (which != 0) return mk_fail(MK_ERR_ARG, "SYMMLQ has one product per pass");
        const double *blk = d_scal + S_BLK + (int)(it & 1) * BLK;
        double *r1 = d_r[it & 1], *r2 = d_r[(it + 1) & 1];
        double *y = d_prec ? d_y : r2;
        mk_launch_spmv(this, y, EpiK1{blk, y, r1, d_v, d_t, prm.shift, prm.has_shift, 0.0, 0.0, mk_store_nt(A)}, false);
        return MK_OK;
    }

    int enqueue_pass() override {
        const int par = (int)(it & 1);
        const double *blk = d_scal + S_BLK + par * BLK;
        double *r1 = d_r[it & 1], *r2 = d_r[(it + 1) & 1];
        double *y = d_prec ? d_y : r2;                                         // symmlq.py:308-309
        int rc = exchange(y);
        if (rc != MK_OK) return rc;
        mk_launch_spmv(this, y, EpiK1{blk, y, r1, d_v, d_t, prm.shift, prm.has_shift, 0.0, 0.0, mk_store_nt(A)}, true,
                       CountGate{d_status, 1 + it});
        if ((rc = allreduce(SLOT_A, 1)) != MK_OK) return rc;
        mk_launch_stream(this, OpK2{d_part, np_spmv, d_scal, blk, r2, d_t, r1, d_prec, d_y, 0.0}, n);
        if (precon_fn) {                                    // y = precon * r2 ; <r2, y> re-formed   symmlq.py:308-310
            if ((rc = host_precon(r1, d_y)) != MK_OK) return rc;          // (OpK2 wrote the new r2 into r1's storage)
            mk_launch_stream(this, MkOpDot<SLOT_B>{r1, d_y}, n);
        }
        if ((rc = allreduce(SLOT_B, 1)) != MK_OK) return rc;
        mk_launch_stream(this, OpK3{d_part, np_stream, d_scal, d_status, par, 2 + it, prm.matvec_max, prm.rtol, d_v,
                                    d_w, d_x, 0, 0, 0, 0, false}, n);
        return MK_OK;
    }

    int finish(mk_result *res) override {
        int rc = poll();
        if (rc != MK_OK) return rc;
        if (halted && !finished) {
            // everything after the loop (symmlq.py:361-382), once
            const double *b = h_scal + S_BLK + ((int)h_scal[S_LASTBLK]) * BLK;   // state after the last full pass
            const double beta1 = h_scal[S_BETA1];
            const double cgnorm = h_scal[S_CGNORM], lqnorm = h_scal[S_LQNORM];
            double bstep = b[B_BSTEP], zbar = 0.0;
            int cg_point = 0;
            if (cgnorm < lqnorm) {                                            // symmlq.py:361-365
                zbar = b[B_RHS1] / h_scal[S_DIAG];
                bstep = b[B_SNPROD] * zbar + bstep;
                cg_point = 1;
            }
            if (beta1 != 0) bstep = bstep / beta1;                            // symmlq.py:369
            const MkHalt nh{d_nohalt, 0, mk_comm_active() ? MK_MAXP : 0};
            const double *bdir = d_b;
            if (precon_fn) {                                 // the step is along precon * b   symmlq.py:372-373
                if ((rc = host_precon(d_b, d_t, true)) != MK_OK) return rc;
                bdir = d_t;                                  // (with the unit diagonal the kernel multiplies it by 1.0)
            }
            hipLaunchKernelGGL(mk_stream_kernel<OpFinX>, dim3(mk_grid_stream(n)), dim3(MK_BLOCK), 0, stream,
                               OpFinX{d_w, bdir, d_x, d_prec, zbar, bstep, cg_point}, n, nh, d_part);
            if ((rc = allreduce(SLOT_A, 1)) != MK_OK) return rc;
            if ((rc = exchange(d_x)) != MK_OK) return rc;
            if ((rc = mk_exchange_wait(A, stream)) != MK_OK) return rc;      // one launch over all tiles below
            mk_spmv_launch(A, mk_grid_spmv_for(A), stream, d_x, EpiFinR{d_x, d_b, prm.shift, prm.has_shift},
                           MkNoGate(), nh, d_part);
            if ((rc = allreduce(SLOT_B, 1)) != MK_OK) return rc;
            hipLaunchKernelGGL(fin_norms_kernel, dim3(1), dim3(MK_BLOCK), 0, stream, d_part, np_stream, np_spmv, d_out);
            double out[2];
            MK_HIP(hipMemcpyAsync(out, d_out, sizeof(out), hipMemcpyDeviceToHost, stream));
            MK_HIP(hipStreamSynchronize(stream));
            xnorm = out[0];
            rnorm = out[1];
            finished = true;
        }
        fill_result(res);
        if (finished) res->nMatvec += 1;                                      // symmlq.py:378
        res->istop = (int32_t)h_scal[S_ISTOP];
        res->residNorm = rnorm;
        res->xnorm = xnorm;
        res->Anorm = h_scal[S_ANORM];
        res->Acond = h_scal[S_ACOND];
        res->residNorm0 = h_scal[S_BETA1];
        res->aux[0] = h_scal[S_CGNORM];
        res->aux[1] = h_scal[S_LQNORM];
        return MK_OK;
    }

    const double *x() const override { return d_x; }
    const double *vector(int i) const override { return i == 0 ? d_v : (i == 1 ? d_t : nullptr); }
};

}  // namespace

mk_solver *mk_make_symmlq() { return new SymmlqSolver(); }
