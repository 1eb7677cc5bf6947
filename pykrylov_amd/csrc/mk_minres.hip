// mk_minres.hip -- MINRES, device resident.   Reference: pykrylov/minres/minres.py:115-410.
//
// One pass of the reference loop (minres.py:220-383, unpreconditioned: y and r2 hold the same values)
// = 3 kernels; the four full-vector copies of the reference (:247-248, :293-294) are pointer rotations:
//   K1  v = (1/beta) r2 ; t = A v - shift v - (beta/oldb) r1 ; partial <v, t>   [Lanczos, :236-245]
//       (the scaling is applied on the fly to the gathered entries: a_ij * (s * r2_j))
//   K2  alfa ; y = (-alfa/beta) r2 + t  (written over r1, which becomes the new r2) ; partial <y, y>   [:245-251]
//   K3  beta ; the whole scalar recurrence (rotations, norm estimates, stopping tests, :250-361) ;
//       w = (v - oldeps w1 - delta w2) / gamma (written over w1) ; x += phi w                        [:293-297]
// Algorithmic traffic per pass: B_spmv + 24n (K1: r1 in, v and t out) + 24n (K2) + 48n (K3).
// With a diagonal preconditioner (minres.py:162-163,249) y = d*r2 is a vector of its own: K2 writes it beside the
// new r2 and sums <r2, y>; K1 scales and gathers y instead of r2.
#include "mk_solver.h"

namespace {

constexpr int MAXWIN = 16;
// fixed scalars
enum { S_BETA1 = 0, S_ALFA = 1, S_RNORM = 2, S_ARNORM = 3, S_ANORM = 4, S_ACOND = 5, S_YNORM = 6, S_ISTOP = 7,
       S_TEST2 = 8, S_GBAR = 9,               // (what the reference's `show` table prints besides: minres.py:372-376)
       S_BLK = 16, BLK = 48 };
// ping-pong block: read from scal[S_BLK + par*BLK + k], written to the other parity by K3's lead lane
enum { B_OLDB = 0, B_BETA, B_DBAR, B_EPSLN, B_PHIBAR, B_RHS1, B_RHS2, B_TNORM2, B_YNORM2, B_CS, B_SN, B_GMAX, B_GMIN,
       B_XNRG2, B_ISTOP, B_DERR };       // B_DERR .. B_DERR + MAXWIN - 1
enum { SLOT_ALFA = 0, SLOT_YY = 1 };

__device__ __forceinline__ double hyp(double a, double b) { return __dsqrt_rn(a * a + b * b); }   // minres.py:112-113

struct GateK1 {      // `while itn < itnlim` (minres.py:220)
    MkStatus *st;
    int64_t it, itnlim;
    __device__ bool open(double *, bool lead, bool *stop) {
        if (it >= itnlim) {
            *stop = true;
            return false;
        }
        if (lead) {
            st->itn = it + 1;                                                 // minres.py:221
            st->nMatvec = it + 1;
        }
        return true;
    }
};

struct EpiK1 {
    static constexpr int NACC = 1, SLOT0 = SLOT_ALFA;
    const double *blk;        // state block of this pass
    const double *r2, *r1;    // r2 here is the vector the reference calls y: precon * r2, or r2 itself
    double *v, *t;
    double shift;
    int first;                // itn == 1: no r1 term (minres.py:242)
    double s, c;
    int nt;                   // v and t go past the caches (vectors beyond the Infinity Cache; mk_store_stream)
    __device__ void prologue(double *) {
        s = 1.0 / blk[B_BETA];                                                // minres.py:236
        c = first ? 0.0 : blk[B_BETA] / blk[B_OLDB];                          // minres.py:243
    }
    __device__ double xin(double yj) const { return s * yj; }                 // minres.py:237 on the fly
    __device__ void row(int64_t i, double sum, double *acc) {
        const double vv = s * r2[i];                                          // minres.py:237
        mk_store_stream(v + i, vv, nt);
        double tv = sum - shift * vv;                                         // minres.py:239-240
        if (!first) tv = tv - c * r1[i];                                      // minres.py:243
        mk_store_stream(t + i, tv, nt);
        acc[0] += vv * tv;                                                    // minres.py:245
    }
    // pipelined kernels (brick march): r1[i] arrives as o[0], loaded at the top of the step
    static constexpr int NPF = 1;
    __device__ const double *pf_vec(int) const { return r1; }
    __device__ void row_x_pf(int64_t i, double sum, double vv, const double *o, double *acc) {
        mk_store_stream(v + i, vv, nt);
        double tv = sum - shift * vv;                                         // minres.py:239-240
        if (!first) tv = tv - c * o[0];                                       // minres.py:243
        mk_store_stream(t + i, tv, nt);
        acc[0] += vv * tv;                                                    // minres.py:245
    }
    // r2 IS the product's input vector: where the kernel holds xin(r2[i]) = s * r2[i] already (pattern format: the
    // diagonal entry's LDS slot) it passes it, and r2 is not streamed a second time
    __device__ void row_x(int64_t i, double sum, double vv, double *acc) {
        mk_store_stream(v + i, vv, nt);
        double tv = sum - shift * vv;                                         // minres.py:239-240
        if (!first) tv = tv - c * r1[i];                                      // minres.py:243
        mk_store_stream(t + i, tv, nt);
        acc[0] += vv * tv;                                                    // minres.py:245
    }
};

struct OpK2 {
    static constexpr int NACC = 1, SLOT0 = SLOT_YY;
    const double *part;
    int np;
    double *scal;
    const double *blk;
    const double *r2, *t;
    double *ynew;             // r1's storage: it becomes r2 / y of the next pass
    const double *dg;         // preconditioner diagonal or null
    double *yprec;            // y = precon * r2 (only with dg)
    double c;
    __device__ bool prologue(double *s4, bool lead) {
        const double alfa = mk_total(part + SLOT_ALFA * MK_MAXP, np, s4);
        c = -alfa / blk[B_BETA];                                              // minres.py:246
        if (lead) scal[S_ALFA] = alfa;
        return false;
    }
    __device__ bool skip() const { return false; }
    __device__ void pair(int64_t i, double *acc) {
        const double2 rv = mk_ld2(r2, i), tv = mk_ld2(t, i);
        double2 yv;
        yv.x = c * rv.x + tv.x;                                               // minres.py:246
        yv.y = c * rv.y + tv.y;
        mk_st2(ynew, i, yv);
        if (dg) {                                                             // minres.py:249
            const double2 gv = mk_ld2(dg, i);
            double2 pv;
            pv.x = gv.x * yv.x;
            pv.y = gv.y * yv.y;
            mk_st2(yprec, i, pv);
            acc[0] += yv.x * pv.x;                                            // minres.py:251
            acc[0] += yv.y * pv.y;
        } else {
            acc[0] += yv.x * yv.x;                                            // minres.py:251
            acc[0] += yv.y * yv.y;
        }
    }
    __device__ void one(int64_t i, double *acc) {
        const double yv = c * r2[i] + t[i];
        ynew[i] = yv;
        if (dg) {
            const double pv = dg[i] * yv;
            yprec[i] = pv;
            acc[0] += yv * pv;
        } else {
            acc[0] += yv * yv;
        }
    }
};

struct OpK3 {
    static constexpr int NACC = 0, SLOT0 = 0;
    const double *part;
    int np;
    double *scal;
    MkStatus *st;
    double *hist;
    int par;
    int64_t itn, itnlim;      // itn = index of this pass, 1-based (host-known)
    int window;
    double rtol, etol;
    const double *v, *w1r, *w2;
    double *wnew;             // w1's storage
    double *x;
    double oldeps, delta, denom, phi;
    bool brk;
    __device__ bool prologue(double *s4, bool lead) {
        const double eps = 2.220446049250313e-16;
        const double *bi = scal + S_BLK + par * BLK;
        double *bo = scal + S_BLK + (par ^ 1) * BLK;
        const double beta1 = scal[S_BETA1];
        const double alfa = scal[S_ALFA];
        double beta = mk_total(part + SLOT_YY * MK_MAXP, np, s4);             // minres.py:251
        int istop = (int)bi[B_ISTOP];
        brk = false;
        if (beta < 0) {                                                       // minres.py:252-254
            brk = true;
            if (lead) {
                scal[S_ISTOP] = 6.0;
                bo[B_ISTOP] = 6.0;
            }
            return true;
        }
        const double oldb = bi[B_BETA];                                       // minres.py:250
        beta = __dsqrt_rn(beta);
        const double tnorm2 = bi[B_TNORM2] + alfa * alfa + oldb * oldb + beta * beta;   // minres.py:256
        double gmax = bi[B_GMAX], gmin = bi[B_GMIN];
        if (itn == 1) {                                                       // minres.py:258-264
            if (beta / beta1 <= 10 * eps) istop = -1;
            gmax = fabs(alfa);
            gmin = gmax;
        }
        const double cs0 = bi[B_CS], sn0 = bi[B_SN], dbar0 = bi[B_DBAR], phibar0 = bi[B_PHIBAR];
        oldeps = bi[B_EPSLN];                                                 // minres.py:270-278
        delta = cs0 * dbar0 + sn0 * alfa;
        const double gbar = sn0 * dbar0 - cs0 * alfa;
        const double epsln = sn0 * beta;
        const double dbar = -cs0 * beta;
        const double root = hyp(gbar, dbar);
        const double Arnorm = phibar0 * root;
        double gamma = hyp(gbar, beta);                                       // minres.py:282-287
        gamma = (eps > gamma) ? eps : gamma;
        const double cs = gbar / gamma;
        const double sn = beta / gamma;
        phi = cs * phibar0;
        const double phibar = sn * phibar0;
        denom = 1.0 / gamma;                                                  // minres.py:291
        // direct-error window                                                   minres.py:302-310
        const double xnrg2 = bi[B_XNRG2] + phi * phi;
        double derr_ratio = __builtin_nan("");
        const int slot = (int)(itn % window);
        double dw[MAXWIN];                     // the whole window in one batch of loads (not one round trip each)
#pragma unroll
        for (int k = 0; k < MAXWIN; ++k) dw[k] = bi[B_DERR + k];
        if (itn > window) {
            double ss = 0.0;
#pragma unroll
            for (int k = 0; k < MAXWIN; ++k) {
                if (k < window) {
                    const double e = (k == slot) ? phi : dw[k];
                    ss += e * e;
                }
            }
            const double trnc = __dsqrt_rn(ss);
            const double xnrg = __dsqrt_rn(xnrg2);
            derr_ratio = trnc / xnrg;
            if (trnc < etol * xnrg) istop = 10;
        }
        gmax = (gamma > gmax) ? gamma : gmax;                                 // minres.py:314-319
        gmin = (gamma < gmin) ? gamma : gmin;
        const double z = bi[B_RHS1] / gamma;
        const double ynorm2 = z * z + bi[B_YNORM2];
        const double rhs1 = bi[B_RHS2] - delta * z;
        const double rhs2 = -epsln * z;
        const double Anorm = __dsqrt_rn(tnorm2);                              // minres.py:323-334
        const double ynorm = __dsqrt_rn(ynorm2);
        const double epsx = Anorm * ynorm * eps;
        const double rnorm = phibar;
        const double test1 = rnorm / (Anorm * ynorm);
        const double test2 = root / Anorm;
        const double Acond = gmax / gmin;                                     // minres.py:344
        if (istop == 0) {                                                     // minres.py:349-361
            const double t1 = 1 + test1, t2 = 1 + test2;
            if (t2 <= 1) istop = 2;
            if (t1 <= 1) istop = 1;
            if (itn >= itnlim) istop = 6;
            if (Acond >= 0.1 / eps) istop = 4;
            if (epsx >= beta1) istop = 3;
            if (test2 <= rtol) istop = 2;
            if (test1 <= rtol) istop = 1;
        }
        if (lead) {
            bo[B_OLDB] = oldb;
            bo[B_BETA] = beta;
            bo[B_DBAR] = dbar;
            bo[B_EPSLN] = epsln;
            bo[B_PHIBAR] = phibar;
            bo[B_RHS1] = rhs1;
            bo[B_RHS2] = rhs2;
            bo[B_TNORM2] = tnorm2;
            bo[B_YNORM2] = ynorm2;
            bo[B_CS] = cs;
            bo[B_SN] = sn;
            bo[B_GMAX] = gmax;
            bo[B_GMIN] = gmin;
            bo[B_XNRG2] = xnrg2;
            bo[B_ISTOP] = (double)istop;
#pragma unroll
            for (int k = 0; k < MAXWIN; ++k)
                if (k < window) bo[B_DERR + k] = (k == slot) ? phi : dw[k];
            scal[S_RNORM] = rnorm;
            scal[S_ARNORM] = Arnorm;
            scal[S_ANORM] = Anorm;
            scal[S_ACOND] = Acond;
            scal[S_YNORM] = ynorm;
            scal[S_ISTOP] = (double)istop;
            scal[S_TEST2] = test2;
            scal[S_GBAR] = gbar;
            const int64_t h = st->hist_len % MK_HIST_RING;
            hist[h] = rnorm;                                                  // minres.py:336
            hist[MK_HIST_RING + h] = derr_ratio;                              // minres.py:308
            st->hist_len += 1;
        }
        return istop > 0;                                                     // minres.py:381
    }
    __device__ bool skip() const { return brk; }
    __device__ void elem(double vv, double w1v, double w2v, double &wv, double &xv) {
        wv = (vv - oldeps * w1v - delta * w2v) * denom;                       // minres.py:296
        xv = xv + phi * wv;                                                   // minres.py:297
    }
    __device__ void pair(int64_t i, double *) {
        const double2 vv = mk_ld2(v, i), a = mk_ld2(w1r, i), b = mk_ld2(w2, i);
        double2 xv = mk_ld2(x, i), wv;
        elem(vv.x, a.x, b.x, wv.x, xv.x);
        elem(vv.y, a.y, b.y, wv.y, xv.y);
        mk_st2(wnew, i, wv);
        mk_st2(x, i, xv);
    }
    __device__ void one(int64_t i, double *) {
        double wv, xv = x[i];
        elem(v[i], w1r[i], w2[i], wv, xv);
        wnew[i] = wv;
        x[i] = xv;
    }
};

// minres.py:161-208
__global__ __launch_bounds__(MK_BLOCK) void minres_init_kernel(const double *part, int np, double *scal, MkStatus *st,
                                                               MkHalt halt, int64_t itnlim) {
    __shared__ double s4[4];
    double beta1 = mk_total(part + SLOT_YY * MK_MAXP, np, s4);                // minres.py:166
    if (threadIdx.x == 0) {
        int istop = 0;
        bool done = false;
        if (beta1 < 0) {
            istop = 9;
            done = true;
        }
        if (beta1 == 0.0) done = true;
        if (beta1 > 0) beta1 = __dsqrt_rn(beta1);
        scal[S_BETA1] = beta1;
        scal[S_ISTOP] = (double)istop;
        for (int p = 0; p < 2; ++p) {
            double *b = scal + S_BLK + p * BLK;
            for (int k = 0; k < BLK; ++k) b[k] = 0.0;
            b[B_BETA] = beta1;                                                // minres.py:202-205
            b[B_PHIBAR] = beta1;
            b[B_RHS1] = beta1;
            b[B_CS] = -1.0;
            b[B_ISTOP] = (double)istop;
        }
        st->itn = 0;
        st->nMatvec = 0;
        halt.out(done || (0 >= itnlim));
    }
}

struct MinresSolver : mk_solver {
    double *d_x = nullptr, *d_v = nullptr, *d_t = nullptr;
    double *d_r[2] = {nullptr, nullptr};          // r1 / r2 swap roles every pass
    double *d_w[3] = {nullptr, nullptr, nullptr}; // w1, w2, w rotate
    double *d_y = nullptr;                        // precon * r2 (only with a preconditioner)
    bool takes_precon() const override { return true; }

    int setup(const double *rhs, const double *guess) override {
        if (guess) return mk_fail(MK_ERR_UNSUPPORTED, "MINRES always starts from x = 0 (minres.py:136)");
        if (prm.window < 1 || prm.window > MAXWIN)
            return mk_fail(MK_ERR_ARG, "MINRES: window must be in 1..%d", MAXWIN);
        use_hist2 = true;
        if (!d_x) {
            int rc;
            if ((rc = alloc_vec(&d_x, n)) || (rc = alloc_vec(&d_v, n)) || (rc = alloc_vec(&d_t, n)) ||
                (rc = alloc_vec(&d_r[0], nx)) || (rc = alloc_vec(&d_r[1], nx)) || (rc = alloc_vec(&d_w[0], n)) ||
                (rc = alloc_vec(&d_w[1], n)) || (rc = alloc_vec(&d_w[2], n)))
                return rc;
        }
        int rc0 = MK_OK;
        MK_HIP(hipMemsetAsync(d_x, 0, sizeof(double) * (size_t)n, stream));
        for (int k = 0; k < 3; ++k) MK_HIP(hipMemsetAsync(d_w[k], 0, sizeof(double) * (size_t)n, stream));
        mk_launch_stream(this, MkOpCopy{rhs, d_r[0]}, n);                      // r1 = b            minres.py:161
        mk_launch_stream(this, MkOpCopy{rhs, d_r[1]}, n);                      // y = r2 = b.copy() minres.py:165,208
        if (d_prec) {
            if (!d_y && (rc0 = alloc_vec(&d_y, nx))) return rc0;
            mk_launch_stream(this, MkOpMul{d_prec, d_r[0], d_y}, n);           // y = precon * b    minres.py:162-163
            if (precon_fn && host_precon(d_r[0], d_y) != MK_OK) return MK_ERR_STATE;
            mk_launch_stream(this, MkOpDot<SLOT_YY>{d_r[0], d_y}, n);          // beta1 = <b, y>    minres.py:166
        } else {
            mk_launch_stream(this, MkOpDot<SLOT_YY>{d_r[0], d_r[1]}, n);       // beta1 = <b, y>    minres.py:166
        }
        int rc = allreduce(SLOT_YY, 1);
        if (rc != MK_OK) return rc;
        hipLaunchKernelGGL(minres_init_kernel, dim3(1), dim3(MK_BLOCK), 0, stream, d_part, np_stream, d_scal, d_status,
                           next_halt(), prm.itnlim);
        return MK_OK;
    }

    int enqueue_spmv_only(int which) override {            // (timing aid: the product kernel of a pass without its gate;
        if (which != 0) return mk_fail(MK_ERR_ARG, "MINRES has one product per pass");
        const int par = (int)(it & 1);                     //  writes v and t of the current pass, which the next pass
        const double *blk = d_scal + S_BLK + par * BLK;    //  overwrites anyway)
        double *r1 = d_r[it & 1], *r2 = d_r[(it + 1) & 1];
        mk_launch_spmv(this, d_prec ? d_y : r2, EpiK1{blk, d_prec ? d_y : r2, r1, d_v, d_t, prm.shift, 0, 0.0, 0.0, mk_store_nt(A)}, false);
        return MK_OK;
    }

    int enqueue_pass() override {
        const int par = (int)(it & 1);
        const double *blk = d_scal + S_BLK + par * BLK;
        double *r1 = d_r[it & 1], *r2 = d_r[(it + 1) & 1];                     // after pass it: roles swap
        double *w1 = d_w[it % 3], *w2 = d_w[(it + 1) % 3], *wn = d_w[(it + 2) % 3];
        // (w1, w2, w) <- (w2, w, new): the new vector overwrites the storage of the vector that was w1 ... two
        // passes ago; at pass `it` the roles are w1 = d_w[it%3] (dead after this pass), w2, and the current w.
        // The reference reads w1 := old w2 and w2 := old w; see below.
        double *y = d_prec ? d_y : r2;                                         // minres.py:249
        int rc = exchange(y);
        if (rc != MK_OK) return rc;
        mk_launch_spmv(this, y, EpiK1{blk, y, r1, d_v, d_t, prm.shift, it == 0 ? 1 : 0, 0.0, 0.0, mk_store_nt(A)}, true,
                       GateK1{d_status, it, prm.itnlim});
        if ((rc = allreduce(SLOT_ALFA, 1)) != MK_OK) return rc;
        mk_launch_stream(this, OpK2{d_part, np_spmv, d_scal, blk, r2, d_t, r1, d_prec, d_y, 0.0}, n);
        if (precon_fn) {                                    // y = precon * r2 ; <r2, y> re-formed   minres.py:249-251
            if ((rc = host_precon(r1, d_y)) != MK_OK) return rc;          // (OpK2 wrote the new r2 into r1's storage)
            mk_launch_stream(this, MkOpDot<SLOT_YY>{r1, d_y}, n);
        }
        if ((rc = allreduce(SLOT_YY, 1)) != MK_OK) return rc;
        // reference: w1 = w2 ; w2 = w ; w = f(v, w1, w2).  With storage (a, b, c) = (old w1, old w2, old w):
        // new w1 = b, new w2 = c, new w is written into a.
        (void)w1;
        mk_launch_stream(this, OpK3{d_part, np_stream, d_scal, d_status, d_hist, par, it + 1, prm.itnlim, prm.window,
                                    prm.rtol, prm.etol, d_v, w2, wn, w1, d_x, 0, 0, 0, 0, false}, n);
        return MK_OK;
    }

    int finish(mk_result *res) override {
        int rc = poll();
        if (rc != MK_OK) return rc;
        fill_result(res);
        res->istop = (int32_t)h_scal[S_ISTOP];
        res->residNorm = h_scal[S_RNORM];
        res->residNorm0 = h_scal[S_BETA1];
        res->Arnorm = h_scal[S_ARNORM];
        res->Anorm = h_scal[S_ANORM];
        res->Acond = h_scal[S_ACOND];
        res->ynorm = h_scal[S_YNORM];
        res->aux[0] = h_scal[S_TEST2];                    // test2 = root / Anorm and gbar of the last pass (minres.py:374-375)
        res->aux[1] = h_scal[S_GBAR];
        const int is = res->istop;
        res->converged = (is == 1 || is == 2 || is == 3 || is == 4 || is == 10) ? 1 : 0;   // minres.py:395
        return MK_OK;
    }

    const double *x() const override { return d_x; }
};

}  // namespace

mk_solver *mk_make_minres() { return new MinresSolver(); }
