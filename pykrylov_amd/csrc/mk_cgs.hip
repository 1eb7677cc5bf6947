// mk_cgs.hip -- Conjugate Gradient Squared, device resident.   Reference: pykrylov/cgs/cgs.py:40-123.
//
// One pass of the reference loop (cgs.py:76-117, unpreconditioned: y is p, z = u + q) = 4 kernels:
//   B  v = A p ; partial <r0, v>                                                        (:83-84)
//   C  alpha = rho / sigma ; q = u - alpha v ; z = u + q ; x += alpha z                  (:85-94)
//   D  Az = A z, with r -= alpha Az and the partials <r,r>, <r0,r> in its row epilogue   (:95-105)
//   F  [loop test on ||r||] beta = rho'/rho ; u = r + beta q ; p = beta (beta p + q) + u  (:101-114)
// Algorithmic traffic per pass: 2 B_spmv + 8n (r0 in B) + 48n (C) + 24n (r, r0 in D) + 48n (F).
// With a diagonal preconditioner (cgs.py:79-82,88-91): B reads y = d*p, which F writes beside p, and C forms
// z = d*(u + q).
#include "mk_solver.h"

namespace {

enum { S_RHO0 = 0, S_RHO1 = 1, S_THRESH = 2, S_RESID = 3, S_RESID0 = 4, S_ALPHA = 5 };
enum { SLOT_SIGMA = 0, SLOT_RR = 1, SLOT_R0R = 2 };

struct BEpi {    // v = A p, fused <r0, v>; the product is counted by its gate
    static constexpr int NACC = 1, SLOT0 = SLOT_SIGMA;
    const double *r0;
    double *v;
    int nt;            // the product vector goes past the caches (mk_store_stream, mk_solver.h)
    __device__ void prologue(double *) {}
    __device__ double xin(double x) const { return x; }
    __device__ void row(int64_t r, double s, double *acc) {
        mk_store_stream(v + r, s, nt);
        acc[0] += r0[r] * s;
    }
    static constexpr int NPF = 1;                         // pipelined kernels: r0[r] arrives as o[0]
    __device__ const double *pf_vec(int) const { return r0; }
    __device__ void row_pf(int64_t r, double s, const double *o, double *acc) {
        mk_store_stream(v + r, s, nt);
        acc[0] += o[0] * s;
    }
};

struct CountGate {   // no test, only `nMatvec += 1` (cgs.py:83, :95)
    MkStatus *st;
    int64_t nmv;
    __device__ bool open(double *, bool lead, bool *) {
        if (lead) st->nMatvec = nmv + 1;
        return true;
    }
};

struct OpC {
    static constexpr int NACC = 0, SLOT0 = 0;
    const double *part;
    int np;
    double *scal;
    int par;
    const double *u, *v;
    double *q, *z, *x;
    const double *dg;                                                         // preconditioner diagonal or null
    double alpha;
    int defer_x;              // host preconditioner: z is replaced after this kernel, x += alpha z follows (OpXZ)
    __device__ bool prologue(double *s4, bool lead) {
        const double sigma = mk_total(part + SLOT_SIGMA * MK_MAXP, np, s4);   // cgs.py:84
        alpha = scal[S_RHO0 + par] / sigma;                                   // cgs.py:85
        if (lead) scal[S_ALPHA] = alpha;
        return false;
    }
    __device__ bool skip() const { return false; }
    __device__ void elem(double uv, double vv, double dv, double &qv, double &zv, double &xv) {
        qv = uv - alpha * vv;                                                 // cgs.py:86
        zv = uv + qv;                                                         // cgs.py:91
        if (dg) zv = dv * zv;                                                 // cgs.py:88-89
        if (!defer_x) xv = xv + alpha * zv;                                   // cgs.py:94
    }
    __device__ void pair(int64_t i, double *) {
        const double2 uv = mk_ld2(u, i), vv = mk_ld2(v, i);
        double2 xv = mk_ld2(x, i), qv, zv, dv{0, 0};
        if (dg) dv = mk_ld2(dg, i);
        elem(uv.x, vv.x, dv.x, qv.x, zv.x, xv.x);
        elem(uv.y, vv.y, dv.y, qv.y, zv.y, xv.y);
        mk_st2(q, i, qv);
        mk_st2(z, i, zv);
        mk_st2(x, i, xv);
    }
    __device__ void one(int64_t i, double *) {
        double qv, zv, xv = x[i];
        elem(u[i], v[i], dg ? dg[i] : 0.0, qv, zv, xv);
        q[i] = qv;
        z[i] = zv;
        x[i] = xv;
    }
};

struct OpXZ {    // x += alpha z with the device's alpha (host preconditioner: z = precon * (u + q) came from the host)
    static constexpr int NACC = 0, SLOT0 = 0;
    const double *scal;
    const double *z;
    double *x;
    double alpha;
    __device__ bool prologue(double *, bool) {
        alpha = scal[S_ALPHA];
        return false;
    }
    __device__ bool skip() const { return false; }
    __device__ void pair(int64_t i, double *) {
        const double2 zv = mk_ld2(z, i);
        double2 xv = mk_ld2(x, i);
        xv.x = xv.x + alpha * zv.x;                                           // cgs.py:94
        xv.y = xv.y + alpha * zv.y;
        mk_st2(x, i, xv);
    }
    __device__ void one(int64_t i, double *) { x[i] = x[i] + alpha * z[i]; }
};

struct DEpi {    // Az = A z ; r -= alpha Az ; <r,r>, <r0,r>
    static constexpr int NACC = 2, SLOT0 = SLOT_RR;
    const double *scal;
    const double *r0;
    double *r;
    double alpha;
    __device__ void prologue(double *) { alpha = scal[S_ALPHA]; }
    __device__ double xin(double x) const { return x; }
    __device__ void row(int64_t i, double az, double *acc) {
        const double rv = r[i] - alpha * az;                                  // cgs.py:96
        r[i] = rv;
        acc[0] += rv * rv;                                                    // cgs.py:99
        acc[1] += r0[i] * rv;                                                 // cgs.py:105
    }
    static constexpr int NPF = 2;                         // pipelined kernels: r[i], r0[i] arrive as o[0], o[1]
    __device__ const double *pf_vec(int j) const { return j == 0 ? r : r0; }
    __device__ void row_pf(int64_t i, double az, const double *o, double *acc) {
        const double rv = o[0] - alpha * az;                                  // cgs.py:96
        r[i] = rv;
        acc[0] += rv * rv;                                                    // cgs.py:99
        acc[1] += o[1] * rv;                                                  // cgs.py:105
    }
};

struct OpF {
    static constexpr int NACC = 0, SLOT0 = 0;
    const double *part;
    int np;
    double *scal;
    MkStatus *st;
    int par;
    int64_t matvec_max, nmv;        // products done so far (host-known)
    const double *r, *q;
    double *u, *p;
    const double *dg;                                                         // preconditioner diagonal or null
    double *y;                                                                // y = precon * p (only with dg)
    double beta;
    bool fin;
    __device__ bool prologue(double *s4, bool lead) {
        const double resid = __dsqrt_rn(mk_total(part + SLOT_RR * MK_MAXP, np, s4));
        fin = (resid <= scal[S_THRESH]) || (nmv >= matvec_max);               // cgs.py:101
        const double rho_next = mk_total(part + SLOT_R0R * MK_MAXP, np, s4);
        beta = rho_next / scal[S_RHO0 + par];                                 // cgs.py:106
        if (lead) {
            scal[S_RESID] = resid;
            scal[S_RHO0 + (par ^ 1)] = rho_next;                              // cgs.py:107
            st->itn += 1;
        }
        return fin;
    }
    __device__ bool skip() const { return fin; }
    __device__ void elem(double rv, double qv, double &uv, double &pv) {
        uv = rv + beta * qv;                                                  // cgs.py:108
        pv = pv * beta;                                                       // cgs.py:111
        pv = pv + qv;                                                         // cgs.py:112
        pv = pv * beta;                                                       // cgs.py:113
        pv = pv + uv;                                                         // cgs.py:114
    }
    __device__ void pair(int64_t i, double *) {
        const double2 rv = mk_ld2(r, i), qv = mk_ld2(q, i);
        double2 pv = mk_ld2(p, i), uv;
        elem(rv.x, qv.x, uv.x, pv.x);
        elem(rv.y, qv.y, uv.y, pv.y);
        mk_st2(u, i, uv);
        mk_st2(p, i, pv);
        if (dg) {                                                             // cgs.py:79-80 (next pass)
            const double2 dv = mk_ld2(dg, i);
            double2 yv;
            yv.x = dv.x * pv.x;
            yv.y = dv.y * pv.y;
            mk_st2(y, i, yv);
        }
    }
    __device__ void one(int64_t i, double *) {
        double uv, pv = p[i];
        elem(r[i], q[i], uv, pv);
        u[i] = uv;
        p[i] = pv;
        if (dg) y[i] = dg[i] * pv;
    }
};

__global__ __launch_bounds__(MK_BLOCK) void cgs_init_kernel(const double *part, int np, double *scal, MkStatus *st,
                                                            MkHalt halt, double abstol, double reltol,
                                                            int64_t matvec_max) {
    __shared__ double s4[4];
    const double rho = mk_total(part + SLOT_RR * MK_MAXP, np, s4);            // cgs.py:62
    if (threadIdx.x == 0) {
        const double resid0 = fabs(__dsqrt_rn(rho));
        const double rel = reltol * resid0;
        const double thresh = (rel > abstol) ? rel : abstol;
        scal[S_RHO0] = rho;
        scal[S_THRESH] = thresh;
        scal[S_RESID] = resid0;
        scal[S_RESID0] = resid0;
        st->nMatvec = 0;                                                      // cgs.py:59-60: guess product not counted
        halt.out((resid0 <= thresh) || (0 >= matvec_max));                    // cgs.py:68
    }
}

struct CgsSolver : mk_solver {
    double *d_x = nullptr, *d_r0 = nullptr, *d_r = nullptr, *d_u = nullptr, *d_p = nullptr, *d_q = nullptr,
           *d_v = nullptr, *d_z = nullptr, *d_y = nullptr;
    bool takes_precon() const override { return true; }

    int setup(const double *rhs, const double *guess) override {
        if (!d_x) {
            int rc;
            if ((rc = alloc_vec(&d_x, nx)) || (rc = alloc_vec(&d_r0, n)) || (rc = alloc_vec(&d_r, n)) ||
                (rc = alloc_vec(&d_u, n)) || (rc = alloc_vec(&d_p, nx)) || (rc = alloc_vec(&d_q, n)) ||
                (rc = alloc_vec(&d_v, n)) || (rc = alloc_vec(&d_z, nx)))
                return rc;
        }
        if (d_prec && !d_y) {
            int rc = alloc_vec(&d_y, nx);
            if (rc) return rc;
        }
        if (guess) {
            MK_HIP(hipMemcpyAsync(d_x, guess, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream));
            int rc = exchange(d_x);
            if (rc != MK_OK) return rc;
            mk_launch_spmv(this, d_x, MkPlainEpi{d_v}, false);                // r0 = rhs - A x   cgs.py:59-60
            mk_launch_stream(this, MkOpSub{rhs, d_v, d_r0}, n);
        } else {
            MK_HIP(hipMemsetAsync(d_x, 0, sizeof(double) * (size_t)nx, stream));
            mk_launch_stream(this, MkOpCopy{rhs, d_r0}, n);
        }
        mk_launch_stream(this, MkOpDot<SLOT_RR>{d_r0, d_r0}, n);
        int rc = allreduce(SLOT_RR, 1);
        if (rc != MK_OK) return rc;
        hipLaunchKernelGGL(cgs_init_kernel, dim3(1), dim3(MK_BLOCK), 0, stream, d_part, np_stream, d_scal, d_status,
                           next_halt(), prm.abstol, prm.reltol, prm.matvec_max);
        mk_launch_stream(this, MkOpCopy{d_r0, d_r}, n);                        // r = r0.copy()    cgs.py:72
        mk_launch_stream(this, MkOpCopy{d_r0, d_u}, n);                        // u = r0           cgs.py:73
        mk_launch_stream(this, MkOpCopy{d_r0, d_p}, n);                        // p = r0.copy()    cgs.py:74
        if (d_prec) mk_launch_stream(this, MkOpMul{d_prec, d_r0, d_y}, n);     // y = precon * p   cgs.py:79-80
        if (precon_fn && (rc = host_precon(d_p, d_y)) != MK_OK) return rc;
        return MK_OK;
    }

    int enqueue_spmv_only(int which) override {            // (timing aid: a product's kernel without its gate)
        if (which == 0) mk_launch_spmv(this, d_prec ? d_y : d_p, BEpi{d_r0, d_v, mk_store_nt(A)}, false);
        else if (which == 1) mk_launch_spmv(this, d_z, DEpi{d_scal, d_r0, d_r, 0.0}, false);
        else return mk_fail(MK_ERR_ARG, "CGS has two products per pass");
        return MK_OK;
    }

    int enqueue_pass() override {
        const int par = (int)(it & 1);
        double *yin = d_prec ? d_y : d_p;
        int rc;
        if (precon_fn && it > 0 && (rc = host_precon(d_p, d_y)) != MK_OK) return rc;   // y = precon * p   cgs.py:79-80
        if ((rc = exchange(yin)) != MK_OK) return rc;
        mk_launch_spmv(this, yin, BEpi{d_r0, d_v, mk_store_nt(A)}, true, CountGate{d_status, 2 * it});
        if ((rc = allreduce(SLOT_SIGMA, 1)) != MK_OK) return rc;
        mk_launch_stream(this, OpC{d_part, np_spmv, d_scal, par, d_u, d_v, d_q, d_z, d_x, d_prec, 0.0, precon_fn ? 1 : 0}, n);
        if (precon_fn) {                                    // z = precon * (u + q) ; x += alpha z     cgs.py:88-94
            if ((rc = host_precon(d_z, d_z)) != MK_OK) return rc;
            mk_launch_stream(this, OpXZ{d_scal, d_z, d_x, 0.0}, n);
        }
        if ((rc = exchange(d_z)) != MK_OK) return rc;
        mk_launch_spmv(this, d_z, DEpi{d_scal, d_r0, d_r, 0.0}, true, CountGate{d_status, 2 * it + 1});
        if ((rc = allreduce(SLOT_RR, 2)) != MK_OK) return rc;
        mk_launch_stream(this, OpF{d_part, np_spmv, d_scal, d_status, par, prm.matvec_max, 2 * it + 2, d_r, d_q, d_u,
                                   d_p, d_prec, d_y, 0.0, false}, n);
        return MK_OK;
    }

    int finish(mk_result *res) override {
        int rc = poll();
        if (rc != MK_OK) return rc;
        fill_result(res);
        res->residNorm = h_scal[S_RESID];
        res->residNorm0 = h_scal[S_RESID0];
        res->threshold = h_scal[S_THRESH];
        res->converged = (h_scal[S_RESID] <= h_scal[S_THRESH]) ? 1 : 0;        // cgs.py:120
        return MK_OK;
    }

    const double *x() const override { return d_x; }
    const double *vector(int i) const override { return i == 0 ? d_r : (i == 1 ? d_p : nullptr); }
};

}  // namespace

mk_solver *mk_make_cgs() { return new CgsSolver(); }
