// mk_bicgstab.hip -- Bi-CGSTAB, device resident.   Reference: pykrylov/bicgstab/bicgstab.py:43-151.
//
// One pass of the reference loop (bicgstab.py:85-145, unpreconditioned: q is p, z is s) = 4 kernels:
//   B  [loop test on ||r||]  v = A p ; partial <r0, v>                                   (:101-103)
//   C  alpha = rho / <r0,v> ; s = r - alpha v ; partial <s, s>                           (:103-107)
//   D  [early exits on ||s||] t = A s ; partials <t,s>, <t,t>, <r0,t>                    (:111-127)
//   F  omega, rho' ; r = s - omega t ; s *= omega ; x += s ; x += alpha p ; partial <r,r> ;
//      and, already here, next pass's  p = beta p - beta omega v + r                     (:130-139, :87-93)
// Algorithmic traffic per pass: 2 B_spmv + 8n (r0 in B) + 32n (C) + 16n (r0, s in D) + 80n (F).
// With a diagonal preconditioner (q = precon*p, z = precon*s; :96-99,:120-123) the products read q and z, which
// the kernels that produce p and s write alongside (C: z = d*s; F: z *= omega, x += z + alpha q, q' = d*p').
#include "mk_solver.h"

namespace {

enum { S_RHO0 = 0, S_RHO1 = 1, S_THRESH = 2, S_RESID = 3, S_RESID0 = 4, S_ALPHA = 5, S_EXIT = 6 };
// a kernel never writes a slot it reads: a workgroup that finishes early would otherwise overwrite partial
// sums that a later-starting workgroup is still adding up in its prologue / gate
enum { SLOT_R0V = 0, SLOT_SS = 1, SLOT_TS = 2, SLOT_TT = 3, SLOT_R0T = 4, SLOT_RR = 5 };

struct BEpi {    // v = A p, fused <r0, v>
    static constexpr int NACC = 1, SLOT0 = SLOT_R0V;
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

struct GateB {   // loop test after ||r|| (bicgstab.py:139-145), then the product is counted (:101)
    const double *part;
    int np;
    double *scal;
    MkStatus *st;
    int first;
    int64_t matvec_max;
    int64_t nmv;       // products done before this gate: known to the host (nmv0 + 2 * pass), never read
                       // from the status record, which the lead lane updates in this same kernel
    __device__ bool open(double *s4, bool lead, bool *stop) {
        if (!first) {
            const double resid = __dsqrt_rn(mk_total(part + SLOT_RR * MK_MAXP, np, s4));
            const bool fin = (resid <= scal[S_THRESH]) || (nmv >= matvec_max);
            if (lead) scal[S_RESID] = resid;
            if (fin) {
                *stop = true;
                return false;
            }
        }
        if (lead) st->nMatvec = nmv + 1;
        return true;
    }
};

struct OpC {     // alpha, s = r - alpha v, <s,s>
    static constexpr int NACC = 1, SLOT0 = SLOT_SS;
    const double *part;
    int np;
    double *scal;
    int par;
    const double *r, *v;
    double *s;
    const double *dg;                                       // preconditioner diagonal or null
    double *z;                                              // z = precon * s (only with dg)
    double alpha;
    __device__ bool prologue(double *s4, bool lead) {
        const double r0v = mk_total(part + SLOT_R0V * MK_MAXP, np, s4);
        alpha = scal[S_RHO0 + par] / r0v;                   // bicgstab.py:103
        if (lead) scal[S_ALPHA] = alpha;
        return false;
    }
    __device__ bool skip() const { return false; }
    __device__ void pair(int64_t i, double *acc) {
        const double2 rv = mk_ld2(r, i), vv = mk_ld2(v, i);
        double2 sv;
        sv.x = rv.x - alpha * vv.x;                         // bicgstab.py:104
        sv.y = rv.y - alpha * vv.y;
        mk_st2(s, i, sv);
        if (dg) {                                           // bicgstab.py:120-121
            const double2 dv = mk_ld2(dg, i);
            double2 zv;
            zv.x = dv.x * sv.x;
            zv.y = dv.y * sv.y;
            mk_st2(z, i, zv);
        }
        acc[0] += sv.x * sv.x;
        acc[0] += sv.y * sv.y;
    }
    __device__ void one(int64_t i, double *acc) {
        const double sv = r[i] - alpha * v[i];
        s[i] = sv;
        if (dg) z[i] = dg[i] * sv;
        acc[0] += sv * sv;
    }
};

struct DEpi {    // t = A s, fused <t,s>, <t,t>, <r0,t>
    static constexpr int NACC = 3, SLOT0 = SLOT_TS;
    const double *s, *r0;
    double *t;
    int nt;
    __device__ void prologue(double *) {}
    __device__ double xin(double x) const { return x; }
    __device__ void row(int64_t r, double sum, double *acc) {
        mk_store_stream(t + r, sum, nt);
        acc[0] += sum * s[r];
        acc[1] += sum * sum;
        acc[2] += r0[r] * sum;
    }
    static constexpr int NPF = 2;                         // pipelined kernels: s[r], r0[r] arrive as o[0], o[1]
    __device__ const double *pf_vec(int j) const { return j == 0 ? s : r0; }
    __device__ void row_pf(int64_t r, double sum, const double *o, double *acc) {
        mk_store_stream(t + r, sum, nt);
        acc[0] += sum * o[0];
        acc[1] += sum * sum;
        acc[2] += o[1] * sum;
    }
};

struct GateD {   // exits that follow ||s|| (bicgstab.py:107-118); the decision is left in S_EXIT for F
    const double *part;
    int np;
    double *scal;
    MkStatus *st;
    int64_t matvec_max;
    int64_t nmv;       // products done before this gate (nmv0 + 2 * pass + 1)
    __device__ bool open(double *s4, bool lead, bool *stop) {
        const double resid = __dsqrt_rn(mk_total(part + SLOT_SS * MK_MAXP, np, s4));
        int ex = 0;
        if (resid <= scal[S_THRESH]) ex = 1;                // converged: F does x += alpha q and stops
        else if (nmv >= matvec_max) ex = 2;                 // out of products: F stops
        if (lead) {
            scal[S_RESID] = resid;
            scal[S_EXIT] = (double)ex;
            if (!ex) st->nMatvec = nmv + 1;                 // bicgstab.py:125
        }
        return ex == 0;
    }
};

struct OpF {
    static constexpr int NACC = 1, SLOT0 = SLOT_RR;
    const double *part;
    int np;
    double *scal;
    MkStatus *st;
    int par;
    const double *t, *v;
    double *s, *r, *x, *p;
    const double *dg;                                       // preconditioner diagonal or null
    double *q, *z;                                          // q = precon * p, z = precon * s (only with dg)
    double alpha, omega, beta, bo;
    int ex;
    __device__ bool prologue(double *s4, bool lead) {
        alpha = scal[S_ALPHA];
        ex = (int)scal[S_EXIT];
        omega = beta = bo = 0.0;
        if (ex == 0) {
            const double ts = mk_total(part + SLOT_TS * MK_MAXP, np, s4);
            const double tt = mk_total(part + SLOT_TT * MK_MAXP, np, s4);
            const double r0t = mk_total(part + SLOT_R0T * MK_MAXP, np, s4);
            const double rho = scal[S_RHO0 + par];
            omega = ts / tt;                                // bicgstab.py:126
            const double rho_next = -omega * r0t;           // bicgstab.py:127
            beta = rho_next / rho * alpha / omega;          // bicgstab.py:87 (next pass)
            bo = beta * omega;
            if (lead) scal[S_RHO0 + (par ^ 1)] = rho_next;  // bicgstab.py:88
        }
        if (lead) st->itn += 1;
        return ex != 0;
    }
    __device__ bool skip() const { return ex == 2; }
    // zv: the vector that is scaled by omega and added to x (s itself without a preconditioner);
    // qv: the vector x advances along (p itself without a preconditioner)
    __device__ void elem(double tv, double vv, double sv, double &zv, double qv, double dv, double &rv, double &xv,
                         double &pv, double &qn, double *acc) {
        if (ex == 1) {
            xv = xv + alpha * qv;                           // bicgstab.py:112
            return;
        }
        rv = sv - omega * tv;                               // bicgstab.py:130
        zv = zv * omega;                                    // bicgstab.py:135
        xv = xv + zv;                                       // bicgstab.py:136
        xv = xv + alpha * qv;                               // bicgstab.py:137
        acc[0] += rv * rv;                                  // bicgstab.py:139
        pv = pv * beta;                                     // bicgstab.py:91
        pv = pv - bo * vv;                                  // bicgstab.py:92
        pv = pv + rv;                                       // bicgstab.py:93
        qn = dv * pv;                                       // bicgstab.py:96-97 (next pass; unused without dg)
    }
    __device__ void pair(int64_t i, double *acc) {
        double2 tv{0, 0}, vv{0, 0}, sv{0, 0}, rv{0, 0}, zv{0, 0}, dv{0, 0}, qn{0, 0};
        if (ex == 0) {
            tv = mk_ld2(t, i);
            vv = mk_ld2(v, i);
            sv = mk_ld2(s, i);
        }
        double2 xv = mk_ld2(x, i), pv = mk_ld2(p, i), qv = pv;
        if (dg) {
            qv = mk_ld2(q, i);
            if (ex == 0) {
                zv = mk_ld2(z, i);
                dv = mk_ld2(dg, i);
            }
        } else {
            zv = sv;
        }
        elem(tv.x, vv.x, sv.x, zv.x, qv.x, dv.x, rv.x, xv.x, pv.x, qn.x, acc);
        elem(tv.y, vv.y, sv.y, zv.y, qv.y, dv.y, rv.y, xv.y, pv.y, qn.y, acc);
        mk_st2(x, i, xv);
        if (ex == 0) {
            mk_st2(dg ? z : s, i, zv);
            mk_st2(r, i, rv);
            mk_st2(p, i, pv);
            if (dg) mk_st2(q, i, qn);
        }
    }
    __device__ void one(int64_t i, double *acc) {
        double tv = 0, vv = 0, sv = 0, rv = 0, zv = 0, dv = 0, qn = 0, xv = x[i], pv = p[i], qv = pv;
        if (ex == 0) {
            tv = t[i];
            vv = v[i];
            sv = s[i];
        }
        if (dg) {
            qv = q[i];
            if (ex == 0) {
                zv = z[i];
                dv = dg[i];
            }
        } else {
            zv = sv;
        }
        elem(tv, vv, sv, zv, qv, dv, rv, xv, pv, qn, acc);
        x[i] = xv;
        if (ex == 0) {
            (dg ? z : s)[i] = zv;
            r[i] = rv;
            p[i] = pv;
            if (dg) q[i] = qn;
        }
    }
};

// bicgstab.py:67-72
__global__ __launch_bounds__(MK_BLOCK) void bicgstab_init_kernel(const double *part, int np, double *scal, MkStatus *st,
                                                                 MkHalt halt, double abstol, double reltol,
                                                                 int64_t matvec_max, int64_t nmv0) {
    __shared__ double s4[4];
    const double rho = mk_total(part + SLOT_RR * MK_MAXP, np, s4);
    if (threadIdx.x == 0) {
        const double resid0 = fabs(__dsqrt_rn(rho));
        const double rel = reltol * resid0;
        const double thresh = (rel > abstol) ? rel : abstol;
        scal[S_RHO0] = rho;            // "rho_next" of the first pass; beta = rho_next / 1 * 1 / 1 makes p = r
        scal[S_THRESH] = thresh;
        scal[S_RESID] = resid0;
        scal[S_RESID0] = resid0;
        st->nMatvec = nmv0;
        halt.out((resid0 <= thresh) || (nmv0 >= matvec_max));
    }
}

struct BicgstabSolver : mk_solver {
    double *d_x = nullptr, *d_r0 = nullptr, *d_r = nullptr, *d_p = nullptr, *d_v = nullptr, *d_s = nullptr,
           *d_t = nullptr, *d_q = nullptr, *d_z = nullptr;
    int64_t nmv0 = 0;
    bool takes_precon() const override { return true; }

    int setup(const double *rhs, const double *guess) override {
        if (!d_x) {
            int rc;
            if ((rc = alloc_vec(&d_x, nx)) || (rc = alloc_vec(&d_r0, n)) || (rc = alloc_vec(&d_r, n)) ||
                (rc = alloc_vec(&d_p, nx)) || (rc = alloc_vec(&d_v, n)) || (rc = alloc_vec(&d_s, nx)) ||
                (rc = alloc_vec(&d_t, n)))
                return rc;
        }
        if (d_prec && !d_q) {
            int rc;
            if ((rc = alloc_vec(&d_q, nx)) || (rc = alloc_vec(&d_z, nx))) return rc;
        }
        nmv0 = 0;
        if (guess) {
            MK_HIP(hipMemcpyAsync(d_x, guess, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream));
            int rc = exchange(d_x);
            if (rc != MK_OK) return rc;
            mk_launch_spmv(this, d_x, MkPlainEpi{d_t}, false);                 // r0 = rhs - A x   bicgstab.py:63-65
            mk_launch_stream(this, MkOpSub{rhs, d_t, d_r0}, n);
            nmv0 = 1;
        } else {
            MK_HIP(hipMemsetAsync(d_x, 0, sizeof(double) * (size_t)nx, stream));
            mk_launch_stream(this, MkOpCopy{rhs, d_r0}, n);                    // r0 = rhs         bicgstab.py:62
        }
        mk_launch_stream(this, MkOpDot<SLOT_RR>{d_r0, d_r0}, n);               // bicgstab.py:68
        int rc = allreduce(SLOT_RR, 1);
        if (rc != MK_OK) return rc;
        hipLaunchKernelGGL(bicgstab_init_kernel, dim3(1), dim3(MK_BLOCK), 0, stream, d_part, np_stream, d_scal,
                           d_status, next_halt(), prm.abstol, prm.reltol, prm.matvec_max, nmv0);
        // r = r0.copy(); p = v = 0; the first pass's p update (beta p - beta omega v + r) gives exactly r
        mk_launch_stream(this, MkOpCopy{d_r0, d_r}, n);
        mk_launch_stream(this, MkOpCopy{d_r0, d_p}, n);
        if (d_prec) mk_launch_stream(this, MkOpMul{d_prec, d_r0, d_q}, n);     // q = precon * p   bicgstab.py:96-97
        if (precon_fn && host_precon(d_r0, d_q) != MK_OK) return MK_ERR_STATE;
        MK_HIP(hipMemsetAsync(d_v, 0, sizeof(double) * (size_t)n, stream));
        return MK_OK;
    }

    int enqueue_spmv_only(int which) override {            // (timing aid: a product's kernel without its gate)
        if (which == 0) mk_launch_spmv(this, d_prec ? d_q : d_p, BEpi{d_r0, d_v, mk_store_nt(A)}, false);
        else if (which == 1) mk_launch_spmv(this, d_prec ? d_z : d_s, DEpi{d_s, d_r0, d_t, mk_store_nt(A)}, false);
        else return mk_fail(MK_ERR_ARG, "BiCGSTAB has two products per pass");
        return MK_OK;
    }

    int enqueue_pass() override {
        const int par = (int)(it & 1);
        double *qin = d_prec ? d_q : d_p, *zin = d_prec ? d_z : d_s;           // what the two products read
        int rc;
        if (precon_fn && it > 0 && (rc = host_precon(d_p, d_q)) != MK_OK) return rc;   // q = precon * p  bicgstab.py:96-97
        if ((rc = exchange(qin)) != MK_OK) return rc;
        mk_launch_spmv(this, qin, BEpi{d_r0, d_v, mk_store_nt(A)}, true,
                       GateB{d_part, np_stream, d_scal, d_status, it == 0 ? 1 : 0, prm.matvec_max, nmv0 + 2 * it});
        if ((rc = allreduce(SLOT_R0V, 1)) != MK_OK) return rc;
        mk_launch_stream(this, OpC{d_part, np_spmv, d_scal, par, d_r, d_v, d_s, d_prec, d_z, 0.0}, n);
        if ((rc = allreduce(SLOT_SS, 1)) != MK_OK) return rc;
        if (precon_fn && (rc = host_precon(d_s, d_z)) != MK_OK) return rc;     // z = precon * s       bicgstab.py:120-121
        if ((rc = exchange(zin)) != MK_OK) return rc;
        mk_launch_spmv(this, zin, DEpi{d_s, d_r0, d_t, mk_store_nt(A)}, true,
                       GateD{d_part, np_stream, d_scal, d_status, prm.matvec_max, nmv0 + 2 * it + 1});
        if ((rc = allreduce(SLOT_TS, 3)) != MK_OK) return rc;
        mk_launch_stream(this, OpF{d_part, np_spmv, d_scal, d_status, par, d_t, d_v, d_s, d_r, d_x, d_p, d_prec, d_q,
                                   d_z, 0, 0, 0, 0, 0}, n);
        if ((rc = allreduce(SLOT_RR, 1)) != MK_OK) return rc;
        return MK_OK;
    }

    int finish(mk_result *res) override {
        // the last ||r|| is only examined by the next pass's gate: run that gate (and nothing else) now
        int rc = poll();
        if (rc != MK_OK) return rc;
        fill_result(res);
        res->residNorm = h_scal[S_RESID];
        res->residNorm0 = h_scal[S_RESID0];
        res->threshold = h_scal[S_THRESH];
        res->converged = (h_scal[S_RESID] <= h_scal[S_THRESH]) ? 1 : 0;        // bicgstab.py:148
        return MK_OK;
    }

    const double *x() const override { return d_x; }
    const double *vector(int i) const override { return i == 0 ? d_r : (i == 1 ? d_p : nullptr); }
};

}  // namespace

mk_solver *mk_make_bicgstab() { return new BicgstabSolver(); }
