// mk_tfqmr.hip -- transpose-free QMR, device resident.   Reference: pykrylov/tfqmr/tfqmr.py:39-159.
//
// One pass of the reference loop (tfqmr.py:85-153, unpreconditioned: z aliases y) = 5 kernels:
//   P2  alpha = rho / sigma ; w -= alpha u ; d = (theta^2 eta / alpha) d + y ; partial <w,w>     (:88-95)
//   P3  theta, c, tau, eta ; x += eta d ; [exit test] ; y -= alpha v                            (:95-107)
//   P4  u = A y, row epilogue: w -= alpha u ; d = (..) d + y ; partials <w,w>, <r0,w>           (:114-118,:128)
//   P5  theta, c, tau, eta ; x += eta d ; [exit test] ; beta ; y = beta y + w ; v = beta (beta v + u)   (:118-139)
//   P6  u = A y, row epilogue: v += u ; partial <r0, v> (the next pass's sigma)                  (:147-150, :88)
// With a diagonal preconditioner (tfqmr.py:77-80,109-112,142-145) z = d*y is a vector of its own: P3 and P5
// write it beside y; the products and the `d += z` updates read it.
// Algorithmic traffic per pass: 2 B_spmv + 56n (P2) + 48n (P3) + 48n (P4 epilogue) + 72n (P5) + 32n (P6 epilogue).
#include "mk_solver.h"

namespace {

// two sets (A: after the first half-step, B: after the second) so that no kernel reads a scalar it writes
enum { S_RHO0 = 0, S_RHO1 = 1, S_THRESH = 2, S_RESID0 = 3, S_ALPHA = 4, S_M = 5,
       S_THETA_A = 6, S_ETA_A = 7, S_TAU_A = 8, S_THETA_B = 9, S_ETA_B = 10, S_TAU_B = 11 };
enum { SLOT_SIGMA = 0, SLOT_WW = 1, SLOT_R0W = 2 };

struct CountGate {
    MkStatus *st;
    int64_t nmv;
    __device__ bool open(double *, bool lead, bool *) {
        if (lead) st->nMatvec = nmv + 1;
        return true;
    }
};

struct OpP2 {
    static constexpr int NACC = 1, SLOT0 = SLOT_WW;
    const double *part;
    int np;
    double *scal;
    int par;
    const double *u, *y;
    double *w, *d;
    double alpha, c1;
    __device__ bool prologue(double *s4, bool lead) {
        const double sigma = mk_total(part + SLOT_SIGMA * MK_MAXP, np, s4);   // tfqmr.py:88
        alpha = scal[S_RHO0 + par] / sigma;                                   // tfqmr.py:89
        const double theta = scal[S_THETA_B], eta = scal[S_ETA_B];
        c1 = theta * theta * eta / alpha;                                     // tfqmr.py:93
        if (lead) scal[S_ALPHA] = alpha;
        return false;
    }
    __device__ bool skip() const { return false; }
    __device__ void elem(double uv, double yv, double &wv, double &dv, double *acc) {
        wv = wv - alpha * uv;                                                 // tfqmr.py:92
        dv = dv * c1;                                                         // tfqmr.py:93
        dv = dv + yv;                                                         // tfqmr.py:94
        acc[0] += wv * wv;                                                    // tfqmr.py:95
    }
    __device__ void pair(int64_t i, double *acc) {
        const double2 uv = mk_ld2(u, i), yv = mk_ld2(y, i);
        double2 wv = mk_ld2(w, i), dv = mk_ld2(d, i);
        elem(uv.x, yv.x, wv.x, dv.x, acc);
        elem(uv.y, yv.y, wv.y, dv.y, acc);
        mk_st2(w, i, wv);
        mk_st2(d, i, dv);
    }
    __device__ void one(int64_t i, double *acc) {
        double wv = w[i], dv = d[i];
        elem(u[i], y[i], wv, dv, acc);
        w[i] = wv;
        d[i] = dv;
    }
};

// theta, c, tau (residNorm), eta of one half-step (tfqmr.py:95-98 / :118-121)
__device__ __forceinline__ void half_step(double ww, double tau_in, double alpha, double *theta, double *tau,
                                          double *eta) {
    *theta = __dsqrt_rn(ww) / tau_in;
    const double c = 1.0 / __dsqrt_rn(1 + *theta * *theta);
    *tau = tau_in * (*theta * c);
    *eta = c * c * alpha;
}

struct OpP3 {
    static constexpr int NACC = 0, SLOT0 = 0;
    const double *part;
    int np;
    double *scal;
    MkStatus *st;
    int64_t matvec_max, nmv;
    double k;
    const double *d, *v;
    double *x, *y;
    const double *dg;                                                         // preconditioner diagonal or null
    double *z;                                                                // z = precon * y (only with dg)
    double alpha, eta;
    bool fin;
    __device__ bool prologue(double *s4, bool lead) {
        const double ww = mk_total(part + SLOT_WW * MK_MAXP, np, s4);
        double theta, tau;
        alpha = scal[S_ALPHA];
        half_step(ww, scal[S_TAU_B], alpha, &theta, &tau, &eta);
        const double m = 2.0 * k - 1.0;                                       // tfqmr.py:100
        fin = (tau * __dsqrt_rn(m + 1) < scal[S_THRESH]) || (nmv >= matvec_max);   // tfqmr.py:101
        if (lead) {
            scal[S_THETA_A] = theta;
            scal[S_ETA_A] = eta;
            scal[S_TAU_A] = tau;
            scal[S_M] = m;                     // odd m tells finish() that set A holds the final values
        }
        return fin;
    }
    __device__ bool skip() const { return false; }
    __device__ void pair(int64_t i, double *) {
        const double2 dv = mk_ld2(d, i);
        double2 xv = mk_ld2(x, i);
        xv.x = xv.x + eta * dv.x;                                             // tfqmr.py:99
        xv.y = xv.y + eta * dv.y;
        mk_st2(x, i, xv);
        if (!fin) {
            const double2 vv = mk_ld2(v, i);
            double2 yv = mk_ld2(y, i);
            yv.x = yv.x - alpha * vv.x;                                       // tfqmr.py:107
            yv.y = yv.y - alpha * vv.y;
            mk_st2(y, i, yv);
            if (dg) {                                                         // tfqmr.py:109-110
                const double2 gv = mk_ld2(dg, i);
                yv.x = gv.x * yv.x;
                yv.y = gv.y * yv.y;
                mk_st2(z, i, yv);
            }
        }
    }
    __device__ void one(int64_t i, double *) {
        x[i] = x[i] + eta * d[i];
        if (!fin) {
            const double yv = y[i] - alpha * v[i];
            y[i] = yv;
            if (dg) z[i] = dg[i] * yv;
        }
    }
};

struct EpiP4 {   // u = A y ; w -= alpha u ; d = c d + y ; <w,w>, <r0,w>
    static constexpr int NACC = 2, SLOT0 = SLOT_WW;
    const double *scal;
    const double *y, *r0;
    double *u, *w, *d;
    double alpha, c1;
    __device__ void prologue(double *) {
        alpha = scal[S_ALPHA];
        c1 = scal[S_THETA_A] * scal[S_THETA_A] * scal[S_ETA_A] / alpha;       // tfqmr.py:116
    }
    __device__ double xin(double x) const { return x; }
    __device__ void row(int64_t i, double s, double *acc) {
        u[i] = s;
        const double wv = w[i] - alpha * s;                                   // tfqmr.py:115
        w[i] = wv;
        double dv = d[i] * c1;                                                // tfqmr.py:116
        dv = dv + y[i];                                                       // tfqmr.py:117
        d[i] = dv;
        acc[0] += wv * wv;                                                    // tfqmr.py:118
        acc[1] += r0[i] * wv;                                                 // tfqmr.py:128
    }
    static constexpr int NPF = 4;                         // pipelined kernels: w[i], d[i], y[i], r0[i] arrive as o[0..3]
    __device__ const double *pf_vec(int j) const { return j == 0 ? w : (j == 1 ? d : (j == 2 ? y : r0)); }
    __device__ void row_pf(int64_t i, double s, const double *o, double *acc) {
        u[i] = s;
        const double wv = o[0] - alpha * s;                                   // tfqmr.py:115
        w[i] = wv;
        double dv = o[1] * c1;                                                // tfqmr.py:116
        dv = dv + o[2];                                                       // tfqmr.py:117
        d[i] = dv;
        acc[0] += wv * wv;                                                    // tfqmr.py:118
        acc[1] += o[3] * wv;                                                  // tfqmr.py:128
    }
};

struct OpP5 {
    static constexpr int NACC = 0, SLOT0 = 0;
    const double *part;
    int np;
    double *scal;
    MkStatus *st;
    int par;
    int64_t matvec_max, nmv;
    double k;
    const double *d, *w, *u;
    double *x, *y, *v;
    const double *dg;                                                         // preconditioner diagonal or null
    double *z;                                                                // z = precon * y (only with dg)
    double eta, beta;
    bool fin;
    __device__ bool prologue(double *s4, bool lead) {
        const double ww = mk_total(part + SLOT_WW * MK_MAXP, np, s4);
        const double r0w = mk_total(part + SLOT_R0W * MK_MAXP, np, s4);
        double theta, tau;
        half_step(ww, scal[S_TAU_A], scal[S_ALPHA], &theta, &tau, &eta);
        const double m = 2.0 * k;                                             // tfqmr.py:105
        fin = (tau * __dsqrt_rn(m + 1) < scal[S_THRESH]) || (nmv >= matvec_max);   // tfqmr.py:123
        beta = r0w / scal[S_RHO0 + par];                                      // tfqmr.py:129
        if (lead) {
            scal[S_THETA_B] = theta;
            scal[S_ETA_B] = eta;
            scal[S_TAU_B] = tau;
            scal[S_M] = m;
            scal[S_RHO0 + (par ^ 1)] = r0w;                                   // tfqmr.py:130
            st->itn += 1;
        }
        return fin;
    }
    __device__ bool skip() const { return false; }
    __device__ void elem(double dv, double wv, double uv, double &xv, double &yv, double &vv) {
        xv = xv + eta * dv;                                                   // tfqmr.py:122
        if (fin) return;
        yv = yv * beta;                                                       // tfqmr.py:133
        yv = yv + wv;                                                         // tfqmr.py:134
        vv = vv * beta;                                                       // tfqmr.py:137
        vv = vv + uv;                                                         // tfqmr.py:138
        vv = vv * beta;                                                       // tfqmr.py:139
    }
    __device__ void pair(int64_t i, double *) {
        const double2 dv = mk_ld2(d, i);
        double2 xv = mk_ld2(x, i), wv{0, 0}, uv{0, 0}, yv{0, 0}, vv{0, 0};
        if (!fin) {
            wv = mk_ld2(w, i);
            uv = mk_ld2(u, i);
            yv = mk_ld2(y, i);
            vv = mk_ld2(v, i);
        }
        elem(dv.x, wv.x, uv.x, xv.x, yv.x, vv.x);
        elem(dv.y, wv.y, uv.y, xv.y, yv.y, vv.y);
        mk_st2(x, i, xv);
        if (!fin) {
            mk_st2(y, i, yv);
            mk_st2(v, i, vv);
            if (dg) {                                                         // tfqmr.py:142-143
                const double2 gv = mk_ld2(dg, i);
                yv.x = gv.x * yv.x;
                yv.y = gv.y * yv.y;
                mk_st2(z, i, yv);
            }
        }
    }
    __device__ void one(int64_t i, double *) {
        double xv = x[i], yv = 0, vv = 0, wv = 0, uv = 0;
        if (!fin) {
            wv = w[i];
            uv = u[i];
            yv = y[i];
            vv = v[i];
        }
        elem(d[i], wv, uv, xv, yv, vv);
        x[i] = xv;
        if (!fin) {
            y[i] = yv;
            v[i] = vv;
            if (dg) z[i] = dg[i] * yv;
        }
    }
};

template <bool INIT>
struct EpiP6 {   // u = A y ; v += u (INIT: v = u) ; <r0, v>
    static constexpr int NACC = 1, SLOT0 = SLOT_SIGMA;
    const double *r0;
    double *u, *v;
    __device__ void prologue(double *) {}
    __device__ double xin(double x) const { return x; }
    __device__ void row(int64_t i, double s, double *acc) {
        u[i] = s;
        const double vv = INIT ? s : v[i] + s;                                // tfqmr.py:83 / :150
        v[i] = vv;
        acc[0] += r0[i] * vv;                                                 // tfqmr.py:88
    }
    static constexpr int NPF = 2;                         // pipelined kernels: r0[i], v[i] arrive as o[0], o[1] (INIT: v unused)
    __device__ const double *pf_vec(int j) const { return j == 0 ? r0 : v; }
    __device__ void row_pf(int64_t i, double s, const double *o, double *acc) {
        u[i] = s;
        const double vv = INIT ? s : o[1] + s;                                // tfqmr.py:83 / :150
        v[i] = vv;
        acc[0] += o[0] * vv;                                                  // tfqmr.py:88
    }
};

__global__ __launch_bounds__(MK_BLOCK) void tfqmr_init_kernel(const double *part, int np, double *scal, MkStatus *st,
                                                              MkHalt halt, double abstol, double reltol,
                                                              int64_t matvec_max) {
    __shared__ double s4[4];
    const double rho = mk_total(part + SLOT_WW * MK_MAXP, np, s4);            // tfqmr.py:61
    if (threadIdx.x == 0) {
        const double resid0 = fabs(__dsqrt_rn(rho));
        const double rel = reltol * resid0;
        const double thresh = (rel > abstol) ? rel : abstol;
        scal[S_RHO0] = rho;
        scal[S_THRESH] = thresh;
        scal[S_RESID0] = resid0;
        scal[S_TAU_B] = resid0;
        scal[S_THETA_B] = 0.0;                                                // tfqmr.py:73-74
        scal[S_ETA_B] = 0.0;
        scal[S_M] = 0.0;        // the reference leaves m unbound when it never enters the loop (tfqmr.py:156)
        st->nMatvec = 0;
        halt.out((resid0 <= thresh) || (0 >= matvec_max));                    // tfqmr.py:67
    }
}

struct TfqmrSolver : mk_solver {
    double *d_x = nullptr, *d_r0 = nullptr, *d_y = nullptr, *d_w = nullptr, *d_d = nullptr, *d_u = nullptr,
           *d_v = nullptr, *d_z = nullptr;
    bool takes_precon() const override { return true; }
    double *zsrc() const { return d_prec ? d_z : d_y; }                        // what the products and `d += z` read

    int setup(const double *rhs, const double *guess) override {
        if (!d_x) {
            int rc;
            if ((rc = alloc_vec(&d_x, nx)) || (rc = alloc_vec(&d_r0, n)) || (rc = alloc_vec(&d_y, nx)) ||
                (rc = alloc_vec(&d_w, n)) || (rc = alloc_vec(&d_d, n)) || (rc = alloc_vec(&d_u, n)) ||
                (rc = alloc_vec(&d_v, n)))
                return rc;
        }
        if (d_prec && !d_z) {
            int rc = alloc_vec(&d_z, nx);
            if (rc) return rc;
        }
        if (guess) {
            MK_HIP(hipMemcpyAsync(d_x, guess, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream));
            int rc = exchange(d_x);
            if (rc != MK_OK) return rc;
            mk_launch_spmv(this, d_x, MkPlainEpi{d_u}, false);                // r0 = rhs - A x   tfqmr.py:58-59
            mk_launch_stream(this, MkOpSub{rhs, d_u, d_r0}, n);
        } else {
            MK_HIP(hipMemsetAsync(d_x, 0, sizeof(double) * (size_t)nx, stream));
            mk_launch_stream(this, MkOpCopy{rhs, d_r0}, n);
        }
        mk_launch_stream(this, MkOpDot<SLOT_WW>{d_r0, d_r0}, n);
        int rc = allreduce(SLOT_WW, 1);
        if (rc != MK_OK) return rc;
        hipLaunchKernelGGL(tfqmr_init_kernel, dim3(1), dim3(MK_BLOCK), 0, stream, d_part, np_stream, d_scal, d_status,
                           next_halt(), prm.abstol, prm.reltol, prm.matvec_max);
        mk_launch_stream(this, MkOpCopy{d_r0, d_y}, n);                        // y = r0.copy()    tfqmr.py:70
        mk_launch_stream(this, MkOpCopy{d_r0, d_w}, n);                        // w = r0.copy()    tfqmr.py:71
        MK_HIP(hipMemsetAsync(d_d, 0, sizeof(double) * (size_t)n, stream));    // d = 0            tfqmr.py:72
        if (d_prec) mk_launch_stream(this, MkOpMul{d_prec, d_r0, d_z}, n);     // z = precon * y   tfqmr.py:77-78
        if (precon_fn && host_precon(d_y, d_z) != MK_OK) return MK_ERR_STATE;
        if ((rc = exchange(zsrc())) != MK_OK) return rc;
        // u = A z ; v = u.copy() ; first sigma                               tfqmr.py:82-83
        mk_launch_spmv(this, zsrc(), EpiP6<true>{d_r0, d_u, d_v}, false, CountGate{d_status, 0});
        return allreduce(SLOT_SIGMA, 1);
    }

    int enqueue_spmv_only(int which) override {            // (timing aid: a product's kernel without its gate)
        double *zs = zsrc();
        if (which == 0) mk_launch_spmv(this, zs, EpiP4{d_scal, zs, d_r0, d_u, d_w, d_d, 0.0, 0.0}, false);
        else if (which == 1) mk_launch_spmv(this, zs, EpiP6<false>{d_r0, d_u, d_v}, false);
        else return mk_fail(MK_ERR_ARG, "TFQMR has two products per pass");
        return MK_OK;
    }

    int enqueue_pass() override {
        const int par = (int)(it & 1);
        const double k = (double)(it + 1);
        const int64_t nmv = 1 + 2 * it;          // products done when the pass starts
        int rc;
        double *zs = zsrc();
        mk_launch_stream(this, OpP2{d_part, np_spmv, d_scal, par, d_u, zs, d_w, d_d, 0.0, 0.0}, n);
        if ((rc = allreduce(SLOT_WW, 1)) != MK_OK) return rc;
        mk_launch_stream(this, OpP3{d_part, np_stream, d_scal, d_status, prm.matvec_max, nmv, k, d_d, d_v, d_x, d_y,
                                    d_prec, d_z, 0.0, 0.0, false}, n);
        if (precon_fn && (rc = host_precon(d_y, d_z)) != MK_OK) return rc;     // z = precon * y       tfqmr.py:109-110
        if ((rc = exchange(zs)) != MK_OK) return rc;
        mk_launch_spmv(this, zs, EpiP4{d_scal, zs, d_r0, d_u, d_w, d_d, 0.0, 0.0}, true, CountGate{d_status, nmv});
        if ((rc = allreduce(SLOT_WW, 2)) != MK_OK) return rc;
        mk_launch_stream(this, OpP5{d_part, np_spmv, d_scal, d_status, par, prm.matvec_max, nmv + 1, k, d_d, d_w, d_u,
                                    d_x, d_y, d_v, d_prec, d_z, 0.0, 0.0, false}, n);
        if (precon_fn && (rc = host_precon(d_y, d_z)) != MK_OK) return rc;     // z = precon * y       tfqmr.py:142-143
        if ((rc = exchange(zs)) != MK_OK) return rc;
        mk_launch_spmv(this, zs, EpiP6<false>{d_r0, d_u, d_v}, true, CountGate{d_status, nmv + 1});
        return allreduce(SLOT_SIGMA, 1);
    }

    int finish(mk_result *res) override {
        int rc = poll();
        if (rc != MK_OK) return rc;
        fill_result(res);
        const bool first_half = (((int64_t)h_scal[S_M]) & 1) != 0;
        const double tau = first_half ? h_scal[S_TAU_A] : h_scal[S_TAU_B];
        res->residNorm = tau;
        res->residNorm0 = h_scal[S_RESID0];
        res->threshold = h_scal[S_THRESH];
        res->converged = (tau * sqrt(h_scal[S_M] + 1) < h_scal[S_THRESH]) ? 1 : 0;   // tfqmr.py:156
        res->aux[0] = h_scal[S_M];
        return MK_OK;
    }

    const double *x() const override { return d_x; }
};

}  // namespace

mk_solver *mk_make_tfqmr() { return new TfqmrSolver(); }
