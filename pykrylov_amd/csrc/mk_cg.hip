// mk_cg.hip -- conjugate gradients, device resident.   Reference: pykrylov/cg/cg.py:46-165.
//
// One pass of the reference loop (cg.py:113-158) = three kernels:
//   K1  Ap = A p ; partial sums of <p, Ap>                          (cg.py:115-117)
//   K2  alpha = ry / pAp ; r += alpha Ap ; partial sums of <r, r>                  (cg.py:119-127,131,146)
//   K3  x += alpha p ; beta = ry' / ry ; p = beta p - r ; residNorm, history, loop test
//                                                                                  (cg.py:130,149-158, :113)
// The x update rides in K3 because K3 streams p anyway (x is not read by anything in between and K3 always
// runs when K2 did), which saves one pass over p.  Same operations on the same values: results are unchanged.
// Algorithmic traffic per pass: B_spmv + (16n read + 8n write) + (24n read + 16n write)
//   = B_spmv + 64n bytes (the reference's op count gives B_spmv + 104n).
//
// FUSED passes (round 5; storage format 9, one GPU): K3's vector work moves into the NEXT pass's product kernel --
//   K1f  x += alpha p_old ; p = beta p_old - r ; Ap = A p ; partial sums of <p, Ap>   (cg.py:130,150-151 of the pass before,
//        then cg.py:115-117): the brick march of mk_spmv_fmt9.h loads the own rows of p_old, r and x once, forms p and x,
//        writes p to the OTHER p buffer (neighbouring bricks still read p_old for their halo rows) and x in place
//   K2   unchanged
//   K3s  one workgroup: beta, residNorm, history, loop test (the prologue of K3, nothing else); marks the update pending
// -- same operations on the same values in the same order per element, so every bit of x, p, r and of the history is
// the bit of the three-kernel pass.  Traffic per pass: (48n read/write in K1f) + 24n instead of 16n + 24n + 40n: one
// sweep over p less.  The update of the LAST pass is applied when the loop has halted (mk_solver_x / _vector / finish:
// `materialize`); a caller that looks at x between passes gets x + alpha p formed into a scratch vector.
#include "mk_solver.h"

namespace {

enum { S_RY0 = 0, S_RY1 = 1, S_THRESH = 2, S_RESID = 3, S_RESID0 = 4, S_PAP = 5, S_ALPHA = 6, S_BETA = 7, S_PENDING = 8 };

template <bool NTY>
struct CgSpmvEpiT {
    static constexpr int NACC = 1, SLOT0 = 0;
    static constexpr bool SYM_MARCH = true;                  // (CG's matrices are symmetric: format 11, mk_device.h MkSymMarch)
    const double *p;
    double *Ap;
    double pr;
    __device__ void prologue(double *) {}
    __device__ double xin(double v) const { return v; }
    __device__ void pre(int64_t r) { pr = p[r]; }          // issued at the top of the tile
    __device__ void row(int64_t r, double s, double *acc) {
        if constexpr (NTY) __builtin_nontemporal_store(s, Ap + r);
        else Ap[r] = s;
        acc[0] += pr * s;
    }
    // (NTY: the product vector goes past the caches -- mk_store_nt, mk_device.h)
    // p is the product's input vector: where the kernel has p[r] at hand (pattern format: the tile's LDS window) it
    // passes it instead of `pre` loading it again -- one stream of n doubles less per product
    __device__ void row_x(int64_t r, double s, double xr, double *acc) {
        if constexpr (NTY) __builtin_nontemporal_store(s, Ap + r);
        else Ap[r] = s;
        acc[0] += xr * s;
    }
    // the brick march on a general geometry (mk_spmv_fmt9.h, GEN): rows r, r + 1 where they exist -- a row that does not is
    // stored into the dump row and adds +0.0 (the running sum is never -0.0: unchanged bit for bit)
    __device__ void row2_m(int64_t r, mk_d2 s, mk_d2 xr, bool oka, bool okb, double *dump, double *acc) {
        mk_d2u *d = reinterpret_cast<mk_d2u *>(okb ? Ap + r : dump);
        if constexpr (NTY) __builtin_nontemporal_store(s, d);
        else *d = s;
        if (oka && !okb) Ap[r] = s.x;
        const double ta = xr.x * s.x, tb = xr.y * s.y;
        acc[0] += oka ? ta : 0.0;
        acc[0] += okb ? tb : 0.0;
    }
};
using CgSpmvEpi = CgSpmvEpiT<false>;

// K1f: the brick march's "fuse" hooks (mk_spmv_fmt9.h).  Only ever launched on a format-9 matrix; the other formats'
// instantiations exist because the launcher is generic and are never run.
#ifndef MK_FUSE_NT_DEF
#define MK_FUSE_NT_DEF 7                                     // 1 nt loads of x, 2 nt stores of x, 4 nt stores of the new p (profiles/r05_fuse_nt_ab.txt: all three)
#endif
template <bool NTY>
struct CgFusedEpiT {
    static constexpr int NACC = 1, SLOT0 = 0;
    static constexpr bool SYM_MARCH = true;
    static constexpr int FUSE_NT = NTY ? MK_FUSE_NT_DEF : 0;  // vectors beyond the Infinity Cache: x / the new p past the caches too
    double *Ap;
    const double *fuse_r;
    double *fuse_x, *fuse_p, *fuse_dump;
    double *scal;
    double fa, fb;
    __device__ void prologue(double *) {
        fa = scal[S_ALPHA];                                // alpha and beta of the pass before (K2 / K3s)
        fb = scal[S_BETA];
        if (blockIdx.x == 0 && threadIdx.x == 0) scal[S_PENDING] = 0.0;     // this kernel applies the pending update
    }
    __device__ double xin(double v) const { return v; }
    __device__ double fuse_pnew(double po, double rv) const { return fb * po - rv; }       // cg.py:150-151
    __device__ double fuse_xnew(double xv, double po) const { return xv + fa * po; }       // cg.py:130
    __device__ void row(int64_t r, double s, double *) { Ap[r] = s; }                      // (never run: see above)
    __device__ void row_x(int64_t r, double s, double xr, double *acc) {
        if constexpr (NTY) __builtin_nontemporal_store(s, Ap + r);
        else Ap[r] = s;
        acc[0] += xr * s;
    }
    __device__ void row2_m(int64_t r, mk_d2 s, mk_d2 xr, bool oka, bool okb, double *dump, double *acc) {   // (as CgSpmvEpiT's)
        mk_d2u *d = reinterpret_cast<mk_d2u *>(okb ? Ap + r : dump);
        if constexpr (NTY) __builtin_nontemporal_store(s, d);
        else *d = s;
        if (oka && !okb) Ap[r] = s.x;
        const double ta = xr.x * s.x, tb = xr.y * s.y;
        acc[0] += oka ? ta : 0.0;
        acc[0] += okb ? tb : 0.0;
    }
};

// K3s: the scalar part of K3 (CgUpdateXP::prologue), one workgroup
__global__ __launch_bounds__(MK_BLOCK) void cg_beta_kernel(const double *part, int np, double *scal, MkStatus *st, double *hist,
                                                           int par, int64_t matvec_max, MkHalt halt) {
    __shared__ double s4[4];
    if (halt.in()) {
        if (threadIdx.x == 0) halt.out(true);
        return;
    }
    const double ry_next = mk_total(part + MK_MAXP, np, s4);
    if (threadIdx.x == 0) {
        const double ry = scal[S_RY0 + par];
        const double beta = ry_next / ry;                 // cg.py:149
        const double resid = fabs(__dsqrt_rn(ry_next));   // cg.py:154
        const bool go = (resid > scal[S_THRESH]) && (st->nMatvec < matvec_max);   // cg.py:113
        scal[S_RY0 + (par ^ 1)] = ry_next;                // cg.py:153
        scal[S_RESID] = resid;
        scal[S_BETA] = beta;
        scal[S_PENDING] = 1.0;                            // x += alpha p ; p = beta p - r of this pass are still to be applied
        hist[st->hist_len % MK_HIST_RING] = resid;        // cg.py:155
        st->hist_len += 1;
        st->itn += 1;
        halt.out(!go);
    }
}

// the pending update of the last pass, outside the halt protocol: in place once the loop has halted, into scratch
// vectors for a caller that looks at the iterate between passes
__global__ __launch_bounds__(MK_BLOCK) void cg_flush_kernel(int64_t n, const double *scal, const double *r, const double *p,
                                                            const double *x, double *p_out, double *x_out) {
    const bool pend = scal[S_PENDING] != 0.0;
    const double alpha = scal[S_ALPHA], beta = scal[S_BETA];
    for (int64_t i = (int64_t)blockIdx.x * MK_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * MK_BLOCK) {
        const double pv = p[i], xv = x[i];
        x_out[i] = pend ? xv + alpha * pv : xv;           // cg.py:130
        p_out[i] = pend ? beta * pv - r[i] : pv;          // cg.py:150-151
    }
}

struct CgUpdateR {
    static constexpr int NACC = 1, SLOT0 = 1;
    const double *part;
    int np;
    double *scal;
    MkStatus *st;
    int par, check_curv;
    const double *Ap;
    double *r;
    const double *dg;                                     // preconditioner diagonal or null
    double alpha;
    bool bad;
    MkTotalRegs tr;
    double ry_in;
    __device__ void early() {
        mk_total_issue(part, np, tr);
        ry_in = scal[S_RY0 + par];
    }
    __device__ bool prologue(double *s4, bool lead) {
        const double pAp = mk_total_finish(tr, np, s4);
        const double ry = ry_in;
        bad = check_curv && (pAp <= 0.0);                 // cg.py:119-124
        alpha = ry / pAp;                                 // cg.py:127
        if (lead) {
            st->nMatvec += 1;                             // cg.py:116
            scal[S_PAP] = pAp;
            scal[S_ALPHA] = alpha;
            if (bad) st->definite = 0;
        }
        return bad;
    }
    __device__ bool skip() const { return bad; }
    struct Regs {
        double2 av, rv, dv;
    };
    __device__ void load2(int64_t i, Regs &g) const {
        g.av = mk_ld2(Ap, i);
        g.rv = mk_ld2(r, i);
        if (dg) g.dv = mk_ld2(dg, i);
    }
    __device__ void apply2(Regs &g, double *acc) const {
        g.rv.x = g.rv.x + alpha * g.av.x;                 // cg.py:131
        g.rv.y = g.rv.y + alpha * g.av.y;
        if (dg) {                                         // y = precon * r ; <r, y>   cg.py:137-138,146
            acc[0] += g.rv.x * (g.dv.x * g.rv.x);
            acc[0] += g.rv.y * (g.dv.y * g.rv.y);
        } else {
            acc[0] += g.rv.x * g.rv.x;                    // cg.py:146
            acc[0] += g.rv.y * g.rv.y;
        }
    }
    __device__ void store2(int64_t i, const Regs &g) const { mk_st2(r, i, g.rv); }
    __device__ void one(int64_t i, double *acc) {
        const double rv = r[i] + alpha * Ap[i];
        r[i] = rv;
        acc[0] += dg ? rv * (dg[i] * rv) : rv * rv;
    }
};

struct CgUpdateXP {
    static constexpr int NACC = 0, SLOT0 = 0;
    const double *part;
    int np;
    double *scal;
    MkStatus *st;
    double *hist;
    int par;
    int64_t matvec_max;
    const double *r;
    double *p, *x;
    double alpha, beta;
    MkTotalRegs tr;
    double ry_in, thresh_in;
    int64_t nmv_in;
    __device__ void early() {
        mk_total_issue(part + MK_MAXP, np, tr);
        ry_in = scal[S_RY0 + par];
        alpha = scal[S_ALPHA];                            // written by K2 of this pass
        thresh_in = scal[S_THRESH];
        nmv_in = st->nMatvec;
    }
    __device__ bool prologue(double *s4, bool lead) {
        const double ry_next = mk_total_finish(tr, np, s4);
        const double ry = ry_in;
        beta = ry_next / ry;                              // cg.py:149
        const double resid = fabs(__dsqrt_rn(ry_next));   // cg.py:154
        const bool go = (resid > thresh_in) && (nmv_in < matvec_max);   // cg.py:113
        if (lead) {
            scal[S_RY0 + (par ^ 1)] = ry_next;            // cg.py:153
            scal[S_RESID] = resid;
            hist[st->hist_len % MK_HIST_RING] = resid;    // cg.py:155
            st->hist_len += 1;
            st->itn += 1;
        }
        return !go;
    }
    __device__ bool skip() const { return false; }
    struct Regs {
        double2 rv, pv, xv;
    };
    __device__ void load2(int64_t i, Regs &g) const {
        g.rv = mk_ld2(r, i);
        g.pv = mk_ld2(p, i);
        g.xv = mk_ld2(x, i);
    }
    __device__ void apply2(Regs &g, double *) const {
        g.xv.x = g.xv.x + alpha * g.pv.x;                 // cg.py:130 (with the p of this pass)
        g.xv.y = g.xv.y + alpha * g.pv.y;
        g.pv.x = beta * g.pv.x - g.rv.x;                  // cg.py:150-151 (p *= beta; p -= r)
        g.pv.y = beta * g.pv.y - g.rv.y;
    }
    __device__ void store2(int64_t i, const Regs &g) const {
        mk_st2(x, i, g.xv);
        mk_st2(p, i, g.pv);
    }
    __device__ void one(int64_t i, double *) {
        const double pv = p[i];
        x[i] = x[i] + alpha * pv;
        p[i] = beta * pv - r[i];
    }
};

// everything between the initial <r, y> and the loop header (cg.py:99-113)
__global__ __launch_bounds__(MK_BLOCK) void cg_init_kernel(const double *part, int np, double *scal, MkStatus *st,
                                                           double *hist, MkHalt halt, double abstol, double reltol,
                                                           int64_t matvec_max, int64_t nmv0) {
    __shared__ double s4[4];
    const double ry = mk_total(part + MK_MAXP, np, s4);
    if (threadIdx.x == 0) {
        const double resid0 = fabs(__dsqrt_rn(ry));
        const double rel = reltol * resid0;
        const double thresh = (rel > abstol) ? rel : abstol;   // max(abstol, reltol*residNorm0)
        scal[S_RY0] = ry;
        scal[S_THRESH] = thresh;
        scal[S_RESID] = resid0;
        scal[S_RESID0] = resid0;
        hist[0] = resid0;
        st->hist_len = 1;
        st->nMatvec = nmv0;
        st->itn = 0;
        st->definite = 1;
        scal[S_PENDING] = 0.0;
        halt.out(!((resid0 > thresh) && (nmv0 < matvec_max)));
    }
}

struct CgSolver : mk_solver {
    double *d_x = nullptr, *d_r = nullptr, *d_p = nullptr, *d_Ap = nullptr;
    // fused passes (see the top of the file)
    bool fused = false;
    double *d_p2 = nullptr, *d_dump = nullptr;            // the second p buffer; the dump rows of the brick march
    mutable double *d_xv = nullptr, *d_pv = nullptr;      // the iterate / direction as a caller sees them between passes
    mutable bool flushed = false;
    int64_t nmv0 = 0;                                     // products before the loop (the initial guess's)
    double *pbuf(int64_t k) const { return (!fused || (k & 1) == 0) ? d_p : d_p2; }     // p of pass k
    // the buffer of the direction in use: product k WROTE pbuf(k), and the passes the host enqueued behind a halt never ran,
    // so the products the DEVICE counted decide (K2 counts one per product, cg.py:116), not the host's pass counter
    double *pcur() const {
        CgSolver *me = const_cast<CgSolver *>(this);
        if (!fused || !is_setup) return d_p;
        if (hipMemcpyAsync(me->h_status, d_status, sizeof(MkStatus), hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipStreamSynchronize(stream) != hipSuccess)
            return d_p;
        const int64_t done = h_status->nMatvec - nmv0;
        return done > 0 ? pbuf(done - 1) : d_p;
    }

    static bool want_fuse() {
        const char *e = getenv("MK_CG_FUSE");              // (read at every setup: tests switch it between solves)
        return !e || atoi(e) != 0;
    }
    // apply (halted) or form (running) the pending x / p update; returns the vectors a caller may read
    void materialize(const double **xo, const double **po) const {
        CgSolver *me = const_cast<CgSolver *>(this);
        *xo = d_x;
        double *pc = pcur();
        *po = pc;
        if (!fused || it == 0 || !is_setup) return;
        const int grid = (int)((n + MK_BLOCK - 1) / MK_BLOCK > 2048 ? 2048 : (n + MK_BLOCK - 1) / MK_BLOCK);
        if (halted) {
            if (!flushed) {
                hipLaunchKernelGGL(cg_flush_kernel, dim3(grid), dim3(MK_BLOCK), 0, stream, n, d_scal, d_r, pc, d_x, pc, d_x);
                hipMemsetAsync(d_scal + S_PENDING, 0, sizeof(double), stream);
                me->flushed = true;
            }
            return;
        }
        if (!d_xv && (me->alloc_vec(&me->d_xv, n) != MK_OK || me->alloc_vec(&me->d_pv, n) != MK_OK)) return;
        hipLaunchKernelGGL(cg_flush_kernel, dim3(grid), dim3(MK_BLOCK), 0, stream, n, d_scal, d_r, pc, d_x, d_pv, d_xv);
        *xo = d_xv;
        *po = d_pv;
    }

    int setup(const double *rhs, const double *guess) override {
        if (!d_x) {
            int rc;
            // (r with room for received entries too: fused passes on a slab exchange r's boundary planes, not p's)
            if ((rc = alloc_vec(&d_x, nx)) || (rc = alloc_vec(&d_r, nx)) || (rc = alloc_vec(&d_p, nx)) ||
                (rc = alloc_vec(&d_Ap, n)))
                return rc;
        }
        const MkPlan *plan = A ? mk_csr_plan(A) : nullptr;
        // one device, or one rank's slab under a halo exchange (the march takes the neighbours' planes from the received entries)
        const bool slab = plan && A->ex.mode == 0 && nx == n + A->ex.n_halo && (plan->pen_xlo >= 0 || plan->pen_xhi >= 0);
        fused = want_fuse() && plan && mk_fmt_march(plan->fmt) && !precon_fn && A->nops == 0 && !A->comp_kind &&
                ((!mk_comm_active() && A->ex.mode < 0 && nx == n) || slab);
        flushed = false;
        if (fused && !d_p2 && (alloc_vec(&d_p2, nx) != MK_OK || alloc_vec(&d_dump, (int64_t)MK_MAXP * 1024) != MK_OK)) fused = false;
        if (mk_comm_active()) {
            // The choice changes what travels: fused ranks exchange r's boundary planes and form the neighbours' p themselves,
            // the others exchange p -- messages of the same size, so a disagreement would neither hang nor fail, it would
            // multiply by the wrong vector.  (Ranks can differ: a slab one plane thinner falls below the march's row
            // threshold, an allocation fails on one of them.)  So all ranks fuse, or none does.
            double veto = fused ? 0.0 : 1.0;
            int rc = mk_comm_allreduce_host(&veto, 1);
            if (rc != MK_OK) return rc;
            if (veto != 0.0) fused = false;
        }
        nmv0 = 0;
        mk_launch_stream(this, MkOpNegCopy{rhs, d_r}, n);                   // r = -rhs          cg.py:85
        if (guess) {
            MK_HIP(hipMemcpyAsync(d_x, guess, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream));
            int rc = exchange(d_x);
            if (rc != MK_OK) return rc;
            mk_launch_spmv(this, d_x, MkPlainEpi{d_Ap}, false);             // r += A x          cg.py:86-88
            mk_launch_stream(this, MkOpAddTo<1>{d_Ap, d_r}, n);
            nmv0 = 1;
        } else {
            MK_HIP(hipMemsetAsync(d_x, 0, sizeof(double) * (size_t)nx, stream));
        }
        if (d_prec) {
            mk_launch_stream(this, MkOpMul{d_prec, d_r, d_Ap}, n);          // y = precon * r    cg.py:91-92
            if (precon_fn && host_precon(d_r, d_Ap) != MK_OK) return MK_ERR_STATE;
            mk_launch_stream(this, MkOpDot<1>{d_r, d_Ap}, n);               // ry = <r, y>       cg.py:99
        } else {
            mk_launch_stream(this, MkOpDot<1>{d_r, d_r}, n);                // ry = <r, r>       cg.py:99
        }
        int rc = allreduce(1, 1);
        if (rc != MK_OK) return rc;
        hipLaunchKernelGGL(cg_init_kernel, dim3(1), dim3(MK_BLOCK), 0, stream, d_part, np_stream, d_scal, d_status,
                           d_hist, next_halt(), prm.abstol, prm.reltol, prm.matvec_max, nmv0);
        mk_launch_stream(this, MkOpNegCopy{d_r, d_p}, n);                   // p = -r            cg.py:104
        return MK_OK;
    }

    int enqueue_spmv_only(int which) override {
        if (which != 0) return mk_fail(MK_ERR_ARG, "CG has one product per pass");
        if (fused && it > 0) {
            // the fused kernel exactly as a pass launches it (same streams: p_old, r, x in; p, x, Ap out), except that p goes
            // to the buffer it came from's twin and x is updated again with the same alpha -- timing only, after the run
            double *pc = pcur(), *po = (pc == d_p) ? d_p2 : d_p;
            if (mk_store_nt(A)) mk_launch_spmv(this, pc, CgFusedEpiT<true>{d_Ap, d_r, d_x, po, d_dump, d_scal, 0.0, 0.0}, false);
            else mk_launch_spmv(this, pc, CgFusedEpiT<false>{d_Ap, d_r, d_x, po, d_dump, d_scal, 0.0, 0.0}, false);
            return MK_OK;
        }
        if (A && mk_store_nt(A)) mk_launch_spmv(this, d_p, CgSpmvEpiT<true>{d_p, d_Ap, 0.0}, false);
        else mk_launch_spmv(this, d_p, CgSpmvEpi{d_p, d_Ap, 0.0}, false);
        return MK_OK;
    }

    int enqueue_pass() override {
        const int par = (int)(it & 1);
        int rc;
        if (fused) {
            const bool nt = mk_store_nt(A);
            if (it == 0) {                                  // nothing pending yet: the plain product on p = -r
                if ((rc = exchange(d_p)) != MK_OK) return rc;
                if (nt) mk_launch_spmv(this, d_p, CgSpmvEpiT<true>{d_p, d_Ap, 0.0});
                else mk_launch_spmv(this, d_p, CgSpmvEpi{d_p, d_Ap, 0.0});
            } else {                                        // K1f: p_old = the other buffer
                // a slab: the neighbours' planes of p are FORMED here like the own ones, from their p_old (kept behind the own rows
                // of the p buffers since the pass before) and their r -- so it is r whose boundary planes travel, as soon as K2 of
                // the pass before has written them; the interior planes' launch overlaps the messages as everywhere
                if ((rc = exchange(d_r)) != MK_OK) return rc;
                if (nt) mk_launch_spmv(this, pbuf(it - 1), CgFusedEpiT<true>{d_Ap, d_r, d_x, pbuf(it), d_dump, d_scal, 0.0, 0.0});
                else mk_launch_spmv(this, pbuf(it - 1), CgFusedEpiT<false>{d_Ap, d_r, d_x, pbuf(it), d_dump, d_scal, 0.0, 0.0});
            }
            if ((rc = allreduce(0, 1)) != MK_OK) return rc;
            mk_launch_stream(this, CgUpdateR{d_part, np_spmv, d_scal, d_status, par, prm.check_curvature, d_Ap, d_r,
                                             d_prec, 0.0, false}, n);
            if ((rc = allreduce(1, 1)) != MK_OK) return rc;
            hipLaunchKernelGGL(cg_beta_kernel, dim3(1), dim3(MK_BLOCK), 0, stream, d_part, np_stream, d_scal, d_status, d_hist,
                               par, prm.matvec_max, next_halt());
            return MK_OK;
        }
        rc = exchange(d_p);
        if (rc != MK_OK) return rc;
        if (A && mk_store_nt(A)) mk_launch_spmv(this, d_p, CgSpmvEpiT<true>{d_p, d_Ap, 0.0});
        else mk_launch_spmv(this, d_p, CgSpmvEpi{d_p, d_Ap, 0.0});
        if ((rc = allreduce(0, 1)) != MK_OK) return rc;
        mk_launch_stream(this, CgUpdateR{d_part, np_spmv, d_scal, d_status, par, prm.check_curvature, d_Ap, d_r,
                                         d_prec, 0.0, false}, n);
        if (precon_fn) {                                    // y = precon * r ; <r, y> re-formed   cg.py:137-138,146
            if ((rc = host_precon(d_r, d_Ap)) != MK_OK) return rc;
            mk_launch_stream(this, MkOpDot<1>{d_r, d_Ap}, n);
        }
        if ((rc = allreduce(1, 1)) != MK_OK) return rc;
        mk_launch_stream(this, CgUpdateXP{d_part, np_stream, d_scal, d_status, d_hist, par, prm.matvec_max, d_r, d_p,
                                          d_x, 0.0, 0.0}, n);
        return MK_OK;
    }

    int finish(mk_result *res) override {
        int rc = poll();
        if (rc != MK_OK) return rc;
        fill_result(res);
        res->residNorm = h_scal[S_RESID];
        res->residNorm0 = h_scal[S_RESID0];
        res->threshold = h_scal[S_THRESH];
        res->converged = (h_scal[S_RESID] <= h_scal[S_THRESH]) ? 1 : 0;     // cg.py:161
        res->aux[0] = h_scal[S_PAP];
        return MK_OK;
    }

    const double *x() const override {
        const double *xo, *po;
        materialize(&xo, &po);
        return xo;
    }
    const double *vector(int i) const override {
        if (i == 0) return d_r;
        if (i != 1) return nullptr;
        const double *xo, *po;
        materialize(&xo, &po);
        return po;
    }
    bool is_fused() const override { return fused; }
    // The search direction is built from r, not from y = precon*r, exactly as cg.py:104,150-151 do.
    bool takes_precon() const override { return true; }
};

}  // namespace

mk_solver *mk_make_cg() { return new CgSolver(); }
