// mk_solver.h -- host-side driver shared by all device-resident solver loops.
//
// A solver is a fixed sequence of kernels per pass of the reference's `while` body.  The
// host only enqueues; scalars (alpha, beta, rotations, norms, stop tests) live in device
// memory and are recomputed by every workgroup from the previous kernel's partial sums, so
// there is no device->host round trip inside an iteration.  The driver enqueues a batch of
// iterations, then reads back one small status record to learn whether the loop condition
// failed (kernels after that point are no-ops thanks to MkHalt).
#pragma once
#include "mk_device.h"

struct MkStatus {            // lives in device memory, written by one lane only
    int64_t nMatvec;
    int64_t itn;
    int64_t hist_len;
    int32_t istop;
    int32_t definite;
    int32_t converged;
    int32_t pad;
};

constexpr int64_t MK_HIST_RING = 1 << 16;   // residual-history ring (drained after every batch)
constexpr int64_t MK_BATCH_MAX = 256;

// Collective hooks (implemented in mk_comm.hip; no-ops without a communicator).
int mk_comm_active();
int mk_comm_allreduce_sum(double *buf_dev, int64_t count, hipStream_t stream);
int mk_comm_reduce_scatter_sum(const double *full_dev, double *mine_dev, int64_t count_per_rank, hipStream_t stream);
int mk_comm_allgather(const double *mine_dev, double *full_dev, int64_t count_per_rank, hipStream_t stream);
int mk_exchange_begin(const mk_csr *A, double *x_ext);     // may leave the messages in flight on a second stream
int mk_exchange_wait(const mk_csr *A, hipStream_t stream); // ... until here

struct mk_solver {
    const mk_csr *A = nullptr;
    const mk_csr *At = nullptr;     // transposed matrix (least-squares solvers only)
    const double *d_prec = nullptr; // diagonal of a Jacobi-type preconditioner M^-1 (borrowed, n entries) or null
    // general preconditioner through a host callback (mk_solver_set_precon_callback): the kernels run with a
    // diagonal of ones (`1.0 * v` is exact) and every preconditioned vector is replaced by the callback's result
    // right after the kernel that produced it; inner products with it are re-formed by a separate dot kernel
    mk_precon_fn precon_fn = nullptr;
    void *precon_user = nullptr;
    double *d_ones = nullptr, *h_pin = nullptr, *h_pout = nullptr;
    // ... or through a DEVICE operator (mk_solver_set_precon_csr: a sparse approximate inverse, e.g. the inverted
    // diagonal blocks of block-Jacobi, as a device matrix or composite): the same sites, the product stays in HBM
    const mk_csr *precon_op = nullptr;
    double *d_ptmp = nullptr;       // product target when a site preconditions a vector in place
    int *d_nohalt = nullptr;        // two zero words: the halt input of a product that must run after the loop has ended
    int host_precon(const double *in_dev, double *out_dev, bool force = false);   // out = precon * in ; unless `force`, a no-op once the loop has halted
    mk_params prm{};
    int64_t n = 0;        // local rows = length of every solver vector
    int64_t nx = 0;       // length of vectors that feed an SpMV (n + halo)
    hipStream_t stream = nullptr;

    double *d_scal = nullptr;       // MK_NSCAL
    double *d_part = nullptr;       // MK_NDOT * MK_MAXP
    int *d_halt = nullptr;          // 2
    MkStatus *d_status = nullptr;
    double *d_hist = nullptr;       // 2 * MK_HIST_RING: channel 0 = residual history, channel 1 = solver specific
    MkStatus *h_status = nullptr;   // pinned
    double *h_scal = nullptr;       // pinned, MK_NSCAL
    std::vector<double *> vecs;     // owned device vectors
    std::vector<double *> arena_vecs;   // ... carved from the context's vector arena (mk_arena_reserve): returned, not freed
    std::vector<double> hist;       // drained history (host)
    std::vector<double> hist2;      // second channel (MINRES: direct-error estimates)
    bool use_hist2 = false;
    int64_t hist_drained = 0;

    int64_t q = 0;                  // kernels launched so far (halt parity)
    int64_t it = 0;                 // loop passes enqueued so far
    bool is_setup = false, halted = false;
    bool counted_user = false;      // this solver is in its matrix's solver_users count (mk_csr_march_pref)
    int np_spmv = 1, np_stream = 1; // partial counts consumers must add up

    // timing
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double last_iterate_ms = 0.0;
    std::vector<hipEvent_t> spmv_ev;    // pairs
    int64_t spmv_timed = 0;
    double spmv_ms = 0.0;
    int spmv_sample_stride = 0;         // 0 = no per-kernel events

    virtual ~mk_solver();
    virtual int setup(const double *rhs, const double *guess) = 0;
    virtual int enqueue_pass() = 0;                 // one pass of the reference loop body
    virtual int finish(mk_result *res) = 0;
    virtual const double *x() const = 0;
    virtual const double *vector(int) const { return nullptr; }
    virtual bool takes_precon() const { return false; }
    virtual bool is_fused() const { return false; }   // CG on a format-9 matrix: the x / p update rides in the next product kernel
    // enqueue only the solver's (fused) SpMV kernel, exactly as a loop pass launches it; used to time
    // that kernel back to back.  Overwrites the product vector and its partial sums.
    // `which`: 0 = the first product of a pass, 1 = the second (BiCGSTAB / CGS / TFQMR: the product on z; least squares:
    // the A' u product with its fused v update); run without its gate.
    virtual int enqueue_spmv_only(int which = 0) { return mk_fail(MK_ERR_UNSUPPORTED, "SpMV timing is not wired for this solver"); }

    int init_common(const mk_csr *A_, const mk_params *p);
    int alloc_vec(double **out, int64_t len);
    MkHalt next_halt() { return MkHalt{d_halt, (int)(q++ & 1), mk_comm_active() ? MK_MAXP : 0}; }
    int poll();                                     // status + scalars + history -> host
    int iterate(int64_t max_iters, int64_t *done);
    int allreduce(int slot0, int nslots);           // partial sums across ranks (multi-GPU only)
    int exchange(double *x_ext);                    // halo / allgather before an SpMV
    void fill_result(mk_result *res) const;
    // SpMV timing helpers
    void spmv_begin();
    void spmv_end();
    int collect_spmv_timing();
};

mk_solver *mk_make_cg();
mk_solver *mk_make_bicgstab();
mk_solver *mk_make_cgs();
mk_solver *mk_make_tfqmr();
mk_solver *mk_make_minres();
mk_solver *mk_make_symmlq();
mk_solver *mk_make_lls(int kind);
int mk_lls_set_metric(mk_solver *s, const double *dm, const double *dn);
int mk_lls_set_callbacks(mk_solver *s, mk_precon_fn fn_m, void *user_m, mk_precon_fn fn_n, void *user_n);

#ifdef __HIPCC__
// ------------------------------------------------------------------ small shared kernels
struct MkOpNegCopy {            // out = -in                    (cg.py:85,104)
    static constexpr int NACC = 0, SLOT0 = 0;
    const double *in;
    double *out;
    __device__ bool prologue(double *, bool) { return false; }
    __device__ bool skip() const { return false; }
    __device__ void pair(int64_t i, double *) {
        double2 v = mk_ld2(in, i);
        v.x = -v.x;
        v.y = -v.y;
        mk_st2(out, i, v);
    }
    __device__ void one(int64_t i, double *) { out[i] = -in[i]; }
};

struct MkOpCopy {               // out = in
    static constexpr int NACC = 0, SLOT0 = 0;
    const double *in;
    double *out;
    __device__ bool prologue(double *, bool) { return false; }
    __device__ bool skip() const { return false; }
    __device__ void pair(int64_t i, double *) { mk_st2(out, i, mk_ld2(in, i)); }
    __device__ void one(int64_t i, double *) { out[i] = in[i]; }
};

template <int SIGN>             // y = y + SIGN * x   (exact add/sub, no multiply)
struct MkOpAddTo {
    static constexpr int NACC = 0, SLOT0 = 0;
    const double *x;
    double *y;
    __device__ bool prologue(double *, bool) { return false; }
    __device__ bool skip() const { return false; }
    __device__ void pair(int64_t i, double *) {
        double2 a = mk_ld2(x, i), b = mk_ld2(y, i);
        b.x = SIGN > 0 ? b.x + a.x : b.x - a.x;
        b.y = SIGN > 0 ? b.y + a.y : b.y - a.y;
        mk_st2(y, i, b);
    }
    __device__ void one(int64_t i, double *) { y[i] = SIGN > 0 ? y[i] + x[i] : y[i] - x[i]; }
};

struct MkOpSub {                // out = a - b
    static constexpr int NACC = 0, SLOT0 = 0;
    const double *a, *b;
    double *out;
    __device__ bool prologue(double *, bool) { return false; }
    __device__ bool skip() const { return false; }
    __device__ void pair(int64_t i, double *) {
        double2 u = mk_ld2(a, i), v = mk_ld2(b, i);
        u.x -= v.x;
        u.y -= v.y;
        mk_st2(out, i, u);
    }
    __device__ void one(int64_t i, double *) { out[i] = a[i] - b[i]; }
};

struct MkOpMul {                // out = a * b elementwise  (DiagonalOperator matvec `diag*x`, linop.py:503)
    static constexpr int NACC = 0, SLOT0 = 0;
    const double *a, *b;
    double *out;
    __device__ bool prologue(double *, bool) { return false; }
    __device__ bool skip() const { return false; }
    __device__ void pair(int64_t i, double *) {
        double2 u = mk_ld2(a, i), v = mk_ld2(b, i);
        u.x *= v.x;
        u.y *= v.y;
        mk_st2(out, i, u);
    }
    __device__ void one(int64_t i, double *) { out[i] = a[i] * b[i]; }
};

template <int SLOT>             // partial sums of dot(a, b) into slot SLOT
struct MkOpDot {
    static constexpr int NACC = 1, SLOT0 = SLOT;
    const double *a, *b;
    __device__ bool prologue(double *, bool) { return false; }
    __device__ bool skip() const { return false; }
    __device__ void pair(int64_t i, double *acc) {
        double2 u = mk_ld2(a, i), v = mk_ld2(b, i);
        acc[0] += u.x * v.x;
        acc[0] += u.y * v.y;
    }
    __device__ void one(int64_t i, double *acc) { acc[0] += a[i] * b[i]; }
};

// Product vectors of problems beyond the Infinity Cache are written once and read by the NEXT kernel from HBM: storing them
// non-temporally keeps them from pushing the x windows out of the L2 (mk_store_nt, mk_device.h: -5 ... -7 % on the CG product
// at 512^3).  CG carries the choice as a template parameter; the other loops' epilogues take it at run time through this
// helper -- an inline-asm store in the `nt` branch, because the compiler merges `if (nt) nontemporal_store else store` into
// ONE plain store (round 3's finding) -- so that no kernel is instantiated twice for it.  `nt` is launch uniform.
__device__ __forceinline__ void mk_store_stream(double *p, double v, int nt) {
    if (nt) asm volatile("global_store_dwordx2 %0, %1, off nt" : : "v"(p), "v"(v) : "memory");
    else *p = v;
}

struct MkPlainEpi {             // y = A x, nothing fused
    static constexpr int NACC = 0, SLOT0 = 0;
    static constexpr bool SYM_MARCH = true;                  // (may meet format 11: mk_device.h MkSymMarch)
    double *y;
    __device__ void prologue(double *) {}
    __device__ double xin(double v) const { return v; }
    __device__ void row(int64_t r, double s, double *) { y[r] = s; }
    // the brick march on a general geometry (mk_spmv_fmt9.h, GEN): rows r, r + 1 where they exist
    __device__ void row2_m(int64_t r, mk_d2 s, mk_d2, bool oka, bool okb, double *dump, double *) {
        *reinterpret_cast<mk_d2u *>(okb ? y + r : dump) = s;
        if (oka && !okb) y[r] = s.x;
    }
};

template <class Op>
static inline int mk_launch_stream(mk_solver *s, const Op &op, int64_t n) {
    const int grid = mk_grid_stream(n);
    hipLaunchKernelGGL(mk_stream_kernel<Op>, dim3(grid), dim3(MK_BLOCK), 0, s->stream, op, n, s->next_halt(),
                       s->d_part);
    return MK_OK;
}

template <class Epi, class Gate = MkNoGate>
static inline int mk_launch_spmv_on(mk_solver *s, const mk_csr *M, const double *x, const Epi &epi,
                                    const Gate &gate = Gate()) {
    mk_spmv_launch_blocks(M, mk_grid_spmv_for(M), s->stream, x, epi, gate, [&] { return s->next_halt(); }, s->d_part);
    return MK_OK;
}

template <class Epi, class Gate = MkNoGate>
static inline int mk_launch_spmv(mk_solver *s, const double *x, const Epi &epi, bool timed = true,
                                 const Gate &gate = Gate()) {
    const mk_csr *A = s->A;
    if (timed) s->spmv_begin();
    const MkPlan *plan = A->ex.pending ? mk_csr_plan(A) : nullptr;
    const bool march = plan && mk_fmt_march(plan->fmt);
    int za, zb;
    const bool thin = MkSymMarch<Epi>::value && mk_pen_tail_gen();   // (mk_pen_split: only the first and the last plane wait)
    if (A->ex.pending && march && (!mk_march_kernel_for<Epi>(plan) || !mk_pen_split(plan, &za, &zb, thin))) {
        // a slab of too few planes to split -- or a loop without a kernel for this march format, whose product runs as the
        // CSR gather kernel over ALL rows (a plane range means nothing to it): the messages first, then one launch
        int rc = mk_exchange_wait(A, s->stream);
        if (rc != MK_OK) return rc;
        mk_spmv_launch_blocks(A, mk_grid_spmv_for(A), s->stream, x, epi, gate, [&] { return s->next_halt(); }, s->d_part);
    } else if (A->ex.pending) {
        // halo exchange in flight: rows that need no received entry first, the others once the messages are in
        // (the two launches write disjoint ranges of the partial-sum slots)
        int g1, g2;
        const MkCsrView v1 = mk_view_part(A, 1, 0, thin);
        if (march) {                                         // plane ranges of the brick march (mk_pen_split)
            g1 = mk_pen_items(v1);
            g2 = mk_pen_items(mk_view_part(A, 2, 0, thin));
            // (the kernels stride their items by the grid: a plane of >= 1024 bricks must not leave the interior launch a
            //  grid of 0 -- at most half of the partial-sum slots for the boundary planes)
            if (g2 > MK_MAXP / 2) g2 = MK_MAXP / 2;
            if (g1 + g2 > MK_MAXP) g1 = MK_MAXP - g2;
        } else {
            mk_grid_spmv_parts(A, A->ex.n_int, A->ex.n_bnd, &g1, &g2);
        }
        mk_spmv_launch_view(v1, g1, s->stream, x, epi, gate, s->next_halt(), s->d_part);
        int rc = mk_exchange_wait(A, s->stream);
        if (rc != MK_OK) return rc;
        mk_spmv_launch_view(mk_view_part(A, 2, g1, thin), g2, s->stream, x, epi, gate, s->next_halt(), s->d_part);
    } else {
        mk_spmv_launch_blocks(A, mk_grid_spmv_for(A), s->stream, x, epi, gate, [&] { return s->next_halt(); }, s->d_part);
    }
    if (timed) s->spmv_end();
    return MK_OK;
}
#endif
