// mk_solver.hip -- generic driver of the device-resident solver loops + C ABI dispatch.
#include "mk_solver.h"

int mk_solver::init_common(const mk_csr *A_, const mk_params *p) {
    A = A_;
    A->dependents += 1;              // (borrowed for the solver's lifetime: destroying A meanwhile is deferred)
    prm = *p;
    n = A->ex.mode >= 0 ? A->ex.n_local : A->nrows;
    const bool rectangular_ok = prm.kind >= MK_LSQR;
    if (rectangular_ok && A->ex.mode >= 0)
        return mk_fail(MK_ERR_UNSUPPORTED, "the least-squares solvers partition by row blocks with a replicated column "
                       "space (mk_csr_set_row_block), not with a halo / all-gather exchange plan");
    if (!rectangular_ok && A->ex.mode < 0 && A->nrows != A->ncols)
        return mk_fail(MK_ERR_ARG, "solver needs a square operator, got %lld x %lld", (long long)A->nrows,
                       (long long)A->ncols);
    nx = A->x_len();
    stream = mk_ctx().stream;
    MK_HIP(hipMalloc((void **)&d_scal, sizeof(double) * MK_NSCAL));
    MK_HIP(hipMalloc((void **)&d_part, sizeof(double) * MK_NDOT * MK_MAXP));
    MK_HIP(hipMalloc((void **)&d_halt, 2 * sizeof(int)));
    MK_HIP(hipMalloc((void **)&d_status, sizeof(MkStatus)));
    MK_HIP(hipMalloc((void **)&d_hist, sizeof(double) * 2 * MK_HIST_RING));
    MK_HIP(hipHostMalloc((void **)&h_status, sizeof(MkStatus), hipHostMallocDefault));
    MK_HIP(hipHostMalloc((void **)&h_scal, sizeof(double) * MK_NSCAL, hipHostMallocDefault));
    MK_HIP(hipEventCreate(&ev0));
    MK_HIP(hipEventCreate(&ev1));
    // with several ranks every consumer adds all MK_MAXP slots (unused ones stay zero) so that the
    // all-reduced partial vectors mean the same thing on every rank
    const bool multi = mk_comm_active() != 0;
    np_spmv = multi ? MK_MAXP : mk_grid_spmv_for(A);
    np_stream = multi ? MK_MAXP : mk_grid_stream(n);
    spmv_sample_stride = prm.spmv_event_stride;
    return MK_OK;
}

// (stands in for the callback when the preconditioner is a device operator: every `if (precon_fn ...)` site stays as it is)
static int mk_precon_on_device(void *, const double *, double *) { return 1; }

int mk_solver::host_precon(const double *in_dev, double *out_dev, bool force) {
    if (!precon_fn) return MK_OK;
    if (precon_op) {
        // out = precon_op * in on the device.  Like every kernel of the loop the product obeys the halt words: once the
        // loop condition has failed it is a no-op, exactly when the reference applies nothing more -- unless `force`.
        double *dst = (in_dev == out_dev) ? d_ptmp : out_dev;
        const int grid = mk_grid_spmv_for(precon_op);
        if (force) {
            mk_spmv_launch_blocks(precon_op, grid, stream, in_dev, MkPlainEpi{dst}, MkNoGate(),
                                  [&] { return MkHalt{d_nohalt, 0, 0}; }, d_part);
            if (dst != out_dev) MK_HIP(hipMemcpyAsync(out_dev, dst, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream));
        } else {
            mk_spmv_launch_blocks(precon_op, grid, stream, in_dev, MkPlainEpi{dst}, MkNoGate(),
                                  [&] { return next_halt(); }, d_part);
            if (dst != out_dev) mk_launch_stream(this, MkOpCopy{dst, out_dev}, n);
        }
        return mk_ctx().pending_rc;
    }
    int h = 0;
    MK_HIP(hipMemcpyAsync(&h, d_halt + (q & 1), sizeof(int), hipMemcpyDeviceToHost, stream));   // the next kernel's word
    if (n > 0) MK_HIP(hipMemcpyAsync(h_pin, in_dev, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, stream));
    MK_HIP(hipStreamSynchronize(stream));
    if (h && !force) return MK_OK;                           // the loop has ended: the reference applies nothing more
    if (precon_fn(precon_user, h_pin, h_pout) != 0) {
        const int rc = mk_fail(MK_ERR_STATE, "the host preconditioner callback reported a failure");
        if (mk_ctx().pending_rc == MK_OK) mk_ctx().pending_rc = rc;
        return rc;
    }
    if (n > 0) MK_HIP(hipMemcpyAsync(out_dev, h_pout, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, stream));
    return MK_OK;
}

mk_solver::~mk_solver() {
    if (mk_ctx().ready) hipStreamSynchronize(mk_ctx().stream);
    if (precon_op) mk_release_operand(precon_op);
    if (At) mk_release_operand(At);
    if (A && counted_user) mk_csr_count_users(A, -1);
    if (A) mk_release_operand(A);
    hipFree(d_ones);
    hipFree(d_ptmp);
    hipFree(d_nohalt);
    if (h_pin) hipHostFree(h_pin);
    if (h_pout) hipHostFree(h_pout);
    if (mk_ctx().ready) hipStreamSynchronize(mk_ctx().stream);
    for (double *v : vecs) hipFree(v);
    if (!arena_vecs.empty()) {                               // the arena is reusable once its last vector is back
        MkContext &c = mk_ctx();
        c.arena_live -= (int)arena_vecs.size();
        if (c.arena_live <= 0) {
            c.arena_live = 0;
            c.arena_off = 0;
        }
    }
    for (hipEvent_t e : spmv_ev) hipEventDestroy(e);
    hipFree(d_scal);
    hipFree(d_part);
    hipFree(d_halt);
    hipFree(d_status);
    hipFree(d_hist);
    if (h_status) hipHostFree(h_status);
    if (h_scal) hipHostFree(h_scal);
    if (ev0) hipEventDestroy(ev0);
    if (ev1) hipEventDestroy(ev1);
}

int mk_solver::alloc_vec(double **out, int64_t len) {
    double *p = nullptr;
    const size_t bytes = sizeof(double) * (size_t)(len > 0 ? len : 1) + 16;
    MkContext &c = mk_ctx();
    const size_t step = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);   // 2 MiB granules
    // placement experiment (round 6, tools/r06_placement_offsets.py): MK_ARENA_SKEW="s0,s1,..." shifts the k-th vector carved
    // from the arena by s_k bytes (multiples of 256) -- the relative phase of the loop's write streams, chosen instead of drawn
    size_t skew = 0;
    if (c.arena) {
        static const char *env = getenv("MK_ARENA_SKEW");
        if (env) {
            int k = 0;
            for (const char *q = env; *q; ++k) {
                const long v = strtol(q, const_cast<char **>(&q), 10);
                if (k == c.arena_live) skew = (size_t)(v > 0 ? v : 0) & ~(size_t)255;
                if (*q == ',') ++q;
            }
        }
    }
    if (c.arena && c.arena_off + skew + step <= c.arena_size) {     // carved from the arena reserved before the matrix
        c.arena_off += skew;
        p = reinterpret_cast<double *>(c.arena + c.arena_off);
        c.arena_off += step;
        c.arena_live += 1;
        arena_vecs.push_back(p);
    } else {
        MK_HIP(hipMalloc((void **)&p, bytes));
        vecs.push_back(p);
    }
    MK_HIP(hipMemsetAsync(p, 0, bytes, stream));
    *out = p;
    return MK_OK;
}

extern "C" int mk_arena_reserve(size_t bytes) {
    MK_REQUIRE_INIT();
    MkContext &c = mk_ctx();
    if (c.arena_live > 0) return mk_fail(MK_ERR_STATE, "mk_arena_reserve: %d vectors of the current arena are still in use", c.arena_live);
    if (c.arena) {
        MK_HIP(hipStreamSynchronize(c.stream));
        MK_HIP(hipFree(c.arena));
        c.arena = nullptr;
        c.arena_size = c.arena_off = 0;
    }
    if (bytes == 0) return MK_OK;
    MK_HIP(hipMalloc((void **)&c.arena, bytes));
    c.arena_size = bytes;
    c.arena_off = 0;
    return MK_OK;
}

int mk_solver::allreduce(int slot0, int nslots) {
    if (!mk_comm_active()) return MK_OK;
    return mk_comm_allreduce_sum(d_part + (size_t)slot0 * MK_MAXP, (int64_t)nslots * MK_MAXP, stream);
}

int mk_solver::exchange(double *x_ext) {
    if (A->ex.mode < 0) return MK_OK;
    return mk_exchange_begin(A, x_ext);                    // mk_launch_spmv completes it
}

void mk_solver::spmv_begin() {
    if (spmv_sample_stride <= 0 || (it % spmv_sample_stride) != 0) return;
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
    spmv_ev.push_back(a);
    spmv_ev.push_back(b);
    hipEventRecord(a, stream);
}

void mk_solver::spmv_end() {
    if (spmv_sample_stride <= 0 || (it % spmv_sample_stride) != 0 || spmv_ev.empty()) return;
    hipEventRecord(spmv_ev.back(), stream);
}

int mk_solver::collect_spmv_timing() {
    for (size_t i = 0; i + 1 < spmv_ev.size(); i += 2) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, spmv_ev[i], spmv_ev[i + 1]) == hipSuccess) {
            spmv_ms += ms;
            spmv_timed += 1;
        }
        hipEventDestroy(spmv_ev[i]);
        hipEventDestroy(spmv_ev[i + 1]);
    }
    spmv_ev.clear();
    return MK_OK;
}

int mk_solver::poll() {
    MK_HIP(hipMemcpyAsync(h_status, d_status, sizeof(MkStatus), hipMemcpyDeviceToHost, stream));
    MK_HIP(hipMemcpyAsync(h_scal, d_scal, sizeof(double) * MK_NSCAL, hipMemcpyDeviceToHost, stream));
    int h[2];
    MK_HIP(hipMemcpyAsync(h, d_halt, sizeof(h), hipMemcpyDeviceToHost, stream));
    MK_HIP(hipStreamSynchronize(stream));
    halted = h[q & 1] != 0;      // the word the NEXT kernel would read
    // drain new residual-history entries from the ring
    const int64_t have = h_status->hist_len;
    if (have > hist_drained) {
        if (have - hist_drained > MK_HIST_RING)
            return mk_fail(MK_ERR_STATE, "history ring overrun (%lld new entries)", (long long)(have - hist_drained));
        hist.resize((size_t)have);
        if (use_hist2) hist2.resize((size_t)have);
        int64_t pos = hist_drained;
        while (pos < have) {
            const int64_t off = pos % MK_HIST_RING;
            int64_t cnt = have - pos;
            if (cnt > MK_HIST_RING - off) cnt = MK_HIST_RING - off;
            MK_HIP(hipMemcpyAsync(hist.data() + pos, d_hist + off, sizeof(double) * (size_t)cnt,
                                  hipMemcpyDeviceToHost, stream));
            if (use_hist2)
                MK_HIP(hipMemcpyAsync(hist2.data() + pos, d_hist + MK_HIST_RING + off, sizeof(double) * (size_t)cnt,
                                      hipMemcpyDeviceToHost, stream));
            pos += cnt;
        }
        MK_HIP(hipStreamSynchronize(stream));
        hist_drained = have;
    }
    return MK_OK;
}

int mk_solver::iterate(int64_t max_iters, int64_t *done) {
    if (!is_setup) return mk_fail(MK_ERR_STATE, "mk_solver_iterate before mk_solver_setup");
    int64_t launched = 0;
    int64_t batch = 16;
    const int64_t itn0 = h_status->itn;
    MK_HIP(hipEventRecord(ev0, stream));
    while (!halted && launched < max_iters) {
        int64_t todo = max_iters - launched;
        if (todo > batch) todo = batch;
        for (int64_t k = 0; k < todo; ++k) {
            int rc = enqueue_pass();
            // a host operator / preconditioner callback failed inside this pass: stop enqueueing at once -- the
            // caller's operator must not be called again on vectors that no longer mean anything
            if (rc == MK_OK && mk_ctx().pending_rc != MK_OK) rc = mk_ctx().pending_rc;
            if (rc != MK_OK) {
                mk_ctx().pending_rc = MK_OK;                 // (reported here: must not resurface in a later call)
                return rc;
            }
            ++it;
        }
        MK_HIP(hipGetLastError());
        if (mk_ctx().pending_rc != MK_OK) {                  // (a host operator callback failed)
            const int prc = mk_ctx().pending_rc;
            mk_ctx().pending_rc = MK_OK;
            return prc;
        }
        launched += todo;
        int rc = poll();
        if (rc != MK_OK) return rc;
        if (batch < MK_BATCH_MAX) batch *= 2;
    }
    MK_HIP(hipEventRecord(ev1, stream));
    MK_HIP(hipEventSynchronize(ev1));
    float ms = 0.f;
    MK_HIP(hipEventElapsedTime(&ms, ev0, ev1));
    last_iterate_ms = ms;
    collect_spmv_timing();
    if (done) *done = h_status->itn - itn0;
    return MK_OK;
}

void mk_solver::fill_result(mk_result *res) const {
    memset(res, 0, sizeof(*res));
    res->struct_size = (int32_t)sizeof(mk_result);
    res->halted = halted ? 1 : 0;
    res->nMatvec = h_status->nMatvec;
    res->itn = h_status->itn;
    res->hist_len = (int64_t)hist.size();
    res->converged = h_status->converged;
    res->definite = h_status->definite;
    res->istop = h_status->istop;
}

// ======================================================================================
// C ABI
// ======================================================================================
extern "C" int mk_solver_create(const mk_csr *A, const mk_params *params, mk_solver **out) {
    MK_REQUIRE_INIT();
    MK_ARG(A && params && out);
    MK_ARG(params->struct_size == (int32_t)sizeof(mk_params));
    mk_solver *s = nullptr;
    switch (params->kind) {
        case MK_CG: s = mk_make_cg(); break;
        case MK_BICGSTAB: s = mk_make_bicgstab(); break;
        case MK_CGS: s = mk_make_cgs(); break;
        case MK_TFQMR: s = mk_make_tfqmr(); break;
        case MK_MINRES: s = mk_make_minres(); break;
        case MK_SYMMLQ: s = mk_make_symmlq(); break;
        case MK_LSQR:
        case MK_LSMR:
        case MK_CRAIG:
        case MK_CRAIGMR: s = mk_make_lls(params->kind); break;
        default: break;
    }
    if (!s) return mk_fail(MK_ERR_UNSUPPORTED, "mk_solver_create: solver kind %d is not available", params->kind);
    mk_csr_march_pref(A, params->kind == MK_CG ? 1 : 0);     // (before the partial-sum counts are sized for the format in use)
    int rc = s->init_common(A, params);
    if (rc != MK_OK) {
        delete s;
        return rc;
    }
    mk_csr_count_users(A, 1);
    s->counted_user = true;
    *out = s;
    return MK_OK;
}

extern "C" int mk_solver_set_transpose(mk_solver *s, const mk_csr *At) {
    MK_ARG(s && At);
    MK_ARG(At->nrows == s->A->ncols && At->ncols == s->A->nrows && At->nnz == s->A->nnz);
    if (s->At == At) return MK_OK;                           // (re-setting the same operator must not release it)
    At->dependents += 1;                                     // take the new reference first, then drop the old one
    if (s->At) mk_release_operand(s->At);
    s->At = At;
    return MK_OK;
}

extern "C" int mk_csr_set_row_block(mk_csr *A, int on) {
    MK_ARG(A != nullptr);
    if (on && A->ex.mode >= 0)
        return mk_fail(MK_ERR_STATE, "mk_csr_set_row_block: the matrix already carries a halo / all-gather exchange plan");
    if (on < 0 || on > 2) return mk_fail(MK_ERR_ARG, "mk_csr_set_row_block: mode must be 0, 1 or 2");
    A->row_block = on;
    return MK_OK;
}

extern "C" int mk_solver_set_precon_diag(mk_solver *s, const double *diag) {
    MK_ARG(s);
    MK_ARG(MK_ALIGNED16(diag));
    if (diag && !s->takes_precon())
        return mk_fail(MK_ERR_UNSUPPORTED, "this solver kind has no device preconditioner hook");
    s->d_prec = diag;
    return MK_OK;
}

__global__ __launch_bounds__(MK_BLOCK) void mk_fill_kernel(double *v, int64_t n, double a) {
    for (int64_t i = (int64_t)blockIdx.x * MK_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * MK_BLOCK) v[i] = a;
}

extern "C" int mk_solver_set_precon_callback(mk_solver *s, mk_precon_fn fn, void *user) {
    MK_ARG(s);
    if (s->precon_op) mk_release_operand(s->precon_op);
    s->precon_op = nullptr;
    if (!fn) {
        s->precon_fn = nullptr;
        s->d_prec = nullptr;
        return MK_OK;
    }
    if (!s->takes_precon()) return mk_fail(MK_ERR_UNSUPPORTED, "this solver kind has no preconditioner hook");
    if (s->A->ex.mode >= 0)
        return mk_fail(MK_ERR_UNSUPPORTED, "host preconditioner callbacks are single-GPU (the vector would have to be gathered)");
    const size_t len = (size_t)(s->n > 0 ? s->n : 1);
    if (!s->d_ones) {
        MK_HIP(hipMalloc((void **)&s->d_ones, sizeof(double) * len + 16));
        hipLaunchKernelGGL(mk_fill_kernel, dim3(512), dim3(MK_BLOCK), 0, s->stream, s->d_ones, (int64_t)len, 1.0);
        MK_HIP(hipGetLastError());
    }
    if (!s->h_pin) {
        MK_HIP(hipHostMalloc((void **)&s->h_pin, sizeof(double) * len, hipHostMallocDefault));
        MK_HIP(hipHostMalloc((void **)&s->h_pout, sizeof(double) * len, hipHostMallocDefault));
    }
    s->precon_fn = fn;
    s->precon_user = user;
    s->d_prec = s->d_ones;
    return MK_OK;
}

extern "C" int mk_solver_set_precon_csr(mk_solver *s, const mk_csr *M) {
    MK_ARG(s);
    if (!M) {
        if (s->precon_op) mk_release_operand(s->precon_op);
        s->precon_op = nullptr;
        s->precon_fn = nullptr;
        s->d_prec = nullptr;
        return MK_OK;
    }
    if (!s->takes_precon()) return mk_fail(MK_ERR_UNSUPPORTED, "this solver kind has no preconditioner hook");
    if (M->nrows != s->n || M->ncols != s->n || M->ex.mode >= 0)
        return mk_fail(MK_ERR_ARG, "mk_solver_set_precon_csr: the preconditioner must be a square device operator of the "
                       "solver's (local) size %lld without an exchange plan, got %lld x %lld", (long long)s->n,
                       (long long)M->nrows, (long long)M->ncols);
    const size_t len = (size_t)(s->n > 0 ? s->n : 1);
    if (!s->d_ones) {
        MK_HIP(hipMalloc((void **)&s->d_ones, sizeof(double) * len + 16));
        hipLaunchKernelGGL(mk_fill_kernel, dim3(512), dim3(MK_BLOCK), 0, s->stream, s->d_ones, (int64_t)len, 1.0);
        MK_HIP(hipGetLastError());
    }
    if (!s->d_ptmp) {
        MK_HIP(hipMalloc((void **)&s->d_ptmp, sizeof(double) * len + 16));
        MK_HIP(hipMalloc((void **)&s->d_nohalt, 2 * sizeof(int)));
        MK_HIP(hipMemsetAsync(s->d_nohalt, 0, 2 * sizeof(int), s->stream));
    }
    M->dependents += 1;
    if (s->precon_op) mk_release_operand(s->precon_op);
    s->precon_op = M;
    s->precon_fn = mk_precon_on_device;                      // (marks "general preconditioner" at the call sites)
    s->precon_user = nullptr;
    s->d_prec = s->d_ones;
    return MK_OK;
}

extern "C" int mk_solver_set_lls_precon(mk_solver *s, const double *diag_m, const double *diag_n) {
    MK_ARG(s);
    MK_ARG(MK_ALIGNED16(diag_m) && MK_ALIGNED16(diag_n));
    if (s->prm.kind < MK_LSQR || s->prm.kind > MK_CRAIGMR)
        return mk_fail(MK_ERR_UNSUPPORTED, "mk_solver_set_lls_precon: not a least-squares solver");
    return mk_lls_set_metric(s, diag_m, diag_n);
}

extern "C" int mk_solver_set_lls_precon_callback(mk_solver *s, mk_precon_fn fn_m, void *user_m, mk_precon_fn fn_n,
                                                 void *user_n) {
    MK_ARG(s);
    if (s->prm.kind < MK_LSQR || s->prm.kind > MK_CRAIGMR)
        return mk_fail(MK_ERR_UNSUPPORTED, "mk_solver_set_lls_precon_callback: not a least-squares solver");
    return mk_lls_set_callbacks(s, fn_m, user_m, fn_n, user_n);
}

extern "C" int mk_solver_destroy(mk_solver *s) {
    delete s;
    return MK_OK;
}

extern "C" int mk_solver_setup(mk_solver *s, const double *rhs, const double *guess) {
    MK_ARG(s && rhs);
    MK_ARG(MK_ALIGNED16(rhs) && MK_ALIGNED16(guess));        // borrowed vectors are read in 16-byte pairs
    s->q = 0;
    s->it = 0;
    s->halted = false;
    s->hist.clear();
    s->hist2.clear();
    s->hist_drained = 0;
    s->spmv_ms = 0.0;
    s->spmv_timed = 0;
    MK_HIP(hipMemsetAsync(s->d_halt, 0, 2 * sizeof(int), s->stream));
    MK_HIP(hipMemsetAsync(s->d_status, 0, sizeof(MkStatus), s->stream));
    MK_HIP(hipMemsetAsync(s->d_scal, 0, sizeof(double) * MK_NSCAL, s->stream));
    MK_HIP(hipMemsetAsync(s->d_part, 0, sizeof(double) * MK_NDOT * MK_MAXP, s->stream));
    mk_ctx().pending_rc = MK_OK;                             // (a failure of an earlier solve was reported there)
    int rc = s->setup(rhs, guess);
    if (rc != MK_OK) {
        mk_ctx().pending_rc = MK_OK;
        return rc;
    }
    MK_HIP(hipGetLastError());
    if (mk_ctx().pending_rc != MK_OK) {
        rc = mk_ctx().pending_rc;
        mk_ctx().pending_rc = MK_OK;
        return rc;
    }
    s->is_setup = true;
    return s->poll();
}

extern "C" int mk_solver_iterate(mk_solver *s, int64_t max_iters, int64_t *iters_done) {
    MK_ARG(s && max_iters >= 0);
    return s->iterate(max_iters, iters_done);
}

extern "C" int mk_solver_finish(mk_solver *s, mk_result *res) {
    MK_ARG(s && res);
    if (!s->is_setup) return mk_fail(MK_ERR_STATE, "mk_solver_finish before mk_solver_setup");
    return s->finish(res);
}

extern "C" int mk_solver_x(const mk_solver *s, const double **x_dev) {
    MK_ARG(s && x_dev);
    *x_dev = s->x();
    return MK_OK;
}

extern "C" int mk_solver_fused(const mk_solver *s, int32_t *fused) {
    MK_ARG(s && fused);
    *fused = s->is_fused() ? 1 : 0;
    return MK_OK;
}

extern "C" int mk_solver_history(const mk_solver *s, double *hist_host, int64_t cap) {
    MK_ARG(s && (cap == 0 || hist_host));
    int64_t cnt = (int64_t)s->hist.size();
    if (cnt > cap) cnt = cap;
    memcpy(hist_host, s->hist.data(), sizeof(double) * (size_t)cnt);
    return MK_OK;
}

extern "C" int mk_solver_history2(const mk_solver *s, double *hist_host, int64_t cap) {
    MK_ARG(s && (cap == 0 || hist_host));
    int64_t cnt = (int64_t)s->hist2.size();
    if (cnt > cap) cnt = cap;
    memcpy(hist_host, s->hist2.data(), sizeof(double) * (size_t)cnt);
    return MK_OK;
}

extern "C" int mk_solver_vector(const mk_solver *s, int index, const double **v_dev, int64_t *len) {
    MK_ARG(s && v_dev);
    const double *v = s->vector(index);
    if (!v) return mk_fail(MK_ERR_ARG, "mk_solver_vector: index %d is not defined for this solver", index);
    *v_dev = v;
    if (len) *len = s->n;
    return MK_OK;
}

extern "C" int mk_solver_time_spmv(mk_solver *s, int64_t launches, double *avg_us) {
    return mk_solver_time_product(s, 0, launches, avg_us);
}

extern "C" int mk_solver_time_product(mk_solver *s, int which, int64_t launches, double *avg_us) {
    MK_ARG(s && launches > 0 && avg_us && which >= 0);
    if (!s->is_setup) return mk_fail(MK_ERR_STATE, "mk_solver_time_product before mk_solver_setup");
    int rc = s->enqueue_spmv_only(which);                  // one untimed launch first
    if (rc != MK_OK) return rc;
    MK_HIP(hipEventRecord(s->ev0, s->stream));
    for (int64_t k = 0; k < launches; ++k)
        if ((rc = s->enqueue_spmv_only(which)) != MK_OK) return rc;
    MK_HIP(hipEventRecord(s->ev1, s->stream));
    MK_HIP(hipEventSynchronize(s->ev1));
    MK_HIP(hipGetLastError());
    float ms = 0.f;
    MK_HIP(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    *avg_us = 1e3 * (double)ms / (double)launches;
    return MK_OK;
}

extern "C" int mk_solver_timing(const mk_solver *s, double *iterate_ms, double *spmv_ms, int64_t *spmv_launches) {
    MK_ARG(s != nullptr);
    if (iterate_ms) *iterate_ms = s->last_iterate_ms;
    if (spmv_ms) *spmv_ms = s->spmv_ms;
    if (spmv_launches) *spmv_launches = s->spmv_timed;
    return MK_OK;
}

extern "C" int mk_solver_solve(mk_solver *s, const double *rhs, const double *guess, mk_result *res) {
    int rc = mk_solver_setup(s, rhs, guess);
    if (rc != MK_OK) return rc;
    while (!s->halted) {
        rc = s->iterate((int64_t)1 << 20, nullptr);
        if (rc != MK_OK) return rc;
    }
    return mk_solver_finish(s, res);
}
