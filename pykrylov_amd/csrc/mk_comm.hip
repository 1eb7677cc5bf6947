// mk_comm.hip -- multi-GPU plumbing: one process per GPU, RCCL over xGMI.
//
// The reference is single-process; this file is new design.  Two collectives
// exist on the solver path: (C1) the exchange of the SpMV input vector -- neighbour
// send/recv of exactly the off-rank entries a rank's rows reference ("halo", the path that
// fits the xGMI budget) or a full all-gather (the general path north_star names) -- and
// (C2) a sum all-reduce of the per-workgroup partial sums of every dot product.
// RCCL is bound at run time with dlopen so that single-GPU use needs no RCCL at all and so
// that the process shares whichever librccl.so.1 PyTorch may already have loaded.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "mk_solver.h"

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommSplit)(ncclComm_t, int, int, ncclComm_t *, ncclConfig_t *) = nullptr;   // optional
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;                               // optional
};

Rccl g_rccl;
ncclComm_t g_comm = nullptr;
// The halo messages of an overlapped product travel on a second stream while all-reduces are enqueued on the main
// one: they get their own communicator (ncclCommSplit of the first), so that no two operations of ONE communicator
// are ever in flight on different streams.  Falls back to the shared communicator if the library lacks the call.
ncclComm_t g_comm_halo = nullptr;
int g_nranks = 1, g_rank = 0;

// Host-staged transport (mk_comm_init_host): the same collectives carried by caller-supplied functions that work
// on HOST buffers (e.g. torch.distributed/gloo).  Every call stages through pinned memory and synchronises the
// stream -- for tests of the multi-rank logic on machines without a second GPU, not for speed.
struct HostComm {
    bool active = false;
    mk_host_allreduce_fn allreduce = nullptr;
    mk_host_exchange_fn exchange = nullptr;
    mk_host_allgather_fn allgather = nullptr;
    double *send = nullptr, *recv = nullptr;      // pinned staging
    size_t send_cap = 0, recv_cap = 0;
    int reserve(double **buf, size_t *cap, size_t count) {
        if (count <= *cap) return MK_OK;
        if (*buf) hipHostFree(*buf);
        *buf = nullptr;
        *cap = 0;
        MK_HIP(hipHostMalloc((void **)buf, sizeof(double) * count));
        *cap = count;
        return MK_OK;
    }
};
HostComm g_host;

int load_rccl() {
    if (g_rccl.handle) return MK_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *nm : names) {
        h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return mk_fail(MK_ERR_COMM, "cannot dlopen librccl.so.1: %s", dlerror());
#define MK_SYM(field, name)                                                        \
    g_rccl.field = (decltype(g_rccl.field))dlsym(h, name);                         \
    if (!g_rccl.field) return mk_fail(MK_ERR_COMM, "librccl lacks symbol %s", name)
    MK_SYM(GetUniqueId, "ncclGetUniqueId");
    MK_SYM(CommInitRank, "ncclCommInitRank");
    MK_SYM(CommDestroy, "ncclCommDestroy");
    MK_SYM(AllReduce, "ncclAllReduce");
    MK_SYM(AllGather, "ncclAllGather");
    MK_SYM(ReduceScatter, "ncclReduceScatter");
    MK_SYM(Send, "ncclSend");
    MK_SYM(Recv, "ncclRecv");
    MK_SYM(GroupStart, "ncclGroupStart");
    MK_SYM(GroupEnd, "ncclGroupEnd");
    MK_SYM(GetErrorString, "ncclGetErrorString");
#undef MK_SYM
    g_rccl.CommSplit = (decltype(g_rccl.CommSplit))dlsym(h, "ncclCommSplit");
    g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(h, "ncclCommCount");
    g_rccl.handle = h;
    return MK_OK;
}

#define MK_NCCL(expr)                                                                                  \
    do {                                                                                               \
        ncclResult_t _r = (expr);                                                                      \
        if (_r != ncclSuccess)                                                                         \
            return mk_fail(MK_ERR_COMM, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r),     \
                           __FILE__, __LINE__);                                                        \
    } while (0)

__global__ __launch_bounds__(MK_BLOCK) void pack_kernel(int64_t cnt, const int32_t *__restrict__ idx,
                                                        const double *__restrict__ x, double *__restrict__ buf) {
    for (int64_t k = (int64_t)blockIdx.x * MK_BLOCK + threadIdx.x; k < cnt; k += (int64_t)gridDim.x * MK_BLOCK)
        buf[k] = x[idx[k]];
}

// a tile (256 rows) is "boundary" if one of its rows references a received entry (column >= n_local)
__global__ __launch_bounds__(MK_BLOCK) void classify_tiles_kernel(int64_t nrows, int64_t n_local,
                                                                  const int32_t *__restrict__ indptr,
                                                                  const int32_t *__restrict__ indices,
                                                                  int *__restrict__ flags) {
    for (int64_t r = (int64_t)blockIdx.x * MK_BLOCK + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * MK_BLOCK) {
        bool halo = false;
        for (int32_t j = indptr[r]; j < indptr[r + 1]; ++j) halo |= (indices[j] >= n_local);
        if (halo) atomicOr(&flags[r / MK_ROWS_PER_TILE], 1);
    }
}

// interior / boundary tile lists and the second stream (halo mode with a non-empty halo)
int build_overlap_plan(mk_csr *A) {
    MkExchange &ex = A->ex;
    if (getenv("MK_NO_OVERLAP") || A->ntiles < 2 || ex.n_halo == 0) return MK_OK;
    hipStream_t st = mk_ctx().stream;
    int *d_flags = nullptr;
    MK_HIP(hipMalloc((void **)&d_flags, sizeof(int) * (size_t)A->ntiles));
    MK_HIP(hipMemsetAsync(d_flags, 0, sizeof(int) * (size_t)A->ntiles, st));
    int grid = (int)((A->nrows + MK_BLOCK - 1) / MK_BLOCK);
    grid = grid < 1 ? 1 : (grid > 65536 ? 65536 : grid);
    hipLaunchKernelGGL(classify_tiles_kernel, dim3(grid), dim3(MK_BLOCK), 0, st, A->nrows, ex.n_local, A->d_indptr,
                       A->d_indices, d_flags);
    std::vector<int> flags((size_t)A->ntiles);
    MK_HIP(hipMemcpyAsync(flags.data(), d_flags, sizeof(int) * (size_t)A->ntiles, hipMemcpyDeviceToHost, st));
    MK_HIP(hipStreamSynchronize(st));
    MK_HIP(hipFree(d_flags));
    std::vector<int32_t> list;
    list.reserve((size_t)A->ntiles);
    for (int64_t t = 0; t < A->ntiles; ++t)
        if (!flags[t]) list.push_back((int32_t)t);
    const int64_t n_int = (int64_t)list.size();
    for (int64_t t = 0; t < A->ntiles; ++t)
        if (flags[t]) list.push_back((int32_t)t);
    if (n_int == 0 || n_int == A->ntiles) return MK_OK;      // nothing to overlap
    MK_HIP(hipMalloc((void **)&ex.d_tiles, sizeof(int32_t) * list.size()));
    MK_HIP(hipMemcpy(ex.d_tiles, list.data(), sizeof(int32_t) * list.size(), hipMemcpyHostToDevice));
    ex.n_int = n_int;
    ex.n_bnd = A->ntiles - n_int;
    MK_HIP(hipStreamCreateWithFlags(&ex.comm_stream, hipStreamNonBlocking));
    MK_HIP(hipEventCreateWithFlags(&ex.ev_pack, hipEventDisableTiming));
    MK_HIP(hipEventCreate(&ex.ev_comm0));                    // (timed pair: duration of the last message group,
    MK_HIP(hipEventCreate(&ex.ev_comm));                     //  mk_csr_comm_last_us)
    return MK_OK;
}

}  // namespace

int mk_comm_active() { return (g_comm != nullptr || g_host.active) && g_nranks > 1; }

int mk_comm_allreduce_sum(double *buf, int64_t count, hipStream_t stream) {
    if (g_host.active) {
        int rc = g_host.reserve(&g_host.send, &g_host.send_cap, (size_t)count);
        if (rc != MK_OK) return rc;
        MK_HIP(hipMemcpyAsync(g_host.send, buf, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, stream));
        MK_HIP(hipStreamSynchronize(stream));
        if (g_host.allreduce(g_host.send, count) != 0) return mk_fail(MK_ERR_COMM, "host all-reduce callback failed");
        MK_HIP(hipMemcpyAsync(buf, g_host.send, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, stream));
        MK_HIP(hipStreamSynchronize(stream));
        return MK_OK;
    }
    if (!g_comm) return mk_fail(MK_ERR_COMM, "all-reduce without a communicator");
    MK_NCCL(g_rccl.AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, g_comm, stream));
    return MK_OK;
}

// mine = this rank's block of `count` entries of the element-wise sum over the ranks of `full` (count * nranks entries)
int mk_comm_reduce_scatter_sum(const double *full, double *mine, int64_t count, hipStream_t stream) {
    if (g_host.active) {                                     // host transport: all-reduce on the host, keep one block
        const size_t tot = (size_t)count * (size_t)g_nranks;
        int rc = g_host.reserve(&g_host.send, &g_host.send_cap, tot ? tot : 1);
        if (rc != MK_OK) return rc;
        MK_HIP(hipMemcpyAsync(g_host.send, full, sizeof(double) * tot, hipMemcpyDeviceToHost, stream));
        MK_HIP(hipStreamSynchronize(stream));
        if (g_host.allreduce(g_host.send, (int64_t)tot) != 0) return mk_fail(MK_ERR_COMM, "host all-reduce callback failed");
        MK_HIP(hipMemcpyAsync(mine, g_host.send + (size_t)g_rank * (size_t)count, sizeof(double) * (size_t)count,
                              hipMemcpyHostToDevice, stream));
        MK_HIP(hipStreamSynchronize(stream));
        return MK_OK;
    }
    if (!g_comm) return mk_fail(MK_ERR_COMM, "reduce-scatter without a communicator");
    MK_NCCL(g_rccl.ReduceScatter(full, mine, (size_t)count, ncclDouble, ncclSum, g_comm, stream));
    return MK_OK;
}

// full = the ranks' blocks of `count` entries one after the other (mine may be full + rank * count: in place)
int mk_comm_allgather(const double *mine, double *full, int64_t count, hipStream_t stream) {
    if (g_host.active) {
        int rc = g_host.reserve(&g_host.send, &g_host.send_cap, (size_t)(count ? count : 1));
        if (rc == MK_OK) rc = g_host.reserve(&g_host.recv, &g_host.recv_cap, (size_t)count * (size_t)g_nranks + 1);
        if (rc != MK_OK) return rc;
        MK_HIP(hipMemcpyAsync(g_host.send, mine, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, stream));
        MK_HIP(hipStreamSynchronize(stream));
        if (g_host.allgather(g_host.send, count, g_host.recv) != 0) return mk_fail(MK_ERR_COMM, "host all-gather callback failed");
        MK_HIP(hipMemcpyAsync(full, g_host.recv, sizeof(double) * (size_t)count * (size_t)g_nranks, hipMemcpyHostToDevice, stream));
        MK_HIP(hipStreamSynchronize(stream));
        return MK_OK;
    }
    if (!g_comm) return mk_fail(MK_ERR_COMM, "all-gather without a communicator");
    MK_NCCL(g_rccl.AllGather(mine, full, (size_t)count, ncclDouble, g_comm, stream));
    return MK_OK;
}

extern "C" int mk_comm_unique_id(void *id128) {
    MK_ARG(id128 != nullptr);
    int rc = load_rccl();
    if (rc != MK_OK) return rc;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
    ncclUniqueId id;
    MK_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return MK_OK;
}

extern "C" int mk_comm_init(int nranks, int rank, const void *id128) {
    MK_REQUIRE_INIT();
    MK_ARG(nranks >= 1 && rank >= 0 && rank < nranks && id128);
    if (g_comm) return mk_fail(MK_ERR_STATE, "mk_comm_init: communicator already exists");
    int rc = load_rccl();
    if (rc != MK_OK) return rc;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    MK_NCCL(g_rccl.CommInitRank(&g_comm, nranks, id, rank));
    g_nranks = nranks;
    g_rank = rank;
    g_comm_halo = nullptr;
    if (nranks > 1 && g_rccl.CommSplit && !getenv("MK_SHARED_COMM")) {
        ncclComm_t c2 = nullptr;
        if (g_rccl.CommSplit(g_comm, 0, rank, &c2, nullptr) == ncclSuccess && c2) g_comm_halo = c2;
    }
    return MK_OK;
}

extern "C" int mk_comm_init_host(int nranks, int rank, mk_host_allreduce_fn allreduce, mk_host_exchange_fn exchange,
                                 mk_host_allgather_fn allgather) {
    MK_REQUIRE_INIT();
    MK_ARG(nranks >= 1 && rank >= 0 && rank < nranks && allreduce && exchange && allgather);
    if (g_comm || g_host.active) return mk_fail(MK_ERR_STATE, "mk_comm_init_host: communicator already exists");
    g_host.allreduce = allreduce;
    g_host.exchange = exchange;
    g_host.allgather = allgather;
    g_host.active = true;
    g_nranks = nranks;
    g_rank = rank;
    return MK_OK;
}

extern "C" int mk_comm_destroy(void) {
    if (g_host.active) {
        if (mk_ctx().ready) hipStreamSynchronize(mk_ctx().stream);
        if (g_host.send) hipHostFree(g_host.send);
        if (g_host.recv) hipHostFree(g_host.recv);
        g_host = HostComm();
    }
    if (g_comm) {
        if (mk_ctx().ready) hipDeviceSynchronize();
        if (g_comm_halo) g_rccl.CommDestroy(g_comm_halo);
        g_comm_halo = nullptr;
        g_rccl.CommDestroy(g_comm);
        g_comm = nullptr;
    }
    g_nranks = 1;
    g_rank = 0;
    return MK_OK;
}

extern "C" int mk_comm_info(int *nranks, int *rank) {
    if (nranks) *nranks = g_nranks;
    if (rank) *rank = g_rank;
    return MK_OK;
}

extern "C" int mk_comm_transport(int *kind, int *rccl_ranks, int *halo_comm_split) {
    if (kind) *kind = g_host.active ? 2 : (g_comm ? 1 : 0);
    if (rccl_ranks) {
        *rccl_ranks = 0;
        if (g_comm && g_rccl.CommCount) {
            int c = 0;
            if (g_rccl.CommCount(g_comm, &c) == ncclSuccess) *rccl_ranks = c;
        }
    }
    if (halo_comm_split) *halo_comm_split = g_comm_halo ? 1 : 0;
    return MK_OK;
}

extern "C" int mk_csr_set_exchange(mk_csr *A, int mode, int64_t n_local, int64_t n_halo, const int64_t *send_count,
                                   const int64_t *recv_count, const int32_t *send_idx_host) {
    MK_REQUIRE_INIT();
    MK_ARG(A && (mode == 0 || mode == 1) && n_local >= 0 && n_halo >= 0);
    if (A->comp_kind || A->host_fn)                          // (composites and shells have no arrays to split into tiles)
        return mk_fail(MK_ERR_UNSUPPORTED, "mk_csr_set_exchange: only a plain device matrix can carry an exchange plan "
                       "(partition the operands of a composite, not the composite)");
    if (A->plan.built && mk_fmt_march(A->plan.fmt)) mk_csr_plan_reset(A);   // (single-device formats: rebuilt as 4 / 5)
    MK_ARG(n_local == A->nrows);
    MK_ARG(n_local + n_halo == A->ncols);     // columns are already remapped to [local | halo]
    MkExchange &ex = A->ex;
    ex.n_local = n_local;
    ex.n_halo = n_halo;
    ex.send_count.assign(g_nranks, 0);
    ex.recv_count.assign(g_nranks, 0);
    ex.send_off.assign(g_nranks + 1, 0);
    ex.recv_off.assign(g_nranks + 1, 0);
    if (mode == 0) {
        MK_ARG(send_count && recv_count);
        for (int r = 0; r < g_nranks; ++r) {
            MK_ARG(send_count[r] >= 0 && recv_count[r] >= 0);
            ex.send_count[r] = send_count[r];
            ex.recv_count[r] = recv_count[r];
            ex.send_off[r + 1] = ex.send_off[r] + send_count[r];
            ex.recv_off[r + 1] = ex.recv_off[r] + recv_count[r];
        }
        MK_ARG(ex.recv_off[g_nranks] == n_halo);
        MK_ARG(ex.send_count[g_rank] == 0 && ex.recv_count[g_rank] == 0);
        ex.send_total = ex.send_off[g_nranks];
        if (ex.send_total > 0) {
            MK_ARG(send_idx_host != nullptr);
            MK_HIP(hipMalloc((void **)&ex.d_send_idx, sizeof(int32_t) * (size_t)ex.send_total));
            MK_HIP(hipMalloc((void **)&ex.d_send_buf, sizeof(double) * (size_t)ex.send_total));
            MK_HIP(hipMemcpy(ex.d_send_idx, send_idx_host, sizeof(int32_t) * (size_t)ex.send_total,
                             hipMemcpyHostToDevice));
        }
    } else {
        // all-gather: every rank contributes exactly n_halo / nranks entries; a rank that owns fewer rows (the last
        // one when n is not a multiple of the rank count) sends a zero-padded copy
        MK_ARG(n_halo % g_nranks == 0 && n_halo / g_nranks >= n_local);
        if (n_halo / g_nranks > n_local) {
            ex.send_total = n_halo / g_nranks;
            MK_HIP(hipMalloc((void **)&ex.d_send_buf, sizeof(double) * (size_t)ex.send_total));
            MK_HIP(hipMemset(ex.d_send_buf, 0, sizeof(double) * (size_t)ex.send_total));
        }
    }
    ex.mode = mode;
    if (mode == 0) return build_overlap_plan(A);
    return MK_OK;
}

namespace {
__global__ __launch_bounds__(MK_BLOCK) void col_extent_kernel(int64_t nnz, const int32_t *__restrict__ idx,
                                                              int *__restrict__ minmax) {
    int lo = 2147483647, hi = -1;
    for (int64_t j = (int64_t)blockIdx.x * MK_BLOCK + threadIdx.x; j < nnz; j += (int64_t)gridDim.x * MK_BLOCK) {
        const int c = idx[j];
        lo = c < lo ? c : lo;
        hi = c > hi ? c : hi;
    }
    atomicMin(&minmax[0], lo);
    atomicMax(&minmax[1], hi);
}

__global__ __launch_bounds__(MK_BLOCK) void col_remap_kernel(int64_t nnz, int32_t *__restrict__ idx, int64_t c0,
                                                             int64_t c1, int64_t lo_begin, int64_t n_local,
                                                             int64_t halo_lo, int mode) {
    for (int64_t j = (int64_t)blockIdx.x * MK_BLOCK + threadIdx.x; j < nnz; j += (int64_t)gridDim.x * MK_BLOCK) {
        const int64_t c = idx[j];
        int64_t v;
        if (mode == 1) v = n_local + c;
        else if (c < c0) v = n_local + (c - lo_begin);
        else if (c >= c1) v = n_local + halo_lo + (c - c1);
        else v = c - c0;
        idx[j] = (int32_t)v;
    }
}
}  // namespace

extern "C" int mk_csr_localize(mk_csr *A, int mode, int64_t col_begin, int64_t col_end, int64_t gathered_len,
                               int64_t *halo_lo, int64_t *halo_hi) {
    MK_REQUIRE_INIT();
    MK_ARG(A && (mode == 0 || mode == 1) && col_begin >= 0 && col_begin <= col_end && col_end <= A->ncols);
    if (A->comp_kind || A->host_fn)
        return mk_fail(MK_ERR_UNSUPPORTED, "mk_csr_localize: only a plain device matrix has columns to renumber");
    MK_ARG(col_end - col_begin == A->nrows);
    hipStream_t st = mk_ctx().stream;
    const int64_t n_local = A->nrows;
    int64_t lo = 0, hi = 0, lo_begin = col_begin;
    int grid = (int)((A->nnz + MK_BLOCK - 1) / MK_BLOCK);
    grid = grid < 1 ? 1 : (grid > 8192 ? 8192 : grid);
    if (mode == 0 && A->nnz > 0) {
        int *d_mm = nullptr;
        int h_mm[2] = {2147483647, -1};
        MK_HIP(hipMalloc((void **)&d_mm, sizeof(h_mm)));
        MK_HIP(hipMemcpyAsync(d_mm, h_mm, sizeof(h_mm), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(col_extent_kernel, dim3(grid), dim3(MK_BLOCK), 0, st, A->nnz, A->d_indices, d_mm);
        MK_HIP(hipMemcpyAsync(h_mm, d_mm, sizeof(h_mm), hipMemcpyDeviceToHost, st));
        MK_HIP(hipStreamSynchronize(st));
        MK_HIP(hipFree(d_mm));
        if (h_mm[0] < col_begin) lo = col_begin - h_mm[0];
        if (h_mm[1] >= col_end) hi = h_mm[1] - col_end + 1;
        lo_begin = col_begin - lo;
    }
    const int64_t new_cols = (mode == 1) ? n_local + gathered_len : n_local + lo + hi;
    if (new_cols > 2147483647LL) return mk_fail(MK_ERR_UNSUPPORTED, "localized column count exceeds int32");
    if (A->nnz > 0)
        hipLaunchKernelGGL(col_remap_kernel, dim3(grid), dim3(MK_BLOCK), 0, st, A->nnz, A->d_indices, col_begin,
                           col_end, lo_begin, n_local, lo, mode);
    MK_HIP(hipGetLastError());
    MK_HIP(hipStreamSynchronize(st));
    A->ncols = new_cols;
    A->loc_lo = (mode == 0) ? lo : 0;                        // (the brick march of a slab takes the neighbours' planes from there)
    A->loc_hi = (mode == 0) ? hi : 0;
    mk_csr_plan_reset(A);                                    // the columns changed: the windowed format is rebuilt
    if (halo_lo) *halo_lo = lo;
    if (halo_hi) *halo_hi = hi;
    return MK_OK;
}

// Start the exchange for the next product.  With an overlap plan the messages go to a second stream and the
// product runs in two launches (interior tiles now, boundary tiles after mk_exchange_wait); otherwise this is the
// plain in-stream exchange.
int mk_exchange_begin(const mk_csr *A, double *x_ext) {
    const MkExchange &ex = A->ex;
    ex.pending = ex.in_flight = false;
    if (ex.mode != 0 || !ex.d_tiles) return mk_exchange(A, x_ext);
    if (g_host.active) {                                     // host-staged transport: synchronous, same two launches
        int rc = mk_exchange(A, x_ext);
        ex.pending = (rc == MK_OK);
        return rc;
    }
    if (!g_comm) return mk_fail(MK_ERR_COMM, "halo exchange without a communicator");
    hipStream_t st = mk_ctx().stream;
    if (ex.send_total > 0) {
        int grid = (int)((ex.send_total + MK_BLOCK - 1) / MK_BLOCK);
        if (grid > 4096) grid = 4096;
        hipLaunchKernelGGL(pack_kernel, dim3(grid), dim3(MK_BLOCK), 0, st, ex.send_total, ex.d_send_idx, x_ext,
                           ex.d_send_buf);
    }
    // the messages read the packed buffer (written above) and write only the halo part of x_ext, which no kernel
    // touches until mk_exchange_wait; the previous exchange's messages were waited for before the buffer is reused
    MK_HIP(hipEventRecord(ex.ev_pack, st));
    MK_HIP(hipStreamWaitEvent(ex.comm_stream, ex.ev_pack, 0));
    MK_HIP(hipEventRecord(ex.ev_comm0, ex.comm_stream));
    ncclComm_t hc = g_comm_halo ? g_comm_halo : g_comm;
    MK_NCCL(g_rccl.GroupStart());
    for (int r = 0; r < g_nranks; ++r) {
        if (ex.send_count[r] > 0)
            MK_NCCL(g_rccl.Send(ex.d_send_buf + ex.send_off[r], (size_t)ex.send_count[r], ncclDouble, r, hc,
                                ex.comm_stream));
        if (ex.recv_count[r] > 0)
            MK_NCCL(g_rccl.Recv(x_ext + ex.n_local + ex.recv_off[r], (size_t)ex.recv_count[r], ncclDouble, r, hc,
                                ex.comm_stream));
    }
    MK_NCCL(g_rccl.GroupEnd());
    MK_HIP(hipEventRecord(ex.ev_comm, ex.comm_stream));
    ex.pending = ex.in_flight = true;
    ex.timed = true;
    return MK_OK;
}

int mk_exchange_wait(const mk_csr *A, hipStream_t stream) {
    const MkExchange &ex = A->ex;
    if (ex.in_flight) MK_HIP(hipStreamWaitEvent(stream, ex.ev_comm, 0));
    ex.pending = ex.in_flight = false;
    return MK_OK;
}

extern "C" int mk_comm_allreduce_host(double *vals, int64_t count) {
    MK_REQUIRE_INIT();
    MK_ARG(vals != nullptr && count >= 0 && count <= MK_MAXP);
    if (!mk_comm_active() || count == 0) return MK_OK;
    MkContext &c = mk_ctx();
    double *buf = c.d_scratch + 2 * MK_MAXP;                 // (the first two rows serve mk_dot / mk_nrm2)
    MK_HIP(hipMemcpyAsync(buf, vals, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, c.stream));
    int rc = mk_comm_allreduce_sum(buf, count, c.stream);
    if (rc != MK_OK) return rc;
    MK_HIP(hipMemcpyAsync(vals, buf, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, c.stream));
    MK_HIP(hipStreamSynchronize(c.stream));
    return MK_OK;
}

extern "C" int mk_csr_overlap_info(const mk_csr *A, int64_t *n_interior_tiles, int64_t *n_boundary_tiles) {
    MK_ARG(A != nullptr);
    if (n_interior_tiles) *n_interior_tiles = A->ex.d_tiles ? A->ex.n_int : 0;
    if (n_boundary_tiles) *n_boundary_tiles = A->ex.d_tiles ? A->ex.n_bnd : 0;
    return MK_OK;
}

extern "C" int mk_exchange(const mk_csr *A, double *x_ext) {
    MK_ARG(A && x_ext);
    const MkExchange &ex = A->ex;
    if (ex.mode < 0) return MK_OK;
    hipStream_t st = mk_ctx().stream;
    const double *ag_src = x_ext;                           // all-gather contribution of this rank
    if (ex.mode == 1 && ex.d_send_buf) {
        MK_HIP(hipMemcpyAsync(ex.d_send_buf, x_ext, sizeof(double) * (size_t)ex.n_local, hipMemcpyDeviceToDevice, st));
        ag_src = ex.d_send_buf;                             // (its tail beyond n_local stays zero)
    }
    if (ex.mode == 1 && g_host.active) {
        const size_t cnt = (size_t)(ex.n_halo / g_nranks);
        int rc = g_host.reserve(&g_host.send, &g_host.send_cap, cnt);
        if (rc == MK_OK) rc = g_host.reserve(&g_host.recv, &g_host.recv_cap, (size_t)ex.n_halo);
        if (rc != MK_OK) return rc;
        MK_HIP(hipMemcpyAsync(g_host.send, ag_src, sizeof(double) * cnt, hipMemcpyDeviceToHost, st));
        MK_HIP(hipStreamSynchronize(st));
        if (g_host.allgather(g_host.send, (int64_t)cnt, g_host.recv) != 0)
            return mk_fail(MK_ERR_COMM, "host all-gather callback failed");
        MK_HIP(hipMemcpyAsync(x_ext + ex.n_local, g_host.recv, sizeof(double) * (size_t)ex.n_halo,
                              hipMemcpyHostToDevice, st));
        MK_HIP(hipStreamSynchronize(st));
        return MK_OK;
    }
    if (ex.mode == 1) {
        if (!g_comm) return mk_fail(MK_ERR_COMM, "all-gather exchange without a communicator");
        const size_t cnt = (size_t)(ex.n_halo / g_nranks);
        MK_NCCL(g_rccl.AllGather(ag_src, x_ext + ex.n_local, cnt, ncclDouble, g_comm, st));
        return MK_OK;
    }
    if (ex.n_halo == 0 && ex.send_total == 0) return MK_OK;
    if (!g_comm && !g_host.active) return mk_fail(MK_ERR_COMM, "halo exchange without a communicator");
    if (ex.send_total > 0) {
        int grid = (int)((ex.send_total + MK_BLOCK - 1) / MK_BLOCK);
        if (grid > 4096) grid = 4096;
        hipLaunchKernelGGL(pack_kernel, dim3(grid), dim3(MK_BLOCK), 0, st, ex.send_total, ex.d_send_idx, x_ext,
                           ex.d_send_buf);
    }
    if (g_host.active) {
        int rc = g_host.reserve(&g_host.send, &g_host.send_cap, (size_t)(ex.send_total ? ex.send_total : 1));
        if (rc == MK_OK) rc = g_host.reserve(&g_host.recv, &g_host.recv_cap, (size_t)(ex.n_halo ? ex.n_halo : 1));
        if (rc != MK_OK) return rc;
        if (ex.send_total > 0)
            MK_HIP(hipMemcpyAsync(g_host.send, ex.d_send_buf, sizeof(double) * (size_t)ex.send_total,
                                  hipMemcpyDeviceToHost, st));
        MK_HIP(hipStreamSynchronize(st));
        if (g_host.exchange(g_host.send, ex.send_count.data(), ex.send_off.data(), g_host.recv, ex.recv_count.data(),
                            ex.recv_off.data()) != 0)
            return mk_fail(MK_ERR_COMM, "host exchange callback failed");
        if (ex.n_halo > 0)
            MK_HIP(hipMemcpyAsync(x_ext + ex.n_local, g_host.recv, sizeof(double) * (size_t)ex.n_halo,
                                  hipMemcpyHostToDevice, st));
        MK_HIP(hipStreamSynchronize(st));
        return MK_OK;
    }
    MK_NCCL(g_rccl.GroupStart());
    for (int r = 0; r < g_nranks; ++r) {
        if (ex.send_count[r] > 0)
            MK_NCCL(g_rccl.Send(ex.d_send_buf + ex.send_off[r], (size_t)ex.send_count[r], ncclDouble, r, g_comm, st));
        if (ex.recv_count[r] > 0)
            MK_NCCL(g_rccl.Recv(x_ext + ex.n_local + ex.recv_off[r], (size_t)ex.recv_count[r], ncclDouble, r, g_comm,
                                st));
    }
    MK_NCCL(g_rccl.GroupEnd());
    return MK_OK;
}

// ---------------------------------------------------------------------------------------------- timing helpers
// Collective: every rank must call these with the same arguments, in the same order.
extern "C" int mk_comm_time_exchange(const mk_csr *A, double *x_ext, int64_t reps, double *avg_us) {
    MK_REQUIRE_INIT();
    MK_ARG(A && x_ext && reps > 0 && avg_us);
    hipStream_t st = mk_ctx().stream;
    hipEvent_t e0, e1;
    MK_HIP(hipEventCreate(&e0));
    MK_HIP(hipEventCreate(&e1));
    int rc = mk_exchange(A, x_ext);                          // untimed first round (connections, buffers)
    if (rc != MK_OK) return rc;
    MK_HIP(hipEventRecord(e0, st));
    for (int64_t k = 0; k < reps; ++k)
        if ((rc = mk_exchange(A, x_ext)) != MK_OK) return rc;
    MK_HIP(hipEventRecord(e1, st));
    MK_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    MK_HIP(hipEventElapsedTime(&ms, e0, e1));
    *avg_us = 1e3 * (double)ms / (double)reps;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return MK_OK;
}

extern "C" int mk_comm_time_allreduce(int64_t count, int64_t reps, double *avg_us) {
    MK_REQUIRE_INIT();
    MK_ARG(count > 0 && count <= MK_MAXP && reps > 0 && avg_us);
    if (!mk_comm_active()) {
        *avg_us = 0.0;
        return MK_OK;
    }
    MkContext &c = mk_ctx();
    double *buf = c.d_scratch + 2 * MK_MAXP;
    MK_HIP(hipMemsetAsync(buf, 0, sizeof(double) * (size_t)count, c.stream));
    hipEvent_t e0, e1;
    MK_HIP(hipEventCreate(&e0));
    MK_HIP(hipEventCreate(&e1));
    int rc = mk_comm_allreduce_sum(buf, count, c.stream);
    if (rc != MK_OK) return rc;
    MK_HIP(hipEventRecord(e0, c.stream));
    for (int64_t k = 0; k < reps; ++k)
        if ((rc = mk_comm_allreduce_sum(buf, count, c.stream)) != MK_OK) return rc;
    MK_HIP(hipEventRecord(e1, c.stream));
    MK_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    MK_HIP(hipEventElapsedTime(&ms, e0, e1));
    *avg_us = 1e3 * (double)ms / (double)reps;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return MK_OK;
}

// duration of the last overlapped halo message group on the second stream (0 if there was none); call after a sync
extern "C" int mk_csr_comm_last_us(const mk_csr *A, double *us) {
    MK_ARG(A && us);
    *us = 0.0;
    const MkExchange &ex = A->ex;
    if (!ex.timed || !ex.ev_comm0 || !ex.ev_comm) return MK_OK;
    if (hipEventSynchronize(ex.ev_comm) != hipSuccess) return MK_OK;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ex.ev_comm0, ex.ev_comm) == hipSuccess) *us = 1e3 * (double)ms;
    return MK_OK;
}
