// mk_internal.h -- shared host/device plumbing of libmikrylov (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/mikrylov.h"

// --------------------------------------------------------------------------------------
// geometry shared by every kernel
// --------------------------------------------------------------------------------------
constexpr int MK_BLOCK = 256;         // 4 wave64 per workgroup
constexpr int MK_WAVE = 64;
constexpr int MK_MAXP = 2048;         // partial-sum slots per reduction (= max grid of a producer)
constexpr int MK_ROWS_PER_TILE = 256; // SpMV: one row per thread in the row-sum phase
constexpr int MK_SPMV_TILE = 2048;    // SpMV: products staged in LDS per pass (16 KiB)
constexpr int MK_NSCAL = 160;         // device scalar file per solver
constexpr int MK_NDOT = 6;           // reduction slots per solver (MK_MAXP doubles each)
constexpr int MK_CARRY_SLOTS = 4;     // fused dots whose per-lane accumulators a stepped product carries between its launches

// --------------------------------------------------------------------------------------
// host context
// --------------------------------------------------------------------------------------
struct MkContext {
    bool ready = false;
    int device = -1;
    hipStream_t stream = nullptr;
    int num_cu = 0;
    std::string last_error;
    int pending_rc = 0;              // error raised inside an enqueue helper that cannot return it (host callbacks)
    // pinned staging for scalar read-backs
    double *h_scratch = nullptr;     // MK_MAXP * MK_NDOT doubles
    double *d_scratch = nullptr;
    // vector arena (mk_arena_reserve): one allocation made BEFORE the matrix, from which solvers carve their vectors
    char *arena = nullptr;
    size_t arena_size = 0, arena_off = 0;
    int arena_live = 0;              // vectors carved and not yet returned
    double *pen_dump = nullptr;      // dump rows of the general-geometry brick march (mk_pen_dump, mk_format.hip)
};

MkContext &mk_ctx();
int mk_fail(int code, const char *fmt, ...);

#define MK_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return mk_fail(MK_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                           __FILE__, __LINE__);                                               \
    } while (0)

#define MK_REQUIRE_INIT()                                                              \
    do {                                                                               \
        if (!mk_ctx().ready) {                                                         \
            int _r = mk_init(0);                                                       \
            if (_r != MK_OK) return _r;                                                \
        }                                                                              \
    } while (0)

#define MK_ARG(cond)                                                                          \
    do {                                                                                      \
        if (!(cond)) return mk_fail(MK_ERR_ARG, "argument check failed: %s (%s:%d)", #cond,   \
                                    __FILE__, __LINE__);                                      \
    } while (0)

// device vectors that cross the C ABI are read and written in 16-byte pairs
#define MK_ALIGNED16(p) ((((uintptr_t)(p)) & 15) == 0)

// --------------------------------------------------------------------------------------
// exchange plan (multi-GPU); empty for a single-device matrix
// --------------------------------------------------------------------------------------
constexpr int MK_CSR_PAD = 8;      // padding entries behind indices / data / slots / codes (aligned 8-entry reads of the SpMV kernel)

struct MkExchange {
    int mode = -1;                 // -1 none, 0 halo send/recv, 1 allgather
    int64_t n_local = 0, n_halo = 0;
    std::vector<int64_t> send_count, recv_count, send_off, recv_off;
    int32_t *d_send_idx = nullptr; // gather list (device)
    double *d_send_buf = nullptr;  // packed values to send
    int64_t send_total = 0;
    bool contiguous_send = false;  // every rank's list is a contiguous range: send straight from x
    // overlap of the halo exchange with the product (halo mode): tiles whose rows reference only owned columns
    // ("interior") are multiplied while the messages travel on a second stream, the others afterwards
    int32_t *d_tiles = nullptr;    // interior tile ids followed by boundary tile ids
    int64_t n_int = 0, n_bnd = 0;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_pack = nullptr, ev_comm0 = nullptr, ev_comm = nullptr;
    mutable bool pending = false;  // an exchange was started and the next product must run in two parts
    mutable bool in_flight = false; // ... and its messages are on comm_stream (wait for ev_comm)
    mutable bool timed = false;     // ev_comm0 / ev_comm bracket a message group
};

// Windowed tile format of a matrix (mk_format.hip), built on first use of the matrix in a product.
constexpr int MK_WCHUNK = 128;     // doubles per window chunk (one wave-level 16-byte load)
constexpr int MK_WCHUNKS_MAX = 16; // chunks per tile (4 per wave): 16 KiB of LDS windows at most
constexpr int MK_WCHUNKS_WIDE = 32; // ... of the wide cover (8 per wave, 32 KiB), tiles of up to MK_WIDE_TILE nonzeros
constexpr int MK_WIDE_TILE = 8192;
static inline bool mk_fmt_march(int fmt) { return fmt >= 9 && fmt <= 11; }   // the brick-march formats (mk_spmv_fmt9.h)
struct MkPlan {
    bool built = false;
    int fmt = 0;                   // 9 z-marching bricks: pattern byte per row + dictionary, 7-point-class matrices (below);
                                   // 0 plain CSR, 1 windows + uint16 slots, 2 windows + slots + value dictionary,
                                   // 3 plain CSR, tile resident in LDS, gathers ordered by column block,
                                   // 4 windows + dictionary + row patterns (one byte per row)
                                   // 5 windows + row patterns + raw values in tile-sliced ELL order
                                   // 6 wide tiles: slots + values in tile-sliced ELL order; 7 wide: row patterns +
                                   // values; 8 wide: row patterns + dictionary
    int wchunks = 0;               // max chunks of a tile
    int ndict = 0;
    int64_t covered = 0;           // tiles on the windowed path
    uint16_t *d_slots = nullptr;
    int32_t *d_wg = nullptr;
    uint32_t *d_wn = nullptr;
    uint32_t *d_pk = nullptr;      // fmt 2: {slot | code << 16} per nonzero
    double *d_dict = nullptr;
    // fmt 4 (row patterns): one byte per row, a table of `pmax` words per pattern (mk_format.hip)
    uint8_t *d_pid = nullptr;
    uint32_t *d_pat = nullptr;
    uint8_t *d_plen = nullptr;
    int npat = 0, pmax = 0;
    // fmt 5 (row patterns + streamed values): the values in tile-sliced ELL order, per tile {block start / 256, width}
    double *d_sval = nullptr;
    int32_t *d_sdesc = nullptr;
    int64_t sell_entries = 0;      // doubles in d_sval (256 * sum of the tile widths)
    // fmt 6, 7, 8 (wide tiles): the cover has 32 chunks per tile (8 per wave) and tiles of up to 8192 nonzeros; fmt 6
    // streams the LDS slots in tile-sliced ELL order beside the values; d_sdesc then holds four ints per tile
    bool wide = false;
    uint16_t *d_sslot = nullptr;
    int64_t slot_entries = 0;      // uint16 in d_sslot
    void *d_ptab = nullptr;        // fmt 8: the pattern table as {offset, value} entries, read through the scalar cache
    int32_t *d_pinfo = nullptr;    // ... and per pattern {entries | position of the diagonal << 8}
    // fmt 3 (resident tiles, column phases): plain CSR arrays, only launch parameters
    int rt_cap = 0;                // LDS capacity per tile in nonzeros (max tile stream length rounded up to 256)
    int rt_k = 1;                  // column phases
    int rt_w = 0;                  // columns per phase
    int rt_c0 = 0;                 // first column of phase 0 (a column block's first column; 0 otherwise)
    int rt_reg = 0;                // rows of <= 5 entries and more tiles than resident workgroups: pairs of tiles (mk_spmv_fmt3r.h)
    int max_row = 0;               // longest row (entries) of the matrix
    // fmt 9 (z-marching bricks, mk_spmv_fmt9.h): the three strides, planes, bricks per line / per plane, planes per chunk
    // and chunks; d_pid holds the pattern byte per row, d_ptab 64 bytes per pattern {7 values, mask}, npat their number
    int64_t pen_L = 0, pen_P = 0;
    int pen_nz = 0, pen_bx = 0, pen_bpp = 0, pen_zc = 0, pen_chunks = 0;
    // general geometry (round 6): 2 = partly empty bricks / unaligned pairs (GEN kernels only), 0 = whole aligned bricks;
    // bricks per XCD of the XCD-contiguous deal (0: round robin); lines per plane
    int pen_gen = 0, pen_per = 0, pen_ny = 0;
    int pen_nol = 0;                                        // 5-point matrix marched line by line: no +-L entries
    // ... of one rank's slab of planes (columns localised to [own | plane below | plane above], mk_csr_localize mode 0): where
    // the neighbours' planes start in the product's input vector (-1: the slab has no such neighbour)
    int64_t pen_xlo = -1, pen_xhi = -1;
    double *d_carry = nullptr;     // per-lane accumulators of fused dots between the launches of a stepped product
    // column blocks (plain-CSR matrices whose x does not fit an XCD's L2): A = [A_0 | A_1 | ...] by column range,
    // each block a CSR matrix over all rows; a product runs block after block with the running row sums carried
    std::vector<struct mk_csr *> cblocks;
    double *d_cbsum = nullptr;     // nrows running sums
};

// grid of device matrices seen as one operator (mk_csr_create_block; reference linop/blkop.py:8-152, :154-257)
struct MkBlockGrid {
    int nbr = 0, nbc = 0;
    std::vector<const struct mk_csr *> blk;    // row major; null = zero block
    std::vector<int64_t> roff, coff;           // nbr + 1 / nbc + 1 offsets into y / x
    double *d_xtmp = nullptr;                  // aligned copy of an x slice that starts at an odd offset
};

// restriction of a device matrix to chosen rows and columns (mk_csr_create_reduced; reference linop/linop.py:560-587)
struct MkReduced {
    int32_t *d_rows = nullptr, *d_cols = nullptr;   // row_indices / col_indices on the device
    double *d_z = nullptr;                           // x scattered into a zero vector of the base's width
    double *d_t = nullptr;                           // the base's product
};

struct mk_csr {
    int64_t nrows = 0, ncols = 0, nnz = 0;
    int64_t loc_lo = 0, loc_hi = 0; // mk_csr_localize mode 0: widths of the lower / upper halo window (columns nrows .. ncols-1)
    int32_t *d_indptr = nullptr;
    int32_t *d_indices = nullptr;
    double *d_data = nullptr;
    int64_t ntiles = 0;            // ceil(nrows / MK_ROWS_PER_TILE)
    MkExchange ex;
    bool alias = false;            // composed operator: the arrays belong to another mk_csr ...
    const mk_csr *base = nullptr;  // ... this one (which also owns the windowed format)
    mutable MkPlan plan;
    int want_fmt = -1;             // mk_csr_set_format: -1 = library default (MK_SPMV_FORMAT or 2)
    int want_cb_kb = -1;           // mk_csr_set_colblocks: -1 = library default (MK_COLBLOCK_KB or off)
    int want_map = -1, want_stripe = 0, want_plane = 0;   // mk_csr_set_tile_order: -1 = library default
    int want_nt = -1;              // mk_csr_set_tile_order: non-temporal loads of the streamed matrix data (-1 = default)
    // sum / difference / product of two device matrices (mk_csr_create_sum / _product): no arrays of its own
    int comp_kind = 0;             // 0 none, 1 A + B, 2 A - B, 3 A * B, 4 block grid, 5 reduced (comp_a restricted)
    MkReduced *red = nullptr;      // comp_kind 5
    const mk_csr *comp_a = nullptr, *comp_b = nullptr;
    MkBlockGrid *grid = nullptr;   // comp_kind 4
    // operands of composites are borrowed: they count their dependents, and a matrix destroyed while composites still
    // use it lives on until the last of them goes (mk_csr_destroy)
    mutable int dependents = 0;
    // brick-march formats (9 / 10) are chosen per matrix but pay only for loops whose product epilogue loads nothing inside
    // the march's pipelined loop (CG, plain products): -1 no solver has asked yet (march allowed), 1 the last solver created
    // on this matrix was CG, 0 another loop (its products keep the windowed formats 4 / 5); mk_csr_march_pref, mk_format.hip
    mutable int march_pref = -1;
    mutable bool no_sym = false;   // the last solver created on the matrix is not CG: a request for format 11 (symmetric march,
                                   // CG's and plain products' kernels only) is served as format 10
    mutable int solver_users = 0;  // live solvers on this matrix (the preference only changes while there is none)
    mutable bool doomed = false;
    double *d_comp_tmp = nullptr;  // first product's row sums (sum / difference) or B x (product)
    // matrix-free operator (mk_csr_create_callback): no arrays; products come from a host callback
    mk_matvec_fn host_fn = nullptr;
    void *host_user = nullptr;
    int host_transpose = 0;
    double *h_cb_in = nullptr, *h_cb_out = nullptr;    // pinned staging (x_len / nrows doubles)
    double *d_cb_in = nullptr, *d_cb_out = nullptr;    // device: materialised input, product
    int *d_cb_go = nullptr;                            // device flag: the gate let this product through
    // mk_csr_set_row_block: this matrix is one rank's block of ROWS of a taller operator whose column space is
    // replicated on every rank (least-squares solvers: u sliced, v whole, A' u summed over the ranks)
    int row_block = 0;             // 1: n-space vectors whole on every rank; 2: sliced like the ranks' column blocks
    int32_t nops = 0;              // row program (mk_csr_compose)
    mk_rowop ops[MK_ROWPROG_MAX] = {};
    // length of the vector an SpMV reads (ncols, or n_local + n_halo with a halo plan)
    int64_t x_len() const { return ex.mode == 0 ? ex.n_local + ex.n_halo : ncols; }
    bool is_plain() const { return !comp_kind && !host_fn && ex.mode < 0; }
};

int mk_csr_alloc(int64_t nrows, int64_t ncols, int64_t nnz, mk_csr **out);
void mk_release_operand(const mk_csr *B);   // a borrower lets go: the matrix is destroyed now if its owner already asked for it
void mk_csr_plan_reset(const mk_csr *A);    // drop the windowed format (it is rebuilt on the next product)
void mk_csr_march_pref(const mk_csr *A, int pref);   // a solver is being created on A: 1 = CG, 0 = any other loop
void mk_csr_count_users(const mk_csr *A, int delta); // a solver was created on / removed from A (and whatever A is composed of)

// grid sizes -------------------------------------------------------------------------
// Persistent-style grids: at most `cap` workgroups which stride over the work.  The caps are tuning
// knobs (MK_GRID_STREAM / MK_GRID_SPMV in the environment override them, <= MK_MAXP).
int mk_cap_stream();
int mk_cap_spmv();
static inline int mk_grid_stream(int64_t n) {          // BLAS-1 style kernels: 2 doubles per thread per step
    int64_t g = (n + 2 * MK_BLOCK - 1) / (2 * MK_BLOCK);
    if (g < 1) g = 1;
    const int cap = mk_cap_stream();
    return (int)(g > cap ? cap : g);
}
static inline int mk_grid_spmv(int64_t ntiles) {
    if (ntiles < 1) ntiles = 1;
    const int cap = mk_cap_spmv();
    int g = (int)(ntiles > cap ? cap : ntiles);
    if (g >= 8) g -= g % 8;             // multiples of 8: one equal share of workgroups per XCD
    return g;
}

// --------------------------------------------------------------------------------------
// device helpers
// --------------------------------------------------------------------------------------
#ifdef __HIPCC__

// Sum over the workgroup, identical value returned to every thread.  Fixed tree:
// shuffle-down 32,16,8,4,2,1 inside each wave64, then the four wave sums added in wave order.
__device__ __forceinline__ double mk_block_sum(double v, double *s4) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();                 // s4 may still be read from a previous call
    if (lane == 0) s4[wave] = v;
    __syncthreads();
    return ((s4[0] + s4[1]) + s4[2]) + s4[3];
}

// Total of `np` partial sums written by a previous kernel (np <= MK_MAXP): thread t adds
// slots t, t+256, ... in order, then mk_block_sum.  Every workgroup of the consumer kernel
// does this redundantly and obtains bit-identical totals, so no extra kernel or fence is needed.
__device__ __forceinline__ double mk_total(const double *part, int np, double *s4) {
    // all (<= MK_MAXP / MK_BLOCK = 8) loads of a lane are issued together: the partials were written by other XCDs
    // and come from the Infinity Cache, ~1 us per round trip -- one round trip instead of up to eight.  Slots past
    // np contribute +0.0, which leaves the running sum unchanged bit for bit.
    constexpr int K = MK_MAXP / MK_BLOCK;
    double t[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int i = (int)threadIdx.x + k * MK_BLOCK;
        t[k] = part[i < np ? i : 0];
        t[k] = (i < np) ? t[k] : 0.0;
    }
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) v += t[k];
    return mk_block_sum(v, s4);
}

// The same total in two steps, for kernels that issue every load of their prologue (halt word, partial sums, scalars)
// BEFORE waiting for any of them: a dependent chain of three memory round trips (~0.7 us each: the data was written
// by other XCDs) becomes one.  Identical arithmetic to mk_total.
struct MkTotalRegs {
    double t[MK_MAXP / MK_BLOCK];
};
__device__ __forceinline__ void mk_total_issue(const double *part, int np, MkTotalRegs &R) {
#pragma unroll
    for (int k = 0; k < MK_MAXP / MK_BLOCK; ++k) {
        const int i = (int)threadIdx.x + k * MK_BLOCK;
        R.t[k] = part[i < np ? i : 0];
    }
}
__device__ __forceinline__ double mk_total_finish(const MkTotalRegs &R, int np, double *s4) {
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < MK_MAXP / MK_BLOCK; ++k) {
        const int i = (int)threadIdx.x + k * MK_BLOCK;
        v += (i < np) ? R.t[k] : 0.0;
    }
    return mk_block_sum(v, s4);
}

// Halting protocol.  Kernel number q of a solver reads halt[q & 1] and (one thread) writes
// halt[(q + 1) & 1] = halt_in | new condition, so no kernel reads the word it writes and a
// raised flag is carried forward by every later kernel, which then does no work: the solver
// state stays frozen exactly where the reference's `while` condition failed.
struct MkHalt {
    int *flags;      // 2 ints
    int parity;      // q & 1
    int ptail;       // multi-rank runs: consumers add MK_MAXP entries of every (all-reduced) partial-sum slot, so a
                     // producer clears its slots from gridDim.x up to here; 0 = nothing to clear (single rank)
    __device__ __forceinline__ bool in() const { return flags[parity] != 0; }
    __device__ __forceinline__ void out(bool v) const { flags[parity ^ 1] = v ? 1 : 0; }
    // a slot is written by kernels of different grid sizes (stream vs SpMV) over a solve: without this, entries of
    // the wider producer would survive, already all-reduced, behind the narrower one's
    template <int NACC, int SLOT0>
    __device__ __forceinline__ void clear_tail(double *partials, int first = -1) const {
        if (NACC > 0 && ptail > 0 && blockIdx.x == 0) {
            const int from = (first >= 0) ? first : (int)gridDim.x;
#pragma unroll
            for (int d = 0; d < NACC; ++d)
                for (int i = from + (int)threadIdx.x; i < ptail; i += MK_BLOCK)
                    partials[(SLOT0 + d) * MK_MAXP + i] = 0.0;
        }
    }
};

#endif  // __HIPCC__
