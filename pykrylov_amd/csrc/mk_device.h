// mk_device.h -- the two kernel skeletons every solver phase is built from (gfx950).
//
//  * mk_spmv_kernel<Epi>   CSR-stream SpMV (K1 of SURVEY.md 2.2) with a per-row epilogue
//                          functor, so the dot that always follows `op * v` in the
//                          reference (e.g. cg.py:115-117) and neighbouring axpy updates are
//                          fused into the same pass over the matrix.
//  * mk_stream_kernel<Op>  fused BLAS-1 pass (K2-K6): a prologue that every workgroup runs
//                          redundantly to turn the previous kernel's partial sums into the
//                          reference's scalars (alpha, beta, Givens rotations ...), then a
//                          16-byte-per-lane grid-stride sweep doing the vector updates and
//                          accumulating up to NACC new dots.
//
// Both write one partial sum per workgroup per dot (fixed tree, deterministic run to run)
// and obey the MkHalt protocol.  Everything is compiled with -ffp-contract=off: each
// multiply and add rounds separately, exactly like the NumPy expressions being replaced.
#pragma once
#include <type_traits>
#include <utility>

#include "mk_internal.h"

struct MkCsrView {
    const int32_t *indptr;
    const int32_t *indices;
    const double *data;
    int64_t nrows;
    int64_t ntiles;
    int map;                 // tile order: 0 round robin; 1 each XCD sweeps its own contiguous eighth of the tiles
                             // (cache-resident problems); 2 every step of the grid is split into eight XCD-contiguous blocks
    int nops;                // row program of a composed operator (mk_csr_compose); 0 for a plain matrix
    mk_rowop ops[MK_ROWPROG_MAX];
    // a launch may cover a subset of the tiles (overlap of the halo exchange, mk_comm.hip): `tiles` lists them
    // (null: all tiles 0 .. ntl-1), `poff` is where this launch's partial sums start, `part` 0 = whole product,
    // 1 = first of two launches, 2 = second (repeats the gate's decision without its side effects)
    const int32_t *tiles;
    int64_t ntl;
    int poff;
    int part;
    // windowed tile format (mk_format.hip); fmt 0: none of this is read
    int fmt;                 // 0 plain CSR (gathers), 1 windows + uint16 LDS slots, 2 windows + slots + value dictionary,
                             // 3 plain CSR with the tile resident in LDS and the gathers ordered by column block,
                             // 4 windows + dictionary + one pattern byte per row instead of a word per nonzero
    int wchunks;             // LDS chunks (128 doubles each) reserved for the x windows of a tile
    int ndict;
    const uint16_t *slots;   // per nonzero: position of its x entry in the tile's LDS window buffer
    const int32_t *wg;       // per tile and wave: global start of the wave's (<= 4) window chunks; bit 0 of [0]: tile eligible
    const uint32_t *wn;      // per tile and wave: half lengths of those chunks, one byte each
    const uint32_t *pk;      // fmt 2, per nonzero: {LDS slot : 16 | index of its value in `dict` : 8}
    const double *dict;
    // fmt 4: per row the number of its pattern; pattern p = words pat[p * pmax ..] = {slot - lane : 16 | code : 8}
    const uint8_t *pid;
    const uint32_t *pat;
    const uint8_t *plen;
    int npat, pmax;
    int allwin;              // every tile of the matrix has windows (no tile ever takes the gather path)
    // resident tiles (fmt 3): LDS capacity per tile in nonzeros (multiple of 256), column phases and their width
    int rt_cap, rt_k, rt_w;
    // column-blocked products (fmt 0, 3): the row sums start from sum_in[r] instead of +0.0 (null: +0.0)
    const double *sum_in;
    // matrix-free operators (host callback): cb_mode 1 = materialise the product's input vector (`vin[j] = xin(x[j])`,
    // `xlen` entries) and report the gate's decision in *cb_go; 2 = the row sums are the entries of `y_ext`
    int cb_mode;
    int64_t xlen;
    double *vin;
    const double *y_ext;
    int *cb_go;
};

// Working sets that fit the 256 MiB Infinity Cache profit from XCD-local tile ranges (every x line is then
// fetched by one L2 instead of by up to five); beyond that size eight distant sweeps cost more in DRAM
// locality than they save (measured: 2-D n=1e6 +5 %, 3-D 512^3 -8 %), so large problems sweep in one front.
static inline int mk_xcd_chunks(const mk_csr *A) {
    const int64_t bytes = 12 * A->nnz + 4 * (A->nrows + 1) + 8 * (A->x_len() + A->nrows) * 3;
    return bytes <= (int64_t)200 * 1024 * 1024 ? 1 : 0;
}
const MkPlan *mk_csr_plan(const mk_csr *A);      // mk_format.hip: builds the windowed format on first use
static inline int mk_tile_map(const mk_csr *A) {
    if (A->comp_kind) return mk_tile_map(A->comp_kind == 3 ? A->comp_a : A->comp_b);
    if (A->host_fn) return 0;
    static const char *env = getenv("MK_SPMV_MAP");
    if (env) return atoi(env);
    if (mk_xcd_chunks(A)) return 1;
    // beyond the Infinity Cache: one front (0) -- except for the pattern kernel, which moves so little per tile that it
    // runs into the fabric on re-fetched x windows: XCD-contiguous blocks within every step of the grid keep neighbouring
    // tiles' windows in one L2 (512^3: fabric reads 4.4 -> 3.6 GB, 838 -> 790 us)
    const MkPlan *P = mk_csr_plan(A);
    return (P && P->fmt == 4) ? 2 : 0;
}

constexpr int MK_PROD_LD = MK_BLOCK + 1;             // (product staging buffer of the SpMV kernels, see below)
constexpr int MK_PROD_LDS = 8 * MK_PROD_LD;          // doubles reserved for products (>= MK_SPMV_TILE of the gather path)

int mk_host_product(const mk_csr *A, hipStream_t st);   // mk_core.hip: D2H, host callback, H2D of a matrix-free operator

// SpMV grid = the workgroups that are resident at once (persistent tiles; a second round only adds a tail).
// CSR path: 4 per CU at <= 64 registers, twice as many smaller shares while the problem is cache resident.
// Windowed paths: 4 per CU (1024 workgroups); the dictionary kernel takes 5 per CU while the problem is cache
// resident -- measured on 512^3 and 2-D n = 1e6 (tools/sweep_fmt.sh): every other count loses 5-25 %.  The pattern
// kernel (fmt 4) ingests so little per tile that it is bound by the latency of its window copies: 7 per CU
// (512^3: 1024 / 1536 / 1792 workgroups -> 1.26 / 1.01 / 0.91 ms).
static inline int mk_grid_spmv_for(const mk_csr *A) {
    if (A->comp_kind) return mk_grid_spmv_for(A->comp_kind == 3 ? A->comp_a : A->comp_b);   // the final launch's matrix
    int g = mk_grid_spmv(A->ntiles);
    if (getenv("MK_GRID_SPMV") || A->host_fn) return g;
    const MkPlan *P = mk_csr_plan(A);
    int64_t cap = mk_cap_spmv();
    if (P && P->fmt == 3) {                                  // as many as fit at once with one tile in LDS each
        int g3 = (int)(A->ntiles > MK_MAXP ? MK_MAXP : A->ntiles);
        if (g3 >= 8) g3 -= g3 % 8;
        return g3;
    }
    if (P && P->fmt == 4) {                                  // up to 7 per CU, as many as LDS holds (8 per CU: 512^3 +3 %,
                                                             // 2-D n = 1e6 -4 %, and MINRES' epilogue spills at 64 registers)
        int64_t top = 128 * (int64_t)P->wchunks + 2;
        if (P->covered != A->ntiles && top < MK_PROD_LDS) top = MK_PROD_LDS;
        const int64_t lds = 8 * (top + MK_BLOCK) + 16 * (int64_t)(P->npat * P->pmax + 1) + 2560;   // + static arrays
        int64_t per_cu = (160 * 1024) / lds;
        per_cu = per_cu > 7 ? 7 : (per_cu < 1 ? 1 : per_cu);
        cap = 256 * per_cu;
    }
    else if (P && P->fmt == 2) cap = mk_xcd_chunks(A) ? 1280 : 1024;
    else if (P && P->fmt == 1) cap = 1024;
    else if (mk_xcd_chunks(A)) cap = 2 * cap > MK_MAXP ? MK_MAXP : 2 * cap;
    g = (int)(A->ntiles > cap ? cap : (A->ntiles < 1 ? 1 : A->ntiles));
    if (g >= 8) g -= g % 8;
    return g;
}

// grids of the two launches of an overlapped (halo) product: the interior tiles (most of the matrix) get what a whole
// product would get, less the few workgroups kept for the boundary tiles; together at most MK_MAXP (the two launches
// write disjoint ranges of the partial-sum slots)
static inline void mk_grid_spmv_parts(const mk_csr *A, int64_t n_int, int64_t n_bnd, int *g_int, int *g_bnd) {
    const int cap = mk_grid_spmv_for(A);
    int g2 = (int)(n_bnd < 1 ? 1 : (n_bnd > 256 ? 256 : n_bnd));
    if (g2 >= 8) g2 -= g2 % 8;
    int room = MK_MAXP - g2;
    if (room > cap) room = cap;
    int g1 = (int)(n_int < 1 ? 1 : (n_int > room ? room : n_int));
    if (g1 >= 8) g1 -= g1 % 8;
    *g_int = g1;
    *g_bnd = g2;
}

static inline MkCsrView mk_view(const mk_csr *A) {
    MkCsrView v{};
    v.indptr = A->d_indptr;
    v.indices = A->d_indices;
    v.data = A->d_data;
    v.nrows = A->nrows;
    v.ntiles = A->ntiles;
    v.map = mk_tile_map(A);
    v.nops = A->nops;
    for (int k = 0; k < A->nops; ++k) v.ops[k] = A->ops[k];
    v.tiles = nullptr;
    v.ntl = A->ntiles;
    v.poff = 0;
    v.part = 0;
    const MkPlan *P = mk_csr_plan(A);
    v.fmt = P ? P->fmt : 0;
    if (v.fmt == 3) {
        v.rt_cap = P->rt_cap;
        v.rt_k = P->rt_k;
        v.rt_w = P->rt_w;
    } else if (v.fmt) {
        v.wchunks = P->wchunks;
        v.ndict = P->ndict;
        v.slots = P->d_slots;
        v.wg = P->d_wg;
        v.wn = P->d_wn;
        v.pk = P->d_pk;
        v.dict = P->d_dict;
        v.pid = P->d_pid;
        v.pat = P->d_pat;
        v.plen = P->d_plen;
        v.npat = P->npat;
        v.pmax = P->pmax;
        v.allwin = (P->covered == A->ntiles) ? 1 : 0;
    }
    return v;
}

// view of the interior (part 1) or boundary (part 2) tiles of a partitioned matrix; poff2 = grid of part 1
static inline MkCsrView mk_view_part(const mk_csr *A, int part, int poff2) {
    MkCsrView v = mk_view(A);
    v.tiles = A->ex.d_tiles + (part == 2 ? A->ex.n_int : 0);
    v.ntl = (part == 2) ? A->ex.n_bnd : A->ex.n_int;
    v.poff = (part == 2) ? poff2 : 0;
    v.part = part;
    v.map = 0;
    return v;
}

#ifdef __HIPCC__

typedef double mk_d2 __attribute__((ext_vector_type(2)));
typedef int mk_i2 __attribute__((ext_vector_type(2)));
typedef int mk_i4 __attribute__((ext_vector_type(4)));

// Composed operators (mk_csr_compose): the reference evaluates `alpha * op`, `op + D`, `op - D` as one NumPy
// expression per node on the product vector (linop.py:307-330, :375-426); the same expressions, in the same order,
// are applied here to the finished row sum.  x_r is the entry of the vector the product is applied to (after the
// epilogue's on-the-fly scaling, e.g. MINRES' v = y / beta).
template <class Epi>
__device__ __forceinline__ double mk_rowprog(const MkCsrView &A, double t, const double *__restrict__ x, int64_t r,
                                             const Epi &epi) {
    double xr = 0.0;
    bool have_x = false;
#pragma unroll
    for (int k = 0; k < MK_ROWPROG_MAX; ++k) {
        if (k >= A.nops) break;
        const mk_rowop op = A.ops[k];
        if (op.code == MK_ROW_SCALE) {
            t = op.scale * t;
            continue;
        }
        if (!have_x) {
            xr = epi.xin(x[r]);
            have_x = true;
        }
        double term = xr;
        if (op.diag) term = op.diag[r] * term;
        if (op.has_scale) term = op.scale * term;
        if (op.code == MK_ROW_ADD) t = t + term;
        else if (op.code == MK_ROW_SUB) t = t - term;
        else t = term - t;
    }
    return t;
}

// ---------------------------------------------------------------------------------------
// CSR-stream SpMV.  A workgroup owns 256 consecutive rows per tile and forms every row sum LEFT TO RIGHT, the
// rounding sequence of a scalar CSR loop (bit-identical to the oracle), in two passes: pass 1 walks the tile's
// nonzeros in storage order with wide coalesced loads and leaves the PRODUCTS in LDS, pass 2 (lane t = row t) adds
// its LDS segment in order.  Two ways of getting at x:
//
//  * windowed tiles (fmt 1, 2; built once per matrix by mk_format.hip).  The tile's column set is covered by a few
//    contiguous windows of x, cut into chunks of 128 doubles; the chunks are staged in LDS with one coalesced
//    16-byte load per lane each (chunk c by wave c % 4; everything about a chunk is wave uniform and comes from
//    scalar loads) and pass 1 multiplies against ds_read_b64 at a per-nonzero uint16 slot.  No gather goes
//    through the texture-address path, which is what saturates first in a CSR product on this chip, and
//    a nonzero costs 2 (slot) + 8 (value) bytes of HBM traffic instead of 4 + 8; with a value dictionary (fmt 2,
//    matrices with <= 256 distinct values) one packed 4-byte word and a kernel of its own (below).  In fmt 1 the whole
//    input of the NEXT tile (matrix stream and windows)
//    is issued into registers as soon as this tile's registers have been consumed, so it lands while the row
//    sums are formed; two barriers per tile.
//  * gathers (fmt 0 and every tile the builder could not cover: scattered columns, > 2048 nonzeros): four
//    nonzeros per lane and step -- one 16-byte index load, two 16-byte value loads, four gathers through L1/L2.
//    Rows longer than the LDS tile are handled by looping over chunks with the running sum in a register.
//
// Epi interface:   double xin(double xj)            value actually multiplied (e.g. s*y[j])
//                  void   row(int64_t r, double s, double *acc)   consume the row result
// ---------------------------------------------------------------------------------------
// optional epilogue hook `void pre(int64_t r)`: loads that do not depend on the row sum (e.g. p[r] for <p, Ap>)
// are issued at the top of the tile instead of after the last barrier
template <class Epi, class = void>
struct MkHasPre : std::false_type {};
template <class Epi>
struct MkHasPre<Epi, std::void_t<decltype(std::declval<Epi &>().pre((int64_t)0))>> : std::true_type {};
// optional epilogue hook `void row_x(int64_t r, double s, double xr, double *acc)`: an epilogue whose own vector IS the
// product's input (CG: <p, Ap>) takes x[r] from the kernel -- in the pattern format it sits in the tile's LDS window --
// instead of loading it a second time through `pre`
template <class Epi, class = void>
struct MkHasRowX : std::false_type {};
template <class Epi>
struct MkHasRowX<Epi, std::void_t<decltype(std::declval<Epi &>().row_x((int64_t)0, 0.0, 0.0, (double *)nullptr))>>
    : std::true_type {};

typedef unsigned mk_u2 __attribute__((ext_vector_type(2)));
typedef unsigned mk_u4 __attribute__((ext_vector_type(4)));

// Staging buffer of the windowed path: product idx (relative to the 8-aligned start of the tile's stream) lives at
// [idx & 7][idx >> 3] of an 8 x 257 array -- lane l writes its i-th product to [i][l] (consecutive lanes, consecutive
// addresses: conflict-free ds_write_b64); column 256 holds zeros, so that pass 2 can mask a read by redirecting its
// ADDRESS there (one 32-bit select) instead of selecting 64-bit values: adding +0.0 never changes a running sum that
// started at +0.0 (it can never be -0.0).
__device__ __forceinline__ int mk_phys(int idx) { return (idx & 7) * MK_PROD_LD + (idx >> 3); }

// Tile id of position p of a launch's tile list, as a SCALAR: every lane computes the same value, but only an explicit
// readfirstlane lets the compiler keep it (and everything indexed by it: window descriptors, row-pointer ends) in
// scalar registers and fetch it with scalar loads -- as a vector value each tile started with a dependent vector
// load of its descriptor in front of all of its copies (measured: 18 % of the fmt 2 kernel at 512^3).
// Wave-uniform read-only data (tile lists, row-pointer ends, window descriptors) through the SCALAR cache: a load from
// the constant address space at a uniform address becomes an s_load.  (Pointers that arrive inside the by-value view
// struct are not scalarised by the compiler on its own: it emits a vector load + s_waitcnt vmcnt(0) + readfirstlane,
// i.e. a full memory round trip that also drains every copy already in flight.)
template <class T>
__device__ __forceinline__ T mk_sload(const T *p) {
    return *reinterpret_cast<const __attribute__((address_space(4))) T *>(reinterpret_cast<uintptr_t>(p));
}

__device__ __forceinline__ int64_t mk_tile_at(const MkCsrView &A, int64_t p) {
    const int t = A.tiles ? mk_sload(A.tiles + p) : (int)p;
    return (int64_t)__builtin_amdgcn_readfirstlane(t);
}

struct MkTileMeta {
    int p_lo, p_hi, my_lo;
};

// one tile through the gather path; `sum` is returned for row r0 + tid.  Starts and ends with the LDS free.
template <class Epi>
__device__ __forceinline__ double mk_tile_gather(const MkCsrView &A, const double *__restrict__ x, Epi &epi, double *prod,
                                                 int *sptr, const MkTileMeta &cur, double sum0 = 0.0) {
    constexpr int QUADS = MK_SPMV_TILE / (4 * MK_BLOCK);   // 2 groups of 4 nonzeros per lane per chunk
    const int tid = threadIdx.x;
    const int p_lo = cur.p_lo, p_hi = cur.p_hi, my_lo = cur.my_lo;
    int my_hi = p_hi;
    double sum = sum0;
    for (int base = p_lo & ~3; base < p_hi; base += MK_SPMV_TILE) {
        const int cnt = (p_hi - base < MK_SPMV_TILE) ? p_hi - base : MK_SPMV_TILE;
        // ---- pass 1: coalesced stream of the chunk, products into LDS.  Straight-line code: loads are clamped
        // instead of predicated and every lane stores its (possibly unused) products, so that the compiler keeps
        // all loads of the chunk in flight together.
        mk_i4 col[QUADS];
        mk_d2 val[QUADS][2];
#pragma unroll
        for (int k = 0; k < QUADS; ++k) {
            int j = 4 * (k * MK_BLOCK + tid);
            j = (j < cnt) ? j : ((cnt - 1) & ~3);
            col[k] = *reinterpret_cast<const mk_i4 *>(A.indices + base + j);
            val[k][0] = *reinterpret_cast<const mk_d2 *>(A.data + base + j);
            val[k][1] = *reinterpret_cast<const mk_d2 *>(A.data + base + j + 2);
            // slots past the chunk: keep the gathers in range
            col[k].y = (j + 1 < cnt) ? col[k].y : col[k].x;
            col[k].z = (j + 2 < cnt) ? col[k].z : col[k].x;
            col[k].w = (j + 3 < cnt) ? col[k].w : col[k].x;
        }
        mk_d2 xv[QUADS][2];
#pragma unroll
        for (int k = 0; k < QUADS; ++k) {
            xv[k][0].x = x[col[k].x];
            xv[k][0].y = x[col[k].y];
            xv[k][1].x = x[col[k].z];
            xv[k][1].y = x[col[k].w];
        }
#pragma unroll
        for (int k = 0; k < QUADS; ++k) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                mk_d2 pr;
                pr.x = val[k][h].x * epi.xin(xv[k][h].x);
                pr.y = val[k][h].y * epi.xin(xv[k][h].y);
                *reinterpret_cast<mk_d2 *>(prod + 4 * (k * MK_BLOCK + tid) + 2 * h) = pr;
            }
        }
        if (base == (p_lo & ~3)) {                       // first chunk of the tile: publish the row starts
            sptr[tid] = my_lo;
            if (tid == 0) sptr[MK_BLOCK] = p_hi;
        }
        __syncthreads();
        if (base == (p_lo & ~3)) my_hi = sptr[tid + 1];
        // ---- pass 2: one lane per row, left-to-right sum of its segment (clamped reads + selects)
        const int lo = ((my_lo > base) ? my_lo : base) - base;
        const int hi = ((my_hi < base + cnt) ? my_hi : base + cnt) - base;
        const int len = hi - lo;
        double t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int idx = lo + k;
            t[k] = prod[(idx < MK_SPMV_TILE && idx >= 0) ? idx : 0];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const double s2 = sum + t[k];
            sum = (k < len) ? s2 : sum;
        }
        for (int k = 8; k < len; ++k) sum += prod[lo + k];
        __syncthreads();
    }
    return sum;
}

template <int FMT, bool PROG, class Epi, int NACC>
__device__ __forceinline__ void mk_spmv_tiles(const MkCsrView &A, const double *__restrict__ x, Epi &epi,
                                              double *prod, double *xw, double (&acc)[NACC]) {
    const int tid = threadIdx.x;
    // Tile order.  Workgroup b is dispatched to XCD b % 8 (observed, MI355X_MICROARCH.md) and each XCD has its own
    // 4 MiB L2; the order only affects speed (and which rows a workgroup's partial sums cover).
    const int G = gridDim.x;
    const bool x8 = (G % 8 == 0);
    int64_t pos, stride, end;
    if (A.map == 1 && x8) {                                  // XCD b % 8 sweeps its own contiguous eighth
        const int64_t chunk = (A.ntl + 7) / 8, c0 = (int64_t)(blockIdx.x % 8) * chunk;
        pos = c0 + blockIdx.x / 8;
        stride = G / 8;
        end = (c0 + chunk < A.ntl) ? c0 + chunk : A.ntl;
    } else if (A.map == 2 && x8) {                           // every step of the grid: eight XCD-contiguous blocks
        pos = (int64_t)(blockIdx.x % 8) * (G / 8) + blockIdx.x / 8;
        stride = G;
        end = A.ntl;
    } else {
        pos = blockIdx.x;
        stride = G;
        end = A.ntl;
    }
    __shared__ int sptr[MK_BLOCK + 1];
    // Row pointers of a tile (one load per lane: a row's end is its neighbour's start and travels through LDS);
    // fetched ahead so that their latency is not on the critical path.
    auto load_meta = [&](int64_t p, MkTileMeta &m) {         // p: position in the tile list of this launch
        m.p_lo = m.p_hi = m.my_lo = 0;
        if (p < end) {
            const int64_t tile = mk_tile_at(A, p);
            const int64_t r0 = tile * MK_ROWS_PER_TILE;
            const int64_t rend = (r0 + MK_ROWS_PER_TILE < A.nrows) ? r0 + MK_ROWS_PER_TILE : A.nrows;
            const int64_t r = r0 + tid;
            m.p_lo = mk_sload(A.indptr + r0);
            m.p_hi = mk_sload(A.indptr + rend);
            m.my_lo = A.indptr[(r < rend) ? r : rend];      // rows past the end start (and end) at p_hi
        }
    };
    if constexpr (FMT == 0) {
        MkTileMeta cur, nxt;
        load_meta(pos, cur);
        for (; pos < end; pos += stride) {
            const int64_t tile = mk_tile_at(A, pos);
            const int64_t r0 = tile * MK_ROWS_PER_TILE;
            const int64_t rend = (r0 + MK_ROWS_PER_TILE < A.nrows) ? r0 + MK_ROWS_PER_TILE : A.nrows;
            const int64_t r = r0 + tid;
            if constexpr (MkHasPre<Epi>::value) {
                if (r < rend) epi.pre(r);
            }
            load_meta(pos + stride, nxt);                    // next tile's row pointers go in flight now
            const double sum0 = (A.sum_in && r < rend) ? A.sum_in[r] : 0.0;   // (column-blocked product: carried sums)
            double sum = mk_tile_gather(A, x, epi, prod, sptr, cur, sum0);
            if constexpr (PROG) {                            // composed operators only (separate instantiation)
                if (r < rend) sum = mk_rowprog(A, sum, x, r, epi);
            }
            if (r < rend) epi.row(r, sum, acc);
            cur = nxt;
        }
    } else if constexpr (FMT == 3) {
        // ---- resident tiles with column phases: for matrices whose x is too long for an XCD's L2 and whose columns no
        // window covers (BASELINE config 3).  The tile's (column, value) stream goes to LDS once (global_load_lds);
        // then lane t walks row t with a cursor, left to right as everywhere, but in rt_k PHASES: phase k takes the
        // row's entries whose column lies in block k (columns ascend within a row, so a phase is a contiguous run).
        // All workgroups of the grid are resident and start their tiles together, so at any moment the whole chip
        // gathers from ONE slice of x that an L2 holds, instead of pulling a 64-byte sector through the fabric per
        // nonzero (rocprof on the gather path: 371 MB of fabric reads for 80 MB of algorithmic bytes; the raw cost of
        // 5 M scattered 8-byte gathers from 8 MB alone is 40 us, tools/ubench/spmv_cb.hip).  No barrier and no
        // global load other than the gathers inside the phases.
        const int lane = tid & 63;
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int cap = A.rt_cap;
        double *lv = prod;                                   // [cap] values, then [cap] columns
        int *lc = reinterpret_cast<int *>(prod + cap);
        for (; pos < end; pos += stride) {
            const int64_t tile = mk_tile_at(A, pos);
            const int64_t r0 = tile * MK_ROWS_PER_TILE;
            const int64_t rend = (r0 + MK_ROWS_PER_TILE < A.nrows) ? r0 + MK_ROWS_PER_TILE : A.nrows;
            const int64_t r = r0 + tid;
            const int p_lo = mk_sload(A.indptr + r0), p_hi = mk_sload(A.indptr + rend);
            const int base = p_lo & ~3, cnt = p_hi - base;   // cnt <= cap (builder)
            const int last = (cnt > 0) ? ((cnt - 1) & ~3) : 0;
            for (int c0 = wv * 256; c0 < cnt; c0 += 4 * 256) {                  // 256 columns per wave-level copy
                int j = c0 + 4 * lane;
                j = j < last ? j : last;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(A.indices + base + j),
                                                 (__attribute__((address_space(3))) void *)(lc + c0), 16, 0, 0);
            }
            const int lastv = (cnt > 0) ? ((cnt - 1) & ~1) : 0;
            for (int c0 = wv * 128; c0 < cnt; c0 += 4 * 128) {                  // 128 values per wave-level copy
                int j = c0 + 2 * lane;
                j = j < lastv ? j : lastv;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(A.data + base + j),
                                                 (__attribute__((address_space(3))) void *)(lv + c0), 16, 0, 0);
            }
            int cur = 0, fin = 0;
            if (r < rend) {
                cur = A.indptr[r] - base;
                fin = A.indptr[r + 1] - base;
            }
            if constexpr (MkHasPre<Epi>::value) {
                if (r < rend) epi.pre(r);
            }
            double sum = (A.sum_in && r < rend) ? A.sum_in[r] : 0.0;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            for (int k = 0; k < A.rt_k; ++k) {
                const int c1 = (k + 1 < A.rt_k) ? (k + 1) * A.rt_w : 0x7fffffff;
                for (;;) {
                    int ca = 0x7fffffff;
                    if (cur < fin) ca = lc[cur];
                    const bool oa = ca < c1;
                    if (oa) {
                        const double xa = x[ca];
                        sum += lv[cur] * epi.xin(xa);
                        cur += 1;
                    }
                    if (!__any(oa)) break;                   // (wave level: no lane of this wave has more in phase k)
                }
            }
            if constexpr (PROG) {
                if (r < rend) sum = mk_rowprog(A, sum, x, r, epi);
            }
            if (r < rend) epi.row(r, sum, acc);
            __syncthreads();                                 // the next tile's copies overwrite this LDS
        }
    } else if constexpr (FMT == 2 || FMT == 4) {
        constexpr bool PAT = (FMT == 4);
        // ---- windowed tiles with a value dictionary: ROW PHASE ONLY.  One 32-bit word per nonzero {slot | code};
        // the words and the x windows of a tile go straight to LDS with global_load_lds (no VGPR round trip, no
        // per-nonzero staging work); after one barrier lane t walks row t left to right: word, x and value from LDS
        // (consecutive rows read consecutive words / x entries: conflict free for the usual odd row lengths).
        // Measured against the product-staging design of fmt 1 with the codes: 512^3 1.50 -> 1.32 ms, 2-D n = 1e6
        // 13.1 -> 9.1 us (tools/ubench/spmv_win2.hip, w3 vs w7).
        // fmt 4 (PAT) goes one step further: a row is described by ONE BYTE, the number of its pattern -- the sequence
        // of its words relative to the lane, {slot - t, code} -- and the pattern table (<= 8 KB) sits in LDS for the
        // whole kernel.  The per-nonzero stream and the row pointers are not read at all: what a tile ingests is its
        // x windows and 256 bytes.
        const int lane = tid & 63;
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        // fmt 2: a tile's packed words behind its windows.  fmt 4: the pattern table, which lives as long as the kernel
        // and therefore sits behind everything the gather path of a tile without windows may overwrite
        // (unless every tile has windows: then the gather path never runs, A.allwin)
        const int wtop = 128 * A.wchunks + 2;
        uint32_t *spk = reinterpret_cast<uint32_t *>(xw + ((PAT && !A.allwin && wtop < MK_PROD_LDS) ? MK_PROD_LDS : wtop));
        __shared__ double sdict[PAT ? 1 : 256];              // (fmt 4 keeps the values in its pattern table)
        if constexpr (!PAT) sdict[tid] = (tid < A.ndict) ? A.dict[tid] : 0.0;   // (read after a barrier below)
        [[maybe_unused]] __shared__ int splen[PAT ? 256 : 1];
        // fmt 4: behind the windows (and whatever the gather path may overwrite) 256 zeros, then the pattern table in
        // the form the row phase consumes with the fewest instructions -- per entry {byte offset of its x value
        // relative to the lane's own cell, the VALUE itself}: one 16-byte LDS read, one add, one 8-byte LDS read, one
        // multiply, one add.  Entries past a pattern's end point at the lane's zero cell with value +0.0: their
        // product is +-0.0 and leaves the running sum (never -0.0, it started at +0.0) unchanged, so nothing is masked.
        struct PatEntry {
            int off, pad;
            double val;
        };
        [[maybe_unused]] PatEntry *ftab = nullptr;
        [[maybe_unused]] int zoff = 0;
        if constexpr (PAT) {
            double *zeros = reinterpret_cast<double *>(spk);
            ftab = reinterpret_cast<PatEntry *>(zeros + MK_BLOCK);
            zoff = (int)((zeros - xw) * (int)sizeof(double));
            zeros[tid] = 0.0;
            for (int e = tid; e < A.npat * A.pmax; e += MK_BLOCK) {
                const int pnum = e / A.pmax, k = e - pnum * A.pmax;
                const uint32_t w = A.pat[e];
                PatEntry en;
                en.pad = 0;
                if (k < (int)A.plen[pnum]) {
                    en.off = 8 * (int)(short)(w & 0xffffu);
                    en.val = A.dict[w >> 16];
                } else {
                    en.off = zoff;
                    en.val = 0.0;
                }
                ftab[e] = en;
            }
            splen[tid] = (tid < A.npat) ? ((int)A.plen[tid] | ((int)A.plen[256 + tid] << 8)) : (255 << 8);
        }
        const double d0 = A.dict[0], d1 = A.dict[A.ndict > 1 ? 1 : 0];
        const bool two = A.ndict <= 2;                       // value picked in registers instead of read from LDS
        // this wave's window descriptor of a tile (scalar loads, issued one tile ahead like the row pointers)
        struct Desc {
            mk_i4 g;
            unsigned nvw;
        };
        auto load_desc = [&](int64_t p, Desc &d) {
            d.g = mk_i4{0, 0, 0, 0};
            d.nvw = 0;
            if (p < end) {
                const int64_t t = mk_tile_at(A, p);
                d.g = mk_sload(reinterpret_cast<const mk_i4 *>(A.wg + (t * 4 + wv) * 4));
                d.nvw = mk_sload(A.wn + t * 4 + wv);
            }
        };
        MkTileMeta cur, nxt;
        Desc dcur, dnxt;
        if constexpr (!PAT) load_meta(pos, cur);
        load_desc(pos, dcur);
        for (; pos < end; pos += stride) {
            const int64_t tile = mk_tile_at(A, pos);
            const int64_t r0 = tile * MK_ROWS_PER_TILE;
            const int64_t rend = (r0 + MK_ROWS_PER_TILE < A.nrows) ? r0 + MK_ROWS_PER_TILE : A.nrows;
            const int64_t r = r0 + tid;
            constexpr bool ROWX = PAT && !PROG && MkHasRowX<Epi>::value;
            if constexpr (MkHasPre<Epi>::value && !ROWX) {
                if (r < rend) epi.pre(r);
            }
            const mk_i4 g = dcur.g;
            double sum = 0.0;
            [[maybe_unused]] double xr_cur = 0.0;
            if (g.x & 1) {
                const unsigned nvw = dcur.nvw;
                const int gs[4] = {g.x & ~1, g.y, g.z, g.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int hc = (int)((nvw >> (8 * i)) & 0xffu);
                    if (hc > 0) {
                        const int l2 = (lane < hc) ? lane : hc - 1;
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(x + gs[i] + 2 * l2),
                                                         (__attribute__((address_space(3))) void *)(xw + (wv + 4 * i) * 128),
                                                         16, 0, 0);
                    }
                }
                int lo = 0, len = 0;
                [[maybe_unused]] int kdiag = 255;
                if constexpr (PAT) {
                    const unsigned id = (r < rend) ? (unsigned)A.pid[r] : 0u;   // one byte per row
                    load_desc(pos + stride, dnxt);           // next tile's descriptor goes in flight
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    lo = (int)id * A.pmax;
                    len = (r < rend) ? (splen[id] & 0xff) : 0;
                    kdiag = splen[id] >> 8;
                    const char *cell = reinterpret_cast<const char *>(xw + tid);    // this lane's own cell
                    const PatEntry *pe = ftab + lo;
                    PatEntry en[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) en[k] = pe[k];
                    double xk[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) xk[k] = epi.xin(*reinterpret_cast<const double *>(cell + en[k].off));
#pragma unroll
                    for (int k = 0; k < 8; ++k) sum += en[k].val * xk[k];
                    for (int k = 8; k < len; ++k)
                        sum += pe[k].val * epi.xin(*reinterpret_cast<const double *>(cell + pe[k].off));
                    if constexpr (ROWX) {                    // x[r] for the epilogue: the diagonal entry's cell
                        if (kdiag < len) xr_cur = epi.xin(*reinterpret_cast<const double *>(cell + pe[kdiag].off));
                        else if (r < rend) xr_cur = epi.xin(x[r]);
                    }
                    __syncthreads();                         // the next tile's copies overwrite this LDS
                    if constexpr (PROG) {
                        if (r < rend) sum = mk_rowprog(A, sum, x, r, epi);
                    }
                    if constexpr (ROWX) {
                        if (r < rend) epi.row_x(r, sum, xr_cur, acc);
                    } else {
                        if (r < rend) epi.row(r, sum, acc);
                    }
                    cur = nxt;
                    dcur = dnxt;
                    continue;
                } else {
                    const int p_lo = cur.p_lo, p_hi = cur.p_hi, my_lo = cur.my_lo;
                    const int base = p_lo & ~3, cnt = p_hi - base;         // 0 < cnt <= MK_SPMV_TILE + 3 (builder)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {                            // 256 words per wave-level copy
                        const int c0 = (wv + 4 * c) * 256;
                        if (c0 < cnt)
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(A.pk + base + c0 + 4 * lane),
                                                             (__attribute__((address_space(3))) void *)(spk + c0), 16, 0, 0);
                    }
                    load_meta(pos + stride, nxt);            // next tile's row pointers and descriptor go in flight
                    load_desc(pos + stride, dnxt);
                    sptr[tid] = my_lo;
                    if (tid == 0) sptr[MK_BLOCK] = p_hi;
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    const int my_hi = sptr[tid + 1];
                    lo = my_lo - base;
                    len = my_hi - my_lo;
                }
                if constexpr (!PAT) {
                auto slot_of = [&](unsigned w, int) -> unsigned { return w & 0xffffu; };
                unsigned wk[8];
                double xk[8], vk[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) wk[k] = spk[lo + k];
                if (two) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        xk[k] = epi.xin(xw[slot_of(wk[k], k)]);
                        vk[k] = (wk[k] >> 16) ? d1 : d0;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        xk[k] = epi.xin(xw[slot_of(wk[k], k)]);
                        vk[k] = sdict[wk[k] >> 16];
                    }
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const double t = vk[k] * xk[k];
                    sum += (k < len) ? t : 0.0;              // (+0.0 never changes a running sum that started at +0.0)
                }
                for (int k = 8; k < len; ++k) {
                    const unsigned w = spk[lo + k];
                    sum += sdict[w >> 16] * epi.xin(xw[slot_of(w, k)]);
                }
                __syncthreads();                             // the next tile's copies overwrite this LDS
                }
            } else {
                if constexpr (ROWX) {
                    if (r < rend) xr_cur = epi.xin(x[r]);
                }
                if constexpr (PAT) load_meta(pos, cur);      // (tiles without windows are rare: their row pointers now)
                else load_meta(pos + stride, nxt);
                load_desc(pos + stride, dnxt);
                sum = mk_tile_gather(A, x, epi, prod, sptr, cur);
            }
            if constexpr (PROG) {
                if (r < rend) sum = mk_rowprog(A, sum, x, r, epi);
            }
            if constexpr (ROWX) {
                if (r < rend) epi.row_x(r, sum, xr_cur, acc);
            } else {
                if (r < rend) epi.row(r, sum, acc);
            }
            cur = nxt;
            dcur = dnxt;
        }
    } else {
        const int lane = tid & 63;
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        struct WRegs {
            mk_u4 s;
            mk_d2 val[4];
            mk_d2 w[4];
            unsigned nvw;
            bool valid;
        };
        // everything a windowed tile needs from memory, into registers (no use of the values here)
        auto issue = [&](int64_t p, const MkTileMeta &m, WRegs &R) {
            R.valid = false;
            R.nvw = 0;
            if (p >= end) return;
            const int64_t tile = mk_tile_at(A, p);
            const mk_i4 g = mk_sload(reinterpret_cast<const mk_i4 *>(A.wg + (tile * 4 + wv) * 4));
            const unsigned nvw = mk_sload(A.wn + tile * 4 + wv);
            if (!(g.x & 1)) return;                          // the builder could not cover this tile: gather path
            R.valid = true;
            R.nvw = nvw;
            const int gs[4] = {g.x & ~1, g.y, g.z, g.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int hc = (int)((R.nvw >> (8 * i)) & 0xffu);
                if (hc > 0) {
                    const int l2 = (lane < hc) ? lane : hc - 1;
                    R.w[i] = *reinterpret_cast<const mk_d2 *>(x + gs[i] + 2 * l2);
                }
            }
            const int base = m.p_lo & ~7, cnt = m.p_hi - base;           // 0 < cnt <= MK_SPMV_TILE (builder)
            int j = 8 * tid;
            j = (j < cnt) ? j : ((cnt - 1) & ~7);
            R.s = *reinterpret_cast<const mk_u4 *>(A.slots + base + j);
#pragma unroll
            for (int h = 0; h < 4; ++h) R.val[h] = *reinterpret_cast<const mk_d2 *>(A.data + base + j + 2 * h);
        };
        MkTileMeta cur, nxt, nx2;
        WRegs R;
        load_meta(pos, cur);
        load_meta(pos + stride, nxt);
        issue(pos, cur, R);
        bool lds_busy = false;                               // products of a windowed tile may still be read by slower waves
        bool zero_ok = false;                                // the zero column of the staging buffer is in place
        for (; pos < end; pos += stride) {
            const int64_t tile = mk_tile_at(A, pos);
            const int64_t r0 = tile * MK_ROWS_PER_TILE;
            const int64_t rend = (r0 + MK_ROWS_PER_TILE < A.nrows) ? r0 + MK_ROWS_PER_TILE : A.nrows;
            const int64_t r = r0 + tid;
            if constexpr (MkHasPre<Epi>::value) {
                if (r < rend) epi.pre(r);
            }
            double sum = 0.0;
            if (R.valid) {
                const int p_lo = cur.p_lo, p_hi = cur.p_hi, my_lo = cur.my_lo;
                const int base = p_lo & ~7;
                // windows -> LDS (the epilogue's on-the-fly scaling of x is applied here, once per entry)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int hc = (int)((R.nvw >> (8 * i)) & 0xffu);
                    if (hc > 0) {
                        mk_d2 v;
                        v.x = epi.xin(R.w[i].x);
                        v.y = epi.xin(R.w[i].y);
                        *reinterpret_cast<mk_d2 *>(xw + (wv + 4 * i) * 128 + 2 * lane) = v;
                    }
                }
                sptr[tid] = my_lo;
                if (tid == 0) sptr[MK_BLOCK] = p_hi;
                __syncthreads();
                const int my_hi = sptr[tid + 1];
                // ---- pass 1: products of this lane's 8 nonzeros against the LDS windows
                const unsigned sw[4] = {R.s.x, R.s.y, R.s.z, R.s.w};
                double pr[8];
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const double x0 = xw[sw[h] & 0xffffu], x1 = xw[sw[h] >> 16];
                    pr[2 * h] = R.val[h].x * x0;
                    pr[2 * h + 1] = R.val[h].y * x1;
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) prod[i * MK_PROD_LD + tid] = pr[i];
                if (!zero_ok) {                              // (a gather tile overwrote the zero column)
                    if (tid < 8) prod[tid * MK_PROD_LD + MK_BLOCK] = 0.0;
                    zero_ok = true;
                }
                // the registers are free: the next tile's input goes in flight and lands during pass 2
                load_meta(pos + 2 * stride, nx2);
                issue(pos + stride, nxt, R);
                __syncthreads();
                // ---- pass 2: one lane per row, left-to-right sum of its segment.  Entry lo + k sits at
                // [(a + k) & 7][b + carry]: one of two precomputed bases plus a compile-time offset.
                const int lo = my_lo - base, len = my_hi - my_lo;
                const int a = lo & 7;
                const int adA = a * MK_PROD_LD + (lo >> 3), adB = adA - (8 * MK_PROD_LD - 1);
                double t[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    int ad = (a + k >= 8) ? adB : adA;
                    ad = (k < len) ? ad : MK_BLOCK;          // past the row: the zero column
                    t[k] = prod[ad + k * MK_PROD_LD];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) sum += t[k];
                for (int k = 8; k < len; ++k) sum += prod[mk_phys(lo + k)];
                lds_busy = true;
            } else {
                if (lds_busy) __syncthreads();               // slower waves may still read the previous tile's products
                load_meta(pos + 2 * stride, nx2);
                sum = mk_tile_gather(A, x, epi, prod, sptr, cur);
                issue(pos + stride, nxt, R);
                lds_busy = false;
                zero_ok = false;
            }
            if constexpr (PROG) {
                if (r < rend) sum = mk_rowprog(A, sum, x, r, epi);
            }
            if (r < rend) epi.row(r, sum, acc);
            cur = nxt;
            nxt = nx2;
        }
    }
}

// Gate: a test on a global sum that must be settled before the product may start (e.g. the loop
// condition on ||r||).  Every workgroup evaluates it identically from the previous kernel's partial
// sums; `open` returns whether to run the product and may request a halt through *stop.
struct MkNoGate {
    __device__ bool open(double *, bool, bool *) { return true; }
};

template <class Epi, class Gate, bool PROG, int FMT>
__global__ __launch_bounds__(MK_BLOCK, (FMT == 0 || FMT == 3) ? 8 : (FMT == 4 ? 7 : 4)) void mk_spmv_kernel(MkCsrView A, const double *__restrict__ x, Epi epi,
                                                           Gate gate, MkHalt halt, double *__restrict__ partials) {
    // fmt 0 / 1: products [MK_PROD_LDS doubles], then the windows.  fmt 2 has no product staging: its windows and
    // packed words share the space the gather path of uncovered tiles uses for products (never live together)
    extern __shared__ __attribute__((aligned(16))) double mk_smem[];
    double *prod = mk_smem;
    double *xw = (FMT == 2 || FMT == 4) ? mk_smem : mk_smem + MK_PROD_LDS;
    __shared__ double s4[4];
    const bool halted = halt.in();
    const bool lead = (blockIdx.x == 0 && threadIdx.x == 0);
    if (halted) {
        if (lead) halt.out(true);
        return;
    }
    bool stop = false;
    const bool go = gate.open(s4, lead && A.part != 2, &stop);     // part 2 repeats the decision, not the writes
    if (lead) halt.out(stop);
    if constexpr (FMT == 0 && !PROG) {
        if (lead && A.cb_mode == 1) *A.cb_go = go ? 1 : 0;          // (zeroed by the host before the launch)
    }
    if (!go) return;
    epi.prologue(s4);
    double acc[Epi::NACC > 0 ? Epi::NACC : 1];
#pragma unroll
    for (int d = 0; d < (Epi::NACC > 0 ? Epi::NACC : 1); ++d) acc[d] = 0.0;
    if constexpr (FMT == 0 && !PROG) {
        if (A.cb_mode == 1) {                                       // matrix-free operator: the callback's input
            for (int64_t j = (int64_t)blockIdx.x * MK_BLOCK + threadIdx.x; j < A.xlen; j += (int64_t)gridDim.x * MK_BLOCK)
                A.vin[j] = epi.xin(x[j]);
            return;
        }
        if (A.cb_mode == 2) {                                       // ... and its result through the row epilogue
            for (int64_t tile = blockIdx.x; tile < A.ntiles; tile += gridDim.x) {
                const int64_t r = tile * MK_ROWS_PER_TILE + threadIdx.x;
                if (r < A.nrows) {
                    if constexpr (MkHasPre<Epi>::value) epi.pre(r);
                    epi.row(r, A.y_ext[r], acc);
                }
            }
        } else {
            mk_spmv_tiles<FMT, PROG>(A, x, epi, prod, xw, acc);
        }
    } else {
        mk_spmv_tiles<FMT, PROG>(A, x, epi, prod, xw, acc);
    }
#pragma unroll
    for (int d = 0; d < Epi::NACC; ++d) {
        const double tot = mk_block_sum(acc[d], s4);
        if (threadIdx.x == 0) partials[(Epi::SLOT0 + d) * MK_MAXP + A.poff + blockIdx.x] = tot;
    }
    if (A.part != 1) halt.template clear_tail<Epi::NACC, Epi::SLOT0>(partials, A.poff + (int)gridDim.x);
}

// Launch the instantiation that matches the operator: plain matrices never pay for the row program, matrices
// without windowed tiles never pay for the window code.
template <class Epi, class Gate, bool PROG>
static inline void mk_spmv_launch_fmt(const MkCsrView &v, int grid, hipStream_t st, const double *x, const Epi &epi,
                                      const Gate &gate, MkHalt halt, double *partials) {
    size_t lds = sizeof(double) * (size_t)(MK_PROD_LDS + (v.fmt == 1 ? 128 * v.wchunks + 2 : 0));
    if (v.fmt == 2) {                                        // windows + packed words, or the gather path's products
        const size_t w = sizeof(double) * (size_t)(128 * v.wchunks + 2) + sizeof(uint32_t) * (MK_SPMV_TILE + 16);
        lds = w > lds ? w : lds;
    }
    if (v.fmt == 4) {                                        // windows + pattern table, or the gather path's products
        size_t wtop = (size_t)(128 * v.wchunks + 2);
        if (!v.allwin && wtop < (size_t)MK_PROD_LDS) wtop = (size_t)MK_PROD_LDS;
        lds = sizeof(double) * (wtop + MK_BLOCK) + 16 * (size_t)(v.npat * v.pmax + 1);   // windows, zeros, table
        hipLaunchKernelGGL((mk_spmv_kernel<Epi, Gate, PROG, 4>), dim3(grid), dim3(MK_BLOCK), lds, st, v, x, epi, gate,
                           halt, partials);
    } else if (v.fmt == 3) {                                 // the tile's values and columns
        lds = (size_t)v.rt_cap * 12;
        hipLaunchKernelGGL((mk_spmv_kernel<Epi, Gate, PROG, 3>), dim3(grid), dim3(MK_BLOCK), lds, st, v, x, epi, gate,
                           halt, partials);
    } else if (v.fmt == 2)
        hipLaunchKernelGGL((mk_spmv_kernel<Epi, Gate, PROG, 2>), dim3(grid), dim3(MK_BLOCK), lds, st, v, x, epi, gate,
                           halt, partials);
    else if (v.fmt == 1)
        hipLaunchKernelGGL((mk_spmv_kernel<Epi, Gate, PROG, 1>), dim3(grid), dim3(MK_BLOCK), lds, st, v, x, epi, gate,
                           halt, partials);
    else
        hipLaunchKernelGGL((mk_spmv_kernel<Epi, Gate, PROG, 0>), dim3(grid), dim3(MK_BLOCK), lds, st, v, x, epi, gate,
                           halt, partials);
}

template <class Epi, class Gate>
static inline void mk_spmv_launch_view(const MkCsrView &v, int grid, hipStream_t st, const double *x, const Epi &epi,
                                       const Gate &gate, MkHalt halt, double *partials) {
    if (v.nops > 0) mk_spmv_launch_fmt<Epi, Gate, true>(v, grid, st, x, epi, gate, halt, partials);
    else mk_spmv_launch_fmt<Epi, Gate, false>(v, grid, st, x, epi, gate, halt, partials);
}

// Column-blocked product (mk_format.hip): the matrix is stored as K column blocks, each a CSR matrix over all rows.
// Block k's launch continues every row sum where block k-1 left it (columns are sorted, so this IS the left-to-right
// sum over the whole row: same bits), gathering only from its own slice of x -- which fits the XCD's L2, whereas a
// random gather over an x larger than the L2 pulls a 128-byte line through the fabric per 8 useful bytes.  All
// launches but the last store the running sums; the last one runs the real epilogue (and the row program).
template <class Epi>
struct MkPartialOf {
    static constexpr int NACC = 0, SLOT0 = 0;
    Epi e;
    double *ysum;
    __device__ void prologue(double *s4) { e.prologue(s4); }
    __device__ double xin(double v) const { return e.xin(v); }
    __device__ void row(int64_t r, double s, double *) { ysum[r] = s; }
};

// Pair operators (mk_csr_create_sum / _product): the second launch's epilogue wrappers.  (The wrapped epilogue's
// optional `pre` hook is forwarded through a base class so that MkHasPre sees it exactly when the epilogue has one.)
template <class Epi, bool = MkHasPre<Epi>::value>
struct MkWrapBase {
    Epi e;
};
template <class Epi>
struct MkWrapBase<Epi, true> {
    Epi e;
    __device__ void pre(int64_t r) { e.pre(r); }
};
template <class Epi, int SIGN>
struct MkAddOf : MkWrapBase<Epi> {   // row sum = t1[r] + s  (or - s): `(A*x) + (B*x)`, linop.py:375-398, :403-426
    static constexpr int NACC = Epi::NACC, SLOT0 = Epi::SLOT0;
    const double *t1;
    __device__ void prologue(double *s4) { this->e.prologue(s4); }
    __device__ double xin(double v) const { return this->e.xin(v); }
    __device__ void row(int64_t r, double s, double *acc) { this->e.row(r, SIGN > 0 ? t1[r] + s : t1[r] - s, acc); }
};
template <class Epi>
struct MkNoXin : MkWrapBase<Epi> {   // the outer product of `A*(B*x)`: its input B*x was formed from xin(x) already
    static constexpr int NACC = Epi::NACC, SLOT0 = Epi::SLOT0;
    __device__ void prologue(double *s4) { this->e.prologue(s4); }
    __device__ double xin(double v) const { return v; }
    __device__ void row(int64_t r, double s, double *acc) { this->e.row(r, s, acc); }
};

// `next` yields the MkHalt of each launch (one per kernel: the halting protocol alternates the flag word)
template <class Epi, class Gate, class HaltSrc>
static inline void mk_spmv_launch_blocks(const mk_csr *A, int grid, hipStream_t st, const double *x, const Epi &epi,
                                         const Gate &gate, HaltSrc &&next, double *partials) {
    if (A->comp_kind) {
        // Two device matrices as one operator: first product into the temporary (gate with its side effects), second
        // product with the combining wrapper around the real epilogue (gate's decision repeated).
        const mk_csr *P = A->comp_a, *Q = A->comp_b;
        if (A->comp_kind == 3) {
            MkCsrView v1 = mk_view(Q);
            v1.part = 1;
            mk_spmv_launch_view(v1, mk_grid_spmv_for(Q), st, x, MkPartialOf<Epi>{epi, A->d_comp_tmp}, gate, next(), partials);
            MkCsrView v2 = mk_view(P);
            v2.part = 2;
            mk_spmv_launch_view(v2, grid, st, A->d_comp_tmp, MkNoXin<Epi>{{epi}}, gate, next(), partials);
        } else {
            MkCsrView v1 = mk_view(P);
            v1.part = 1;
            mk_spmv_launch_view(v1, mk_grid_spmv_for(P), st, x, MkPartialOf<Epi>{epi, A->d_comp_tmp}, gate, next(), partials);
            MkCsrView v2 = mk_view(Q);
            v2.part = 2;
            if (A->comp_kind == 1)
                mk_spmv_launch_view(v2, grid, st, x, MkAddOf<Epi, 1>{{epi}, A->d_comp_tmp}, gate, next(), partials);
            else
                mk_spmv_launch_view(v2, grid, st, x, MkAddOf<Epi, -1>{{epi}, A->d_comp_tmp}, gate, next(), partials);
        }
        return;
    }
    if (A->host_fn) {
        // Matrix-free operator.  Launch 1 evaluates the gate (with its side effects) and materialises the input
        // vector; the host calls the operator if the gate let the product through; launch 2 repeats the gate's
        // decision (no side effects) and feeds the result to the real epilogue.
        MkCsrView v = mk_view(A);
        v.cb_go = A->d_cb_go;
        v.vin = A->d_cb_in;
        v.y_ext = A->d_cb_out;
        v.xlen = A->ncols;
        v.cb_mode = 1;
        v.part = 1;
        hipMemsetAsync(A->d_cb_go, 0, sizeof(int), st);
        mk_spmv_launch_view(v, grid, st, x, MkPartialOf<Epi>{epi, nullptr}, gate, next(), partials);
        const int rc = mk_host_product(A, st);
        if (rc != MK_OK && mk_ctx().pending_rc == MK_OK) mk_ctx().pending_rc = rc;
        v.cb_mode = 2;
        v.part = 2;
        mk_spmv_launch_view(v, grid, st, x, epi, gate, next(), partials);
        return;
    }
    const MkPlan *P = mk_csr_plan(A);
    const size_t K = (P && A->ex.mode < 0) ? P->cblocks.size() : 0;
    if (K < 2) {
        mk_spmv_launch_view(mk_view(A), grid, st, x, epi, gate, next(), partials);
        return;
    }
    for (size_t k = 0; k < K; ++k) {
        MkCsrView v = mk_view(A);
        const mk_csr *B = P->cblocks[k];
        v.indptr = B->d_indptr;
        v.indices = B->d_indices;
        v.data = B->d_data;
        v.sum_in = k ? P->d_cbsum : nullptr;
        v.part = k ? 2 : 1;                                  // the gate's side effects happen in the first launch only
        if (k + 1 < K) {
            v.nops = 0;
            mk_spmv_launch_view(v, grid, st, x, MkPartialOf<Epi>{epi, P->d_cbsum}, gate, next(), partials);
        } else {
            mk_spmv_launch_view(v, grid, st, x, epi, gate, next(), partials);
        }
    }
}

template <class Epi, class Gate>
static inline void mk_spmv_launch(const mk_csr *A, int grid, hipStream_t st, const double *x, const Epi &epi,
                                  const Gate &gate, MkHalt halt, double *partials) {
    // (one halt word for all launches: only for callers whose flags never change)
    mk_spmv_launch_blocks(A, grid, st, x, epi, gate, [&] { return halt; }, partials);
}

// ---------------------------------------------------------------------------------------
// Fused BLAS-1 pass.
// Op interface:  static constexpr int NACC, SLOT0;
//                bool prologue(double *s4, bool lead)     -> returns the new halt condition
//                                                             (lead == block 0 thread 0 may write scalars)
//                bool skip() const                         -> after prologue: do no vector work
//                void pair(int64_t i, double *acc)         -> elements i, i+1 (16-byte access)
//                void one(int64_t i, double *acc)          -> tail element
// Lane g of the grid handles pairs g, g+S, g+2S, ...  (S = total lanes): every wave
// instruction touches 1 KiB of consecutive memory.
// ---------------------------------------------------------------------------------------
// Ops may additionally provide the split form  struct Regs; load2(i, Regs&); apply2(Regs&, acc); store2(i, Regs&):
// the kernel then issues the loads of each lane's FIRST pair before it waits for anything else (halt word, partial
// sums, the prologue's barriers), which takes about one memory latency off every launch -- what matters for the
// cache-resident sizes where a whole kernel lasts 5-10 us.
template <class Op, class = void>
struct MkIsSplit : std::false_type {};
template <class Op>
struct MkIsSplit<Op, std::void_t<typename Op::Regs>> : std::true_type {};
template <class Op, bool = MkIsSplit<Op>::value>
struct MkRegsOf {
    struct type {};
};
template <class Op>
struct MkRegsOf<Op, true> {
    using type = typename Op::Regs;
};

// optional op hook `void early()`: issue (not consume) the loads of the prologue -- partial sums into MkTotalRegs,
// scalars into members -- so that they travel together with the halt word instead of after it
template <class Op, class = void>
struct MkHasEarly : std::false_type {};
template <class Op>
struct MkHasEarly<Op, std::void_t<decltype(std::declval<Op &>().early())>> : std::true_type {};

template <class Op>
__global__ __launch_bounds__(MK_BLOCK) void mk_stream_kernel(Op op, int64_t n, MkHalt halt,
                                                             double *__restrict__ partials) {
    __shared__ double s4[4];
    constexpr bool split = MkIsSplit<Op>::value;
    const int64_t S = (int64_t)gridDim.x * MK_BLOCK;
    const int64_t g = (int64_t)blockIdx.x * MK_BLOCK + threadIdx.x;
    const int64_t npair = n >> 1;
    [[maybe_unused]] typename MkRegsOf<Op>::type first;
    [[maybe_unused]] const bool has_first = g < npair;
    if constexpr (split) {
        if (has_first) op.load2(2 * g, first);       // in flight while the prologue runs (barriers pin it here)
    }
    const int hword = halt.flags[halt.parity];
    if constexpr (MkHasEarly<Op>::value) op.early();
    const bool halted = hword != 0;
    const bool lead = (blockIdx.x == 0 && threadIdx.x == 0);
    if (halted) {
        if (lead) halt.out(true);
        return;
    }
    const bool stop = op.prologue(s4, lead);
    if (lead) halt.out(stop);
    if (op.skip()) return;
    double acc[Op::NACC > 0 ? Op::NACC : 1];
#pragma unroll
    for (int d = 0; d < (Op::NACC > 0 ? Op::NACC : 1); ++d) acc[d] = 0.0;
    if constexpr (split) {
        if (has_first) {
            op.apply2(first, acc);
            op.store2(2 * g, first);
        }
        for (int64_t q = g + S; q < npair; q += S) {
            typename Op::Regs r;
            op.load2(2 * q, r);
            op.apply2(r, acc);
            op.store2(2 * q, r);
        }
    } else {
        for (int64_t q = g; q < npair; q += S) op.pair(2 * q, acc);
    }
    if ((n & 1) && g == (npair % S)) op.one(n - 1, acc);
#pragma unroll
    for (int d = 0; d < Op::NACC; ++d) {
        const double tot = mk_block_sum(acc[d], s4);
        if (threadIdx.x == 0) partials[(Op::SLOT0 + d) * MK_MAXP + blockIdx.x] = tot;
    }
    halt.template clear_tail<Op::NACC, Op::SLOT0>(partials);
}

__device__ __forceinline__ double2 mk_ld2(const double *p, int64_t i) {
    return *reinterpret_cast<const double2 *>(p + i);
}
__device__ __forceinline__ void mk_st2(double *p, int64_t i, double2 v) {
    *reinterpret_cast<double2 *>(p + i) = v;
}

#endif  // __HIPCC__
