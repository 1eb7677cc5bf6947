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
                             // (cache-resident problems); 2 every step of the grid is split into eight XCD-contiguous blocks;
                             // 3 stripes of `stripe` consecutive tiles dealt round-robin to the XCDs (banded matrices:
                             // an XCD marches through ITS stripe of every plane, so the x windows of the far
                             // off-diagonals -- one plane back, one plane ahead -- are still in its own L2)
                             // 4 as 3, but an XCD finishes a narrow strip of every plane -- all `nplanes` planes, `plane`
                             // tiles apart -- before it moves to its next strip: the reuse distance of a window (one and
                             // two planes) shrinks from plane/8 tiles to `stripe` tiles and fits the L2 beside the
                             // streamed matrix data
    int stripe;              // maps 3, 4: tiles per stripe (map 3: a power of two)
    int plane, nplanes;      // map 4: tiles per plane (the far off-diagonals' distance), planes (ntl = plane * nplanes)
    int nt;                  // fmt 5: the value stream is loaded non-temporally (read once: must not push x out of the L2)
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
                             // 4 windows + dictionary + one pattern byte per row instead of a word per nonzero,
                             // 5 windows + pattern byte per row + the raw values streamed in tile-sliced ELL order
                             // 6 / 7 / 8 wide tiles (rows <= 32 entries, <= 32 chunks): slots + values streamed /
                             // pattern byte + values streamed / pattern byte + dictionary
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
    // fmt 5: the values in tile-sliced ELL order and per tile {start of its block / 256, width} (mk_spmv_fmt5.h)
    const double *sval;
    const int32_t *sdesc;
    // fmt 6, 7, 8 (mk_spmv_fmtw.h): wmode = fmt - 6; window chunks per wave (4: 16-chunk cover, 8: 32-chunk cover); the
    // LDS slots in tile-sliced ELL order (fmt 6); sdesc holds FOUR ints per tile {value block, width, slot block, -}
    int wmode, wper;
    const uint16_t *sslot;
    const void *ptab;        // fmt 8: pattern entries {byte offset, value} (mk_spmv_fmtw.h) and per pattern {length | diagonal << 8}
    const int32_t *pinfo;
    int allwin;              // every tile of the matrix has windows (no tile ever takes the gather path)
    // fmt 9 (mk_spmv_fmt9.h): strides of the 7-point-class matrix, planes, bricks per line / plane, planes per chunk, chunks
    // (pid = pattern byte per row, ptab = 64 bytes per pattern)
    int64_t pen_L, pen_P;
    int pen_nz, pen_bx, pen_bpp, pen_zc, pen_chunks;
    // a rank's slab: offsets of the plane below / above in the input vector (-1: none, the march clamps as on one device); a
    // launch covers the planes [pen_za, pen_zb) and [pen_ya, pen_yb), each cut into chunks of pen_zc planes (whole product:
    // [0, nz) and nothing; overlapped halo exchange: the interior planes first, the slab's first and last planes afterwards)
    int64_t pen_xlo, pen_xhi;
    int pen_za, pen_zb, pen_ya, pen_yb;
    // general geometry (round 6): pen_gen 2 = the plane is not tiled by whole aligned bricks (L % 128, P % 4L, odd strides): the GEN
    // kernels only; 1 = an aligned geometry whose launch has leftover planes: the GEN kernel where the epilogue has one (its
    // masked last round replaces the unpipelined planes); pen_per > 0: bricks per XCD of the XCD-contiguous deal;
    // pen_xtop = the input vector's last entry (pair loads are clamped to it); pen_dump = where discarded rows are stored
    int pen_gen, pen_per;
    int pen_nol;                                            // the matrix has no +-L entries at all (a 5-point matrix: L is a fiction)
    int64_t pen_xtop;
    double *pen_dump;
    // resident tiles (fmt 3): LDS capacity per tile in nonzeros (multiple of 256), column phases and their width
    int rt_cap, rt_k, rt_w;
    int rt_c0;               // first column of the phases (0; a column block's first column: its phases cover ITS slice of x)
    int rt_reg;              // fmt 3: rows of <= 5 entries -- a second tile per workgroup rides in registers (mk_spmv_fmt3r.h)
    // ... and a product of MANY pair steps per workgroup runs as one launch per step (mk_spmv_launch_blocks): step0 / nsteps
    // select the steps of this launch (nsteps 0: all), the per-lane accumulators of the fused dots travel through `carry`
    int step0, nsteps;
    double *carry;
    int carry_in, carry_out;
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
double *mk_pen_dump();                           // mk_format.hip: the dump rows of the general-geometry brick march (one per context)
// stripe length (tiles, a power of two) of tile order 3; 0 = the matrix does not use it
static inline const mk_csr *mk_owner(const mk_csr *A) { return A->base ? A->base : A; }
static inline int mk_tile_stripe(const mk_csr *A) {
    static const char *env = getenv("MK_SPMV_STRIPE");
    int s = mk_owner(A)->want_stripe > 0 ? mk_owner(A)->want_stripe : (env ? atoi(env) : 0);
    if (s <= 0) s = 32;
    int p2 = 1;
    while (2 * p2 <= s) p2 *= 2;
    return p2;
}
// plane period (tiles) of tile order 4
static inline int mk_tile_plane(const mk_csr *A) {
    static const char *env = getenv("MK_SPMV_PLANE");
    if (mk_owner(A)->want_plane > 0) return mk_owner(A)->want_plane;
    return env ? atoi(env) : 0;
}
// order 4 needs whole planes and whole strips per XCD
static inline bool mk_tile_map4_ok(const mk_csr *A) {
    const int P = mk_tile_plane(A), S = mk_tile_stripe(A);
    return P > 0 && S > 0 && A->ntiles % P == 0 && P % (8 * S) == 0;
}
// Non-temporal accesses of a product: the value stream of fmt 5 / 6 / 7 (read once) and the product vector (written once)
// go past the caches so that the x windows stay in the L2.  Default: on when a vector is larger than the Infinity Cache
// (256 MiB).  Measured in one process each (tools/ab_spmv.py, AB_NT_Y): 512^3, variable coefficients 1 768 -> 1 711 us
// with the loads, 1 730 with the stores, 1 639 with both, CG pass 3.40 -> 3.27 ms; constant coefficients (fmt 4, stores
// only) 808 -> 768 us.  At 256^3 (134 MB per vector: the update kernel that reads the product finds it in the cache,
// and the caches hold a good part of the pass's working set) the 27-point product gains 5 % and the CG pass LOSES
// 3-7 %, with the stores, the loads or both: off there.
static inline int mk_stream_nt(const mk_csr *A) {
    static const char *env = getenv("MK_SPMV_NT");
    if (mk_owner(A)->want_nt >= 0) return mk_owner(A)->want_nt;
    if (env) return atoi(env);
    return (!A->comp_kind && !A->host_fn && 8 * A->nrows > (int64_t)256 * 1024 * 1024) ? 1 : 0;
}
static inline int mk_store_nt(const mk_csr *A) { return mk_stream_nt(A); }
static inline int mk_tile_map(const mk_csr *A) {
    if (A->comp_kind >= 4) return 0;
    if (A->comp_kind) return mk_tile_map(A->comp_kind == 3 ? A->comp_a : A->comp_b);
    if (A->host_fn) return 0;
    static const char *env = getenv("MK_SPMV_MAP");
    if (mk_owner(A)->want_map >= 0 || env) {
        int m = mk_owner(A)->want_map >= 0 ? mk_owner(A)->want_map : atoi(env);
        if (m == 4 && !mk_tile_map4_ok(A)) m = 2;
        return m;
    }
    if (mk_xcd_chunks(A)) return 1;
    // beyond the Infinity Cache: one front (0) -- except for the pattern kernel, which moves so little per tile that it
    // runs into the fabric on re-fetched x windows: XCD-contiguous blocks within every step of the grid keep neighbouring
    // tiles' windows in one L2 (512^3: fabric reads 4.4 -> 3.6 GB, 838 -> 790 us)
    const MkPlan *P = mk_csr_plan(A);
    return (P && P->fmt >= 4) ? 2 : 0;
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
    if (A->comp_kind >= 4) return mk_grid_spmv(A->ntiles);   // (the final launch walks the accumulated rows)
    if (A->comp_kind) return mk_grid_spmv_for(A->comp_kind == 3 ? A->comp_a : A->comp_b);   // the final launch's matrix
    int g = mk_grid_spmv(A->ntiles);
    if (getenv("MK_GRID_SPMV") || A->host_fn) return g;
    const MkPlan *P = mk_csr_plan(A);
    int64_t cap = mk_cap_spmv();
    if (P && P->cblocks.size() >= 2 && P->cblocks[0]->plan.fmt == 3 && A->ex.mode < 0) {   // column blocks as resident tiles
        int g3 = (int)(A->ntiles > MK_MAXP ? MK_MAXP : A->ntiles);
        if (g3 >= 8) g3 -= g3 % 8;
        return g3;
    }
    if (P && mk_fmt_march(P->fmt)) {                         // one workgroup per (brick, chunk) item, at most MK_MAXP
        const int64_t items = (int64_t)(P->pen_per > 0 ? 8 * P->pen_per : P->pen_bpp) * P->pen_chunks;
        return (int)(items > MK_MAXP ? MK_MAXP : items);
    }
    if (P && P->fmt == 3) {                                  // as many as fit at once with one tile in LDS each
        int g3 = (int)(A->ntiles > MK_MAXP ? MK_MAXP : A->ntiles);
        if (g3 >= 8) g3 -= g3 % 8;
        return g3;
    }
    if (P && (P->fmt == 4 || P->fmt == 5)) {                 // up to 7 per CU, as many as LDS holds (8 per CU: 512^3 +3 %,
                                                             // 2-D n = 1e6 -4 %, and MINRES' epilogue spills at 64 registers)
        int64_t top = 128 * (int64_t)P->wchunks + 2;
        if (P->covered != A->ntiles && top < MK_PROD_LDS) top = MK_PROD_LDS;
        const int64_t lds = 8 * (top + MK_BLOCK) + (P->fmt == 4 ? 16 : 4) * (int64_t)(P->npat * P->pmax + 4) + 2560;   // + static arrays
        int64_t per_cu = (160 * 1024) / lds;
        per_cu = per_cu > 7 ? 7 : (per_cu < 1 ? 1 : per_cu);
        cap = 256 * per_cu;
    }
    else if (P && P->fmt >= 6) {                             // wide tiles: 4 per CU (128 registers; fmt 8: 7), as many as LDS holds
        int64_t top = 128 * (int64_t)P->wchunks + 2;
        if (P->covered != A->ntiles && top < MK_PROD_LDS) top = MK_PROD_LDS;
        const int64_t lds = 8 * (top + MK_BLOCK) + 4 * (int64_t)((P->fmt == 7 ? P->npat * P->pmax : 0) + 4) + 2560;   // + static arrays
        int64_t per_cu = (160 * 1024) / lds;
        const int64_t top_cu = (P->fmt == 8) ? 7 : 4;
        per_cu = per_cu > top_cu ? top_cu : (per_cu < 1 ? 1 : per_cu);
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
    v.stripe = (v.map == 3) ? mk_tile_stripe(A) : 0;
    if (v.map == 4) {
        v.stripe = mk_tile_stripe(A);
        v.plane = mk_tile_plane(A);
        v.nplanes = (int)(A->ntiles / v.plane);
    }
    v.nt = mk_stream_nt(A);
    v.nops = A->nops;
    for (int k = 0; k < A->nops; ++k) v.ops[k] = A->ops[k];
    v.tiles = nullptr;
    v.ntl = A->ntiles;
    v.poff = 0;
    v.part = 0;
    const MkPlan *P = mk_csr_plan(A);
    v.fmt = P ? P->fmt : 0;
    if (mk_fmt_march(v.fmt)) {
        v.pid = P->d_pid;
        v.sval = P->d_sval;                                  // (format 10: seven value arrays, position major)
        v.ptab = P->d_ptab;
        v.npat = P->npat;
        v.pen_L = P->pen_L;
        v.pen_P = P->pen_P;
        v.pen_nz = P->pen_nz;
        v.pen_bx = P->pen_bx;
        v.pen_bpp = P->pen_bpp;
        v.pen_zc = P->pen_zc;
        v.pen_chunks = P->pen_chunks;
        v.pen_xlo = P->pen_xlo;
        v.pen_xhi = P->pen_xhi;
        v.pen_za = 0;
        v.pen_zb = P->pen_nz;
        v.pen_ya = v.pen_yb = 0;
        v.pen_gen = P->pen_gen;
        v.pen_per = P->pen_per;
        v.pen_nol = P->pen_nol;
        v.pen_xtop = A->x_len() - 1;
        v.pen_dump = P->pen_gen ? mk_pen_dump() : nullptr;
    } else if (v.fmt == 3) {
        v.rt_cap = P->rt_cap;
        v.rt_k = P->rt_k;
        v.rt_w = P->rt_w;
        v.rt_reg = P->rt_reg;
        v.carry = P->d_carry;
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
        v.sval = P->d_sval;
        v.sdesc = P->d_sdesc;
        v.allwin = (P->covered == A->ntiles) ? 1 : 0;
        v.wper = P->wide ? 8 : 4;
        v.wmode = v.fmt >= 6 ? v.fmt - 6 : 0;
        v.sslot = P->d_sslot;
        v.ptab = P->d_ptab;
        v.pinfo = P->d_pinfo;
    }
    return v;
}

// The brick march of a slab under an overlapped halo exchange: the planes that need no received entry run while the
// messages travel, the others afterwards.  `thin` (round 6; loops with general-geometry kernels: plain products and CG):
// only the slab's FIRST and LAST plane wait for the messages -- one plane per (brick, chunk) item, run unpipelined: one memory
// round trip -- and the interior's leftover planes go through the masked last round of the GEN kernel.  Otherwise (round 5) the
// boundary launch takes the first MK_PEN_R planes of a slab with a lower neighbour and the last 1 .. MK_PEN_R of one with an
// upper neighbour, so that the interior is a whole number of pipelined rounds.  False: too few planes to split.
static inline bool mk_pen_split(const MkPlan *P, int *za, int *zb, bool thin = false) {
    constexpr int R = 6;                                    // (= MK_PEN_R, mk_spmv_fmt9.h; asserted there)
    const int nz = P->pen_nz;
    if (thin) {
        const int lo = (P->pen_xlo >= 0) ? 1 : 0, hi = (P->pen_xhi >= 0) ? nz - 1 : nz;
        if (hi - lo < R || (lo == 0 && hi == nz)) return false;
        *za = lo;
        *zb = hi;
        return true;
    }
    const int lo = (P->pen_xlo >= 0) ? R : 0;
    int hi_s = nz;
    if (P->pen_xhi >= 0) hi_s = lo + ((nz - lo - 1) / R) * R;
    if (hi_s - lo < R || (lo == 0 && hi_s == nz)) return false;
    *za = lo;
    *zb = hi_s;
    return true;
}
static inline bool mk_pen_tail_gen() {                      // (MK_PEN_TAIL_GEN=0: round 5's split of a slab -- six + leftover boundary planes -- for A/B runs)
    static const char *env = getenv("MK_PEN_TAIL_GEN");
    return !env || atoi(env) != 0;
}
static inline int mk_pen_items(const MkCsrView &v) {
    const int n1 = (v.pen_zb - v.pen_za + v.pen_zc - 1) / v.pen_zc, n2 = (v.pen_yb - v.pen_ya + v.pen_zc - 1) / v.pen_zc;
    const int64_t items = (int64_t)(v.pen_per > 0 ? 8 * v.pen_per : v.pen_bpp) * (n1 + n2);   // (8 per == bpp on an aligned geometry)
    return (int)(items > MK_MAXP ? MK_MAXP : (items < 1 ? 1 : items));
}

// view of the interior (part 1) or boundary (part 2) tiles of a partitioned matrix; poff2 = grid of part 1
static inline MkCsrView mk_view_part(const mk_csr *A, int part, int poff2, bool thin = false) {
    MkCsrView v = mk_view(A);
    if (mk_fmt_march(v.fmt)) {                              // plane ranges instead of tile lists
        int za = 0, zb = v.pen_nz;
        mk_pen_split(mk_csr_plan(A), &za, &zb, thin);       // (the caller checked that the slab splits)
        if (part == 2) {
            v.pen_za = 0;
            v.pen_zb = za;
            v.pen_ya = zb;
            v.pen_yb = v.pen_nz;
        } else {
            v.pen_za = za;
            v.pen_zb = zb;
            // thin split: the interior is no whole number of rounds any more -- its leftover planes take the GEN kernel's
            // masked round (the epilogue has one: `thin` says so) instead of one unpipelined plane after the other
            if (thin && !v.pen_gen && (zb - za) % 6 != 0) {
                v.pen_gen = 1;
                v.pen_dump = mk_pen_dump();
            }
        }
        v.poff = (part == 2) ? poff2 : 0;
        v.part = part;
        return v;
    }
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
typedef unsigned mk_u2 __attribute__((ext_vector_type(2)));
typedef unsigned mk_u4 __attribute__((ext_vector_type(4)));

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
// Compile-time budget.  The brick-march kernels (unrolled, software-pipelined) are by far the most expensive instantiations of
// mk_spmv_kernel, so an epilogue says where it can never meet them: `static constexpr bool NO_MARCH = true` -- the
// least-squares loops' (their operators are rectangular; the march formats are for square 7-point-class matrices) -- and
// `static constexpr bool SYM_MARCH = true` marks the few that may meet format 11, the symmetric twin (plain products and CG:
// mk_csr_march_pref caps every other loop's matrix at format 10).  A launch that falls outside takes the CSR gather kernel
// on the same arrays (format 0: same row sums bit for bit), which only a format forced by hand can bring about.
template <class Epi, class = void>
struct MkNoMarch : std::false_type {};
template <class Epi>
struct MkNoMarch<Epi, std::enable_if_t<Epi::NO_MARCH>> : std::true_type {};
template <class Epi, class = void>
struct MkSymMarch : std::false_type {};
template <class Epi>
struct MkSymMarch<Epi, std::enable_if_t<Epi::SYM_MARCH>> : std::true_type {};

template <class Epi, class = void>
struct MkHasRowX : std::false_type {};
template <class Epi>
struct MkHasRowX<Epi, std::void_t<decltype(std::declval<Epi &>().row_x((int64_t)0, 0.0, 0.0, (double *)nullptr))>>
    : std::true_type {};

typedef unsigned mk_u2 __attribute__((ext_vector_type(2)));
typedef unsigned mk_u4 __attribute__((ext_vector_type(4)));

// optional epilogue hooks for kernels with a software pipeline (the brick march, mk_spmv_fmt9.h): an epilogue that reads
// vectors at its own row (r0[r], r1[r], w[r] ...) declares them -- `static constexpr int NPF`, `const double *pf_vec(int j)`
// -- and takes their values as arguments: `row_pf(r, s, o, acc)` = `row(r, s, acc)` with o[j] = pf_vec(j)[r], and, where the
// epilogue's own vector is the product's input, `row_x_pf(r, s, xr, o, acc)`.  The kernel issues those loads at the TOP of a
// step, before the prefetches of later planes, so that consuming them at the end of the step waits only for loads that are
// older -- a load issued inside `row` is the youngest of the queue and waiting for it drains the whole pipeline.
template <class Epi, class = void>
struct MkHasRowPf : std::false_type {};
template <class Epi>
struct MkHasRowPf<Epi, std::void_t<decltype(std::declval<Epi &>().row_pf((int64_t)0, 0.0, (const double *)nullptr, (double *)nullptr))>>
    : std::true_type {};
template <class Epi, class = void>
struct MkHasRowXPf : std::false_type {};
template <class Epi>
struct MkHasRowXPf<Epi, std::void_t<decltype(std::declval<Epi &>().row_x_pf((int64_t)0, 0.0, 0.0, (const double *)nullptr, (double *)nullptr))>>
    : std::true_type {};

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
    int t = A.tiles ? mk_sload(A.tiles + p) : (int)p;
    if (A.map == 3 && !A.tiles && gridDim.x % 8 == 0) {      // p counts the tiles of this XCD's stripes
        const int S = A.stripe, e = (int)(blockIdx.x % 8);
        t = (((int)p / S) * 8 + e) * S + ((int)p & (S - 1));
    } else if (A.map == 4 && !A.tiles && gridDim.x % 8 == 0) {   // p = ((strip * nplanes) + plane) * S + position in the strip
        const unsigned S = A.stripe, e = blockIdx.x % 8, q = (unsigned)p / S;
        t = (int)((q % (unsigned)A.nplanes) * (unsigned)A.plane + ((q / (unsigned)A.nplanes) * 8 + e) * S + (unsigned)p % S);
    }
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

// Tile order of a launch.  Workgroup b is dispatched to XCD b % 8 (observed, MI355X_MICROARCH.md) and each XCD has its
// own 4 MiB L2; the order only affects speed (and which rows a workgroup's partial sums cover).
struct MkTileRange {
    int64_t pos, stride, end;
};
__device__ __forceinline__ MkTileRange mk_tile_range(const MkCsrView &A) {
    const int G = gridDim.x;
    const bool x8 = (G % 8 == 0);
    MkTileRange t;
    if (A.map == 1 && x8) {                                  // XCD b % 8 sweeps its own contiguous eighth
        const int64_t chunk = (A.ntl + 7) / 8, c0 = (int64_t)(blockIdx.x % 8) * chunk;
        t.pos = c0 + blockIdx.x / 8;
        t.stride = G / 8;
        t.end = (c0 + chunk < A.ntl) ? c0 + chunk : A.ntl;
    } else if (A.map == 2 && x8) {                           // every step of the grid: eight XCD-contiguous blocks
        t.pos = (int64_t)(blockIdx.x % 8) * (G / 8) + blockIdx.x / 8;
        t.stride = G;
        t.end = A.ntl;
    } else if (A.map == 3 && x8 && !A.tiles) {               // XCD b % 8 takes stripes b % 8, b % 8 + 8, ... of S tiles
        const int64_t S = A.stripe, e = blockIdx.x % 8;
        const int64_t nst = (A.ntl + S - 1) / S;             // stripes in all; the last one may be short
        const int64_t cnt = (nst > e) ? (nst - e + 7) / 8 : 0;
        t.pos = blockIdx.x / 8;
        t.stride = G / 8;
        t.end = cnt * S - ((cnt > 0 && (nst - 1) % 8 == e) ? nst * S - A.ntl : 0);
    } else if (A.map == 4 && x8 && !A.tiles) {               // (the host guarantees ntl = plane * nplanes, plane % (8 S) == 0)
        t.pos = blockIdx.x / 8;
        t.stride = G / 8;
        t.end = A.ntl / 8;
    } else {
        t.pos = blockIdx.x;
        t.stride = G;
        t.end = A.ntl;
    }
    return t;
}

// Row pointers of a tile (one load per lane: a row's end is its neighbour's start and travels through LDS);
// fetched ahead so that their latency is not on the critical path.  p: position in the tile list of this launch.
__device__ __forceinline__ void mk_load_meta(const MkCsrView &A, int64_t p, int64_t end, MkTileMeta &m) {
    m.p_lo = m.p_hi = m.my_lo = 0;
    if (p < end) {
        const int64_t tile = mk_tile_at(A, p);
        const int64_t r0 = tile * MK_ROWS_PER_TILE;
        const int64_t rend = (r0 + MK_ROWS_PER_TILE < A.nrows) ? r0 + MK_ROWS_PER_TILE : A.nrows;
        const int64_t r = r0 + (int)threadIdx.x;
        m.p_lo = mk_sload(A.indptr + r0);
        m.p_hi = mk_sload(A.indptr + rend);
        m.my_lo = A.indptr[(r < rend) ? r : rend];          // rows past the end start (and end) at p_hi
    }
}

#include "mk_spmv_fmt0.h"
#include "mk_spmv_fmt1.h"
#include "mk_spmv_fmt24.h"
#include "mk_spmv_fmt3.h"
#include "mk_spmv_fmt3r.h"
#include "mk_spmv_fmt5.h"
#include "mk_spmv_fmtw.h"
#include "mk_spmv_fmt9.h"

constexpr int MK_FMT_WIDE = 7;                       // template values of the wide kernels (6 = format 5, non-temporal):
constexpr int MK_FMT_WIDE_DICT = 8;                  // 7 streams values (fmt 6, 7), 8 takes them from the dictionary (fmt 8),
constexpr int MK_FMT_WIDE_NT = 9;                    // 9 = 7 with non-temporal loads of the streams
constexpr int MK_FMT_PAIR = 10;                      // format 3 with a second tile per workgroup in registers (rows <= 5 entries)
constexpr int MK_FMT_PENCIL = 11;                    // format 9: z-marching bricks (mk_spmv_fmt9.h)
constexpr int MK_FMT_PENCIL_STREAM = 12;             // format 10: the same march with streamed values (no dictionary)
constexpr int MK_FMT_PENCIL_SYM = 13;                // format 11: ... of a symmetric matrix (diagonal and upper values only)
constexpr int MK_FMT_PENCIL_G = 14;                  // formats 9 / 10 / 11 on a general geometry (mk_spmv_fmt9.h, GEN): 14 / 15 / 16
constexpr int MK_FMT_PENCIL_STREAM_G = 15;
constexpr int MK_FMT_PENCIL_SYM_G = 16;

template <int FMT, bool PROG, class Epi, int NACC>
__device__ __forceinline__ void mk_spmv_tiles(const MkCsrView &A, const double *__restrict__ x, Epi &epi,
                                              double *prod, double *xw, double (&acc)[NACC]) {
    if constexpr (FMT == 0) mk_spmv_tiles_fmt0<PROG>(A, x, epi, prod, xw, acc);
    else if constexpr (FMT == 1) mk_spmv_tiles_fmt1<PROG>(A, x, epi, prod, xw, acc);
    else if constexpr (FMT == 3) mk_spmv_tiles_fmt3<PROG>(A, x, epi, prod, xw, acc);
    else if constexpr (FMT == MK_FMT_PAIR) mk_spmv_tiles_fmt3r<5, PROG>(A, x, epi, prod, xw, acc);
    else if constexpr (FMT == MK_FMT_PENCIL) mk_spmv_tiles_fmt9<PROG, false, false, false>(A, x, epi, xw, acc);
    else if constexpr (FMT == MK_FMT_PENCIL_STREAM) mk_spmv_tiles_fmt9<PROG, true, false, false>(A, x, epi, xw, acc);
    else if constexpr (FMT == MK_FMT_PENCIL_SYM) mk_spmv_tiles_fmt9<PROG, true, true, false>(A, x, epi, xw, acc);
    else if constexpr (FMT == MK_FMT_PENCIL_G) mk_spmv_tiles_fmt9<PROG, false, false, true>(A, x, epi, xw, acc);
    else if constexpr (FMT == MK_FMT_PENCIL_STREAM_G) mk_spmv_tiles_fmt9<PROG, true, false, true>(A, x, epi, xw, acc);
    else if constexpr (FMT == MK_FMT_PENCIL_SYM_G) mk_spmv_tiles_fmt9<PROG, true, true, true>(A, x, epi, xw, acc);
    else if constexpr (FMT == MK_FMT_WIDE || FMT == MK_FMT_WIDE_DICT || FMT == MK_FMT_WIDE_NT)
        mk_spmv_tiles_wide<FMT == MK_FMT_WIDE_DICT, FMT == MK_FMT_WIDE_NT, PROG>(A, x, epi, prod, xw, acc);
    else if constexpr (FMT >= 5) mk_spmv_tiles_fmt5<PROG, FMT == 6>(A, x, epi, prod, xw, acc);
    else mk_spmv_tiles_fmt24<FMT, PROG>(A, x, epi, prod, xw, acc);
}

// Gate: a test on a global sum that must be settled before the product may start (e.g. the loop
// condition on ||r||).  Every workgroup evaluates it identically from the previous kernel's partial
// sums; `open` returns whether to run the product and may request a halt through *stop.
struct MkNoGate {
    __device__ bool open(double *, bool, bool *) { return true; }
};

template <class Epi, class Gate, bool PROG, int FMT>
__global__ __launch_bounds__(MK_BLOCK, (FMT == 0 || FMT == 3 || FMT == 10) ? 8 : ((FMT >= 11 && FMT <= 16) ? MK_PEN_OCC : ((FMT == 7 || FMT == 9) ? 4 : (FMT >= 4 ? 7 : 4)))) void mk_spmv_kernel(MkCsrView A, const double *__restrict__ x, Epi epi,
                                                           Gate gate, MkHalt halt, double *__restrict__ partials) {
    // fmt 0 / 1: products [MK_PROD_LDS doubles], then the windows.  fmt 2 has no product staging: its windows and
    // packed words share the space the gather path of uncovered tiles uses for products (never live together)
    extern __shared__ __attribute__((aligned(16))) double mk_smem[];
    double *prod = mk_smem;
    double *xw = (FMT == 2 || (FMT >= 4 && FMT != 10)) ? mk_smem : mk_smem + MK_PROD_LDS;
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
    if constexpr (FMT == 10 && Epi::NACC > 0) {
        if (A.carry_in) {                                           // a later step of a product split into launches
#pragma unroll
            for (int d = 0; d < Epi::NACC; ++d) acc[d] = A.carry[((size_t)d * gridDim.x + blockIdx.x) * MK_BLOCK + threadIdx.x];
        }
    }
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
    if constexpr (FMT == 10) {
        if (A.carry_out) {                                          // not the last step: the accumulators travel on
            if constexpr (Epi::NACC > 0) {
#pragma unroll
                for (int d = 0; d < Epi::NACC; ++d) A.carry[((size_t)d * gridDim.x + blockIdx.x) * MK_BLOCK + threadIdx.x] = acc[d];
            }
            return;
        }
    }
#pragma unroll
    for (int d = 0; d < Epi::NACC; ++d) {
        const double tot = mk_block_sum(acc[d], s4);
        if (threadIdx.x == 0) partials[(Epi::SLOT0 + d) * MK_MAXP + A.poff + blockIdx.x] = tot;
    }
    if (A.part != 1) halt.template clear_tail<Epi::NACC, Epi::SLOT0>(partials, A.poff + (int)gridDim.x);
}

// does this loop's epilogue have a kernel for the plan's march format?  (No: mk_spmv_launch_fmt runs the CSR gather kernel
// on the matrix's arrays instead -- over all rows, so a caller must not cut such a product into plane ranges.)
template <class Epi>
static inline bool mk_march_kernel_for(const MkPlan *P) {
    if (!P || !mk_fmt_march(P->fmt) || MkNoMarch<Epi>::value) return false;
    return MkSymMarch<Epi>::value || (P->fmt != 11 && P->pen_gen != 2);
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
    if (mk_fmt_march(v.fmt) && (MkNoMarch<Epi>::value || ((v.fmt == 11 || v.pen_gen == 2) && !MkSymMarch<Epi>::value))) {
        MkCsrView w = v;                                     // (see MkNoMarch: a format forced by hand on a loop that has no such kernel)
        w.fmt = 0;
        hipLaunchKernelGGL((mk_spmv_kernel<Epi, Gate, PROG, 0>), dim3(grid), dim3(MK_BLOCK), lds, st, w, x, epi, gate, halt, partials);
        return;
    }
    if (mk_fmt_march(v.fmt)) {
        if constexpr (MkSymMarch<Epi>::value) {              // general geometry / masked leftover round (mk_spmv_fmt9.h, GEN)
            if (v.pen_gen) {
                if (v.fmt == 9) {
                    lds = sizeof(double) * (size_t)MK_PEN_LDS + 64 * (size_t)v.npat;
                    hipLaunchKernelGGL((mk_spmv_kernel<Epi, Gate, PROG, MK_FMT_PENCIL_G>), dim3(grid), dim3(MK_BLOCK), lds, st, v, x, epi,
                                       gate, halt, partials);
                } else if (v.fmt == 10) {
                    lds = sizeof(double) * (size_t)MK_PEN_LDS;
                    hipLaunchKernelGGL((mk_spmv_kernel<Epi, Gate, PROG, MK_FMT_PENCIL_STREAM_G>), dim3(grid), dim3(MK_BLOCK), lds, st, v, x,
                                       epi, gate, halt, partials);
                } else {
                    lds = sizeof(double) * (size_t)MK_PEN_LDS_SYM;
                    hipLaunchKernelGGL((mk_spmv_kernel<Epi, Gate, PROG, MK_FMT_PENCIL_SYM_G>), dim3(grid), dim3(MK_BLOCK), lds, st, v, x,
                                       epi, gate, halt, partials);
                }
                return;
            }
        }
        if constexpr (!MkNoMarch<Epi>::value) {
            if (v.fmt == 9) {                                // two plane images of the brick + the dump row
                lds = sizeof(double) * (size_t)MK_PEN_LDS + 64 * (size_t)v.npat;
                hipLaunchKernelGGL((mk_spmv_kernel<Epi, Gate, PROG, MK_FMT_PENCIL>), dim3(grid), dim3(MK_BLOCK), lds, st, v, x, epi,
                                   gate, halt, partials);
            } else if (v.fmt == 10) {                        // ... the same without a pattern table
                lds = sizeof(double) * (size_t)MK_PEN_LDS;
                hipLaunchKernelGGL((mk_spmv_kernel<Epi, Gate, PROG, MK_FMT_PENCIL_STREAM>), dim3(grid), dim3(MK_BLOCK), lds, st, v, x,
                                   epi, gate, halt, partials);
            } else if constexpr (MkSymMarch<Epi>::value) {   // ... of a symmetric matrix: + the image of the plane's values
                lds = sizeof(double) * (size_t)MK_PEN_LDS_SYM;
                hipLaunchKernelGGL((mk_spmv_kernel<Epi, Gate, PROG, MK_FMT_PENCIL_SYM>), dim3(grid), dim3(MK_BLOCK), lds, st, v, x,
                                   epi, gate, halt, partials);
            }
        }
        return;
    }
    if (v.fmt == 4) {                                        // windows + pattern table, or the gather path's products
        size_t wtop = (size_t)(128 * v.wchunks + 2);
        if (!v.allwin && wtop < (size_t)MK_PROD_LDS) wtop = (size_t)MK_PROD_LDS;
        lds = sizeof(double) * (wtop + MK_BLOCK) + 16 * (size_t)(v.npat * v.pmax + 1);   // windows, zeros, table
        hipLaunchKernelGGL((mk_spmv_kernel<Epi, Gate, PROG, 4>), dim3(grid), dim3(MK_BLOCK), lds, st, v, x, epi, gate,
                           halt, partials);
    } else if (v.fmt == 5) {                                 // windows + zeros + offset table, or the gather path's products
        size_t wtop = (size_t)(128 * v.wchunks + 2);
        if (!v.allwin && wtop < (size_t)MK_PROD_LDS) wtop = (size_t)MK_PROD_LDS;
        lds = sizeof(double) * (wtop + MK_BLOCK) + 4 * (size_t)(v.npat * v.pmax + 4);
        if (v.nt)                                       // (template value 6 = format 5 with non-temporal value loads)
            hipLaunchKernelGGL((mk_spmv_kernel<Epi, Gate, PROG, 6>), dim3(grid), dim3(MK_BLOCK), lds, st, v, x, epi, gate,
                               halt, partials);
        else
            hipLaunchKernelGGL((mk_spmv_kernel<Epi, Gate, PROG, 5>), dim3(grid), dim3(MK_BLOCK), lds, st, v, x, epi, gate,
                               halt, partials);
    } else if (v.fmt >= 6) {                                 // wide tiles: zeros + windows (or the gather path's products) + pattern words
        size_t wtop = (size_t)(128 * v.wchunks + 2);
        if (!v.allwin && wtop < (size_t)MK_PROD_LDS) wtop = (size_t)MK_PROD_LDS;
        lds = sizeof(double) * (wtop + MK_BLOCK) + 4 * (size_t)((v.fmt == 7 ? v.npat * v.pmax : 0) + 4);
        if (v.fmt == 8)
            hipLaunchKernelGGL((mk_spmv_kernel<Epi, Gate, PROG, MK_FMT_WIDE_DICT>), dim3(grid), dim3(MK_BLOCK), lds, st, v, x,
                               epi, gate, halt, partials);
        else if (v.nt)
            hipLaunchKernelGGL((mk_spmv_kernel<Epi, Gate, PROG, MK_FMT_WIDE_NT>), dim3(grid), dim3(MK_BLOCK), lds, st, v, x,
                               epi, gate, halt, partials);
        else
            hipLaunchKernelGGL((mk_spmv_kernel<Epi, Gate, PROG, MK_FMT_WIDE>), dim3(grid), dim3(MK_BLOCK), lds, st, v, x, epi,
                               gate, halt, partials);
    } else if (v.fmt == 3) {                                 // the tile's values and columns
        lds = (size_t)v.rt_cap * 12;
        if (v.rt_reg && !v.tiles)
            hipLaunchKernelGGL((mk_spmv_kernel<Epi, Gate, PROG, MK_FMT_PAIR>), dim3(grid), dim3(MK_BLOCK), lds, st, v, x, epi,
                               gate, halt, partials);
        else
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
    if (grid < 1) grid = 1;                                  // (a launch never vanishes: every kernel hands the halt word on)
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
    static constexpr bool NO_MARCH = MkNoMarch<Epi>::value, SYM_MARCH = MkSymMarch<Epi>::value;
    Epi e;
    double *ysum;
    __device__ void prologue(double *s4) { e.prologue(s4); }
    __device__ double xin(double v) const { return e.xin(v); }
    __device__ void row(int64_t r, double s, double *) { ysum[r] = s; }
};

// Reduced operators (mk_csr_create_reduced): z = 0 ; z[col_indices] = x   and   y = t[row_indices]  (linop.py:566-575)
static __global__ __launch_bounds__(MK_BLOCK) void mk_scatter_kernel(int64_t cnt, const int32_t *__restrict__ idx,
                                                                    const double *__restrict__ x, double *__restrict__ z) {
    for (int64_t k = (int64_t)blockIdx.x * MK_BLOCK + threadIdx.x; k < cnt; k += (int64_t)gridDim.x * MK_BLOCK) z[idx[k]] = x[k];
}
static __global__ __launch_bounds__(MK_BLOCK) void mk_gather_kernel(int64_t cnt, const int32_t *__restrict__ idx,
                                                                   const double *__restrict__ t, double *__restrict__ y) {
    for (int64_t k = (int64_t)blockIdx.x * MK_BLOCK + threadIdx.x; k < cnt; k += (int64_t)gridDim.x * MK_BLOCK) y[k] = t[idx[k]];
}

// Block operators (mk_csr_create_block): a block's finished row sums are added to the block row's accumulator,
// `y_i += B_ij * x_j` (blkop.py:94); the first block of a row adds to the +0.0 the reference's np.zeros holds.
template <class Epi>
struct MkAccumOf {
    static constexpr int NACC = 0, SLOT0 = 0;
    Epi e;
    double *ysum;
    int first;
    __device__ void prologue(double *s4) { e.prologue(s4); }
    __device__ double xin(double v) const { return e.xin(v); }
    __device__ void row(int64_t r, double s, double *) { ysum[r] = (first ? 0.0 : ysum[r]) + s; }
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
    static constexpr bool NO_MARCH = MkNoMarch<Epi>::value, SYM_MARCH = MkSymMarch<Epi>::value;
    const double *t1;
    __device__ void prologue(double *s4) { this->e.prologue(s4); }
    __device__ double xin(double v) const { return this->e.xin(v); }
    __device__ void row(int64_t r, double s, double *acc) { this->e.row(r, SIGN > 0 ? t1[r] + s : t1[r] - s, acc); }
};
template <class Epi>
struct MkNoXin : MkWrapBase<Epi> {   // the outer product of `A*(B*x)`: its input B*x was formed from xin(x) already
    static constexpr int NACC = Epi::NACC, SLOT0 = Epi::SLOT0;
    static constexpr bool NO_MARCH = MkNoMarch<Epi>::value, SYM_MARCH = MkSymMarch<Epi>::value;
    __device__ void prologue(double *s4) { this->e.prologue(s4); }
    __device__ double xin(double v) const { return v; }
    __device__ void row(int64_t r, double s, double *acc) { this->e.row(r, s, acc); }
};

// `next` yields the MkHalt of each launch (one per kernel: the halting protocol alternates the flag word)
template <class Epi, class Gate, class HaltSrc>
static inline void mk_spmv_launch_blocks(const mk_csr *A, int grid, hipStream_t st, const double *x, const Epi &epi,
                                         const Gate &gate, HaltSrc &&next, double *partials) {
    if (A->comp_kind == 5) {
        // Restriction of a device matrix: scatter x into a zero vector of the base's width, the base's complete product
        // (gate with its side effects; the epilogue's on-the-fly scaling applies to the scattered entries), gather the
        // chosen rows, and feed them to the real epilogue as a matrix-free operator's result is.
        const mk_csr *B = A->comp_a;
        const MkReduced &R = *A->red;
        auto blocks_for = [](int64_t cnt) { int64_t g = (cnt + MK_BLOCK - 1) / MK_BLOCK; return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); };
        hipMemsetAsync(R.d_z, 0, sizeof(double) * (size_t)(B->ncols > 0 ? B->ncols : 1), st);
        hipLaunchKernelGGL(mk_scatter_kernel, dim3(blocks_for(A->ncols)), dim3(MK_BLOCK), 0, st, A->ncols, R.d_cols, x, R.d_z);
        MkCsrView v1 = mk_view(B);
        v1.part = 1;
        mk_spmv_launch_view(v1, mk_grid_spmv_for(B), st, R.d_z, MkPartialOf<Epi>{epi, R.d_t}, gate, next(), partials);
        hipLaunchKernelGGL(mk_gather_kernel, dim3(blocks_for(A->nrows)), dim3(MK_BLOCK), 0, st, A->nrows, R.d_rows, R.d_t,
                           A->d_comp_tmp);
        MkCsrView v = mk_view(A);
        v.cb_mode = 2;
        v.y_ext = A->d_comp_tmp;
        v.part = 2;
        mk_spmv_launch_view(v, grid, st, x, epi, gate, next(), partials);
        return;
    }
    if (A->comp_kind == 4) {
        // Grid of device matrices: one launch per block into the rows' accumulators (the first launch evaluates the
        // gate with its side effects, the others repeat its decision), then the accumulated rows go through the real
        // epilogue exactly as a matrix-free operator's result does.
        const MkBlockGrid &G = *A->grid;
        bool any = false;
        for (int i = 0; i < G.nbr; ++i) {
            bool first_in_row = true;
            for (int j = 0; j < G.nbc; ++j) {
                const mk_csr *B = G.blk[(size_t)i * G.nbc + j];
                if (!B) continue;
                const double *xs = x + G.coff[j];
                if (G.coff[j] & 1) {                         // (the kernels read x in 16-byte pairs)
                    hipMemcpyAsync(G.d_xtmp, xs, sizeof(double) * (size_t)B->ncols, hipMemcpyDeviceToDevice, st);
                    xs = G.d_xtmp;
                }
                MkCsrView v = mk_view(B);
                v.part = any ? 2 : 1;
                mk_spmv_launch_view(v, mk_grid_spmv_for(B), st, xs,
                                    MkAccumOf<Epi>{epi, A->d_comp_tmp + G.roff[i], first_in_row ? 1 : 0}, gate, next(), partials);
                any = true;
                first_in_row = false;
            }
            if (first_in_row && G.roff[i + 1] > G.roff[i])   // a block row without blocks: zeros
                hipMemsetAsync(A->d_comp_tmp + G.roff[i], 0, sizeof(double) * (size_t)(G.roff[i + 1] - G.roff[i]), st);
        }
        MkCsrView v = mk_view(A);
        v.cb_mode = 2;
        v.y_ext = A->d_comp_tmp;
        v.part = any ? 2 : 0;
        mk_spmv_launch_view(v, grid, st, x, epi, gate, next(), partials);
        return;
    }
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
        MkCsrView v = mk_view(A);
        // Format 3, pair kernel, many steps per workgroup (4e6 rows: 3.8): over the steps the workgroups of an XCD drift out of
        // the lockstep the column phases live on, until every gather pulls its own sector through the fabric (1.38 GB per
        // product for 0.31 GB of data).  One LAUNCH per step re-aligns them for the price of a kernel boundary: the first
        // launch evaluates the gate with its side effects, the others repeat its decision, the per-lane accumulators of the
        // fused dots travel through a carry buffer (same additions in the same order: same bits), the last launch reduces.
        const int64_t per = 2 * (int64_t)grid;
        const int64_t steps = (v.ntl + per - 1) / per;
        if (v.fmt == 3 && v.rt_reg && !v.tiles && v.carry && v.map == 0 && steps >= 2 && steps <= 64 && Epi::NACC <= MK_CARRY_SLOTS) {
            for (int64_t k = 0; k < steps; ++k) {
                MkCsrView w = v;
                w.step0 = (int)k;
                w.nsteps = 1;
                w.carry_in = k > 0;
                w.carry_out = k + 1 < steps;
                w.part = k ? 2 : 1;
                mk_spmv_launch_view(w, grid, st, x, epi, gate, next(), partials);
            }
            return;
        }
        mk_spmv_launch_view(v, grid, st, x, epi, gate, next(), partials);
        return;
    }
    for (size_t k = 0; k < K; ++k) {
        MkCsrView v = mk_view(A);
        const mk_csr *B = P->cblocks[k];
        v.indptr = B->d_indptr;
        v.indices = B->d_indices;
        v.data = B->d_data;
        if (B->plan.fmt == 3) {                              // (resident tiles, one phase: mk_format.hip cblocks_build)
            v.fmt = 3;
            v.rt_cap = B->plan.rt_cap;
            v.rt_k = B->plan.rt_k;
            v.rt_w = B->plan.rt_w;
            v.rt_c0 = B->plan.rt_c0;
            v.rt_reg = 0;
        }
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
