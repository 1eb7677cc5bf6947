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
    int xcd_chunks;          // 1: each XCD sweeps its own contiguous eighth of the tiles (cache-resident problems)
    int nops;                // row program of a composed operator (mk_csr_compose); 0 for a plain matrix
    mk_rowop ops[MK_ROWPROG_MAX];
    // a launch may cover a subset of the tiles (overlap of the halo exchange, mk_comm.hip): `tiles` lists them
    // (null: all tiles 0 .. ntl-1), `poff` is where this launch's partial sums start, `part` 0 = whole product,
    // 1 = first of two launches, 2 = second (repeats the gate's decision without its side effects)
    const int32_t *tiles;
    int64_t ntl;
    int poff;
    int part;
};

// Working sets that fit the 256 MiB Infinity Cache profit from XCD-local tile ranges (every x line is then
// fetched by one L2 instead of by up to five); beyond that size eight distant sweeps cost more in DRAM
// locality than they save (measured: 2-D n=1e6 +5 %, 3-D 512^3 -8 %), so large problems sweep in one front.
static inline int mk_xcd_chunks(const mk_csr *A) {
    const int64_t bytes = 12 * A->nnz + 4 * (A->nrows + 1) + 8 * (A->x_len() + A->nrows) * 3;
    return bytes <= (int64_t)200 * 1024 * 1024 ? 1 : 0;
}

// SpMV grid: twice as many (smaller-share) workgroups pay off only while the problem is cache resident
static inline int mk_grid_spmv_for(const mk_csr *A) {
    int g = mk_grid_spmv(A->ntiles);
    if (!getenv("MK_GRID_SPMV") && mk_xcd_chunks(A)) {
        const int64_t cap = 2 * (int64_t)mk_cap_spmv() > MK_MAXP ? MK_MAXP : 2 * mk_cap_spmv();
        g = (int)(A->ntiles > cap ? cap : (A->ntiles < 1 ? 1 : A->ntiles));
        if (g >= 8) g -= g % 8;
    }
    return g;
}

static inline MkCsrView mk_view(const mk_csr *A) {
    MkCsrView v{A->d_indptr, A->d_indices, A->d_data, A->nrows, A->ntiles, mk_xcd_chunks(A), A->nops, {},
                nullptr, A->ntiles, 0, 0};
    for (int k = 0; k < A->nops; ++k) v.ops[k] = A->ops[k];
    return v;
}

// view of the interior (part 1) or boundary (part 2) tiles of a partitioned matrix; poff2 = grid of part 1
static inline MkCsrView mk_view_part(const mk_csr *A, int part, int poff2) {
    MkCsrView v = mk_view(A);
    v.tiles = A->ex.d_tiles + (part == 2 ? A->ex.n_int : 0);
    v.ntl = (part == 2) ? A->ex.n_bnd : A->ex.n_int;
    v.poff = (part == 2) ? poff2 : 0;
    v.part = part;
    v.xcd_chunks = 0;
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
// CSR-stream SpMV.  A workgroup owns 256 consecutive rows per tile.  Pass 1: all lanes walk
// the tile's nonzeros in storage order, four per lane -- `indices` with one and `data` with two
// 16-byte loads per lane, fully coalesced (the chunk starts at a multiple of four nonzeros so that
// the accesses are naturally aligned; at most three entries of the previous tile are read and ignored); x is
// gathered through L1/L2; the PRODUCTS go to LDS.  Pass 2: lane t owns row t and adds its LDS
// segment left to right, so the per-row rounding sequence is that of a scalar CSR loop
// (bit-identical to the oracle).  Rows longer than the LDS tile are handled by looping over
// chunks with the running sum kept in a register.
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

template <bool PROG, class Epi, int NACC>
__device__ __forceinline__ void mk_spmv_tiles(const MkCsrView &A, const double *__restrict__ x, Epi &epi,
                                              double *prod, double (&acc)[NACC]) {
    constexpr int QUADS = MK_SPMV_TILE / (4 * MK_BLOCK);   // 2 groups of 4 nonzeros per lane per chunk
    const int tid = threadIdx.x;
    // XCD-aware tile order (optional).  Workgroup b is dispatched to XCD b % 8 (observed,
    // MI355X_MICROARCH.md) and each XCD has its own 4 MiB L2.  Placement only affects speed.
    const int G = gridDim.x;
    const int nxcd = (A.xcd_chunks && G % 8 == 0) ? 8 : 1;
    const int per_xcd = G / nxcd;
    const int64_t chunk = (A.ntl + nxcd - 1) / nxcd;
    const int64_t chunk0 = (int64_t)(blockIdx.x % nxcd) * chunk;
    const int64_t chunk_end = (chunk0 + chunk < A.ntl) ? chunk0 + chunk : A.ntl;

    // Row pointers of a tile; fetched one tile ahead so that their latency is not on the critical path.
    // (one load per lane: a row's end is its neighbour's start and travels through LDS, see below)
    struct Meta {
        int p_lo, p_hi, my_lo;
    };
    __shared__ int sptr[MK_BLOCK + 1];
    auto load_meta = [&](int64_t pos, Meta &m) {            // pos: position in the tile list of this launch
        m.p_lo = m.p_hi = m.my_lo = 0;
        if (pos < chunk_end) {
            const int64_t tile = A.tiles ? (int64_t)A.tiles[pos] : pos;
            const int64_t r0 = tile * MK_ROWS_PER_TILE;
            const int64_t rend = (r0 + MK_ROWS_PER_TILE < A.nrows) ? r0 + MK_ROWS_PER_TILE : A.nrows;
            const int64_t r = r0 + tid;
            m.p_lo = A.indptr[r0];
            m.p_hi = A.indptr[rend];
            m.my_lo = A.indptr[(r < rend) ? r : rend];      // rows past the end start (and end) at p_hi
        }
    };
    int64_t pos = chunk0 + blockIdx.x / nxcd;
    Meta cur, nxt;
    load_meta(pos, cur);
    for (; pos < chunk_end; pos += per_xcd) {
        const int64_t tile = A.tiles ? (int64_t)A.tiles[pos] : pos;
        const int64_t r0 = tile * MK_ROWS_PER_TILE;
        const int64_t rend = (r0 + MK_ROWS_PER_TILE < A.nrows) ? r0 + MK_ROWS_PER_TILE : A.nrows;
        const int64_t r = r0 + tid;
        if constexpr (MkHasPre<Epi>::value) {
            if (r < rend) epi.pre(r);
        }
        const int p_lo = cur.p_lo, p_hi = cur.p_hi, my_lo = cur.my_lo;
        int my_hi = p_hi;
        double sum = 0.0;
        bool first = true;
        for (int base = p_lo & ~3; base < p_hi; base += MK_SPMV_TILE) {
            const int cnt = (p_hi - base < MK_SPMV_TILE) ? p_hi - base : MK_SPMV_TILE;
            // ---- pass 1: coalesced stream of the chunk, products into LDS.  Four nonzeros per lane and step: ONE
            // 16-byte index load, two 16-byte value loads, four gathers -- the texture-address unit, not HBM, is
            // the busiest resource of this kernel (~30 clocks per wave-level memory instruction whatever its
            // width; tools/ubench/l1_bench.hip), so the instruction count is what is minimised.  Straight-line
            // code: loads are clamped instead of predicated and every lane stores its (possibly unused)
            // products, so that the compiler keeps all loads of the chunk in flight together.
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
            if (first) {
                load_meta(pos + per_xcd, nxt);              // next tile's row pointers go in flight now
                first = false;
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
        if (first) load_meta(pos + per_xcd, nxt);           // empty tile: still advance the prefetch
        if constexpr (PROG) {                                // composed operators only (separate instantiation)
            if (r < rend) sum = mk_rowprog(A, sum, x, r, epi);
        }
        if (r < rend) epi.row(r, sum, acc);
        cur = nxt;
    }
}

// Gate: a test on a global sum that must be settled before the product may start (e.g. the loop
// condition on ||r||).  Every workgroup evaluates it identically from the previous kernel's partial
// sums; `open` returns whether to run the product and may request a halt through *stop.
struct MkNoGate {
    __device__ bool open(double *, bool, bool *) { return true; }
};

template <class Epi, class Gate, bool PROG>
__global__ __launch_bounds__(MK_BLOCK) void mk_spmv_kernel(MkCsrView A, const double *__restrict__ x, Epi epi,
                                                           Gate gate, MkHalt halt, double *__restrict__ partials) {
    __shared__ double prod[MK_SPMV_TILE];
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
    if (!go) return;
    epi.prologue(s4);
    double acc[Epi::NACC > 0 ? Epi::NACC : 1];
#pragma unroll
    for (int d = 0; d < (Epi::NACC > 0 ? Epi::NACC : 1); ++d) acc[d] = 0.0;
    mk_spmv_tiles<PROG>(A, x, epi, prod, acc);
#pragma unroll
    for (int d = 0; d < Epi::NACC; ++d) {
        const double tot = mk_block_sum(acc[d], s4);
        if (threadIdx.x == 0) partials[(Epi::SLOT0 + d) * MK_MAXP + A.poff + blockIdx.x] = tot;
    }
    if (A.part != 1) halt.template clear_tail<Epi::NACC, Epi::SLOT0>(partials, A.poff + (int)gridDim.x);
}

// Launch the instantiation that matches the operator: plain matrices never pay for the row program.
template <class Epi, class Gate>
static inline void mk_spmv_launch_view(const MkCsrView &v, int grid, hipStream_t st, const double *x, const Epi &epi,
                                       const Gate &gate, MkHalt halt, double *partials) {
    if (v.nops > 0)
        hipLaunchKernelGGL((mk_spmv_kernel<Epi, Gate, true>), dim3(grid), dim3(MK_BLOCK), 0, st, v, x, epi, gate, halt,
                           partials);
    else
        hipLaunchKernelGGL((mk_spmv_kernel<Epi, Gate, false>), dim3(grid), dim3(MK_BLOCK), 0, st, v, x, epi, gate, halt,
                           partials);
}

template <class Epi, class Gate>
static inline void mk_spmv_launch(const mk_csr *A, int grid, hipStream_t st, const double *x, const Epi &epi,
                                  const Gate &gate, MkHalt halt, double *partials) {
    mk_spmv_launch_view(mk_view(A), grid, st, x, epi, gate, halt, partials);
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
    const bool halted = halt.in();
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
