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
#include "mk_internal.h"

struct MkCsrView {
    const int32_t *indptr;
    const int32_t *indices;
    const double *data;
    int64_t nrows;
    int64_t ntiles;
};

static inline MkCsrView mk_view(const mk_csr *A) {
    return MkCsrView{A->d_indptr, A->d_indices, A->d_data, A->nrows, A->ntiles};
}

#ifdef __HIPCC__

// ---------------------------------------------------------------------------------------
// CSR-stream SpMV.  A workgroup owns 256 consecutive rows per tile.  Pass 1: all lanes walk
// the tile's nonzeros in storage order -- `data`/`indices` are read fully coalesced, x is
// gathered through L1/L2, the PRODUCTS go to LDS.  Pass 2: lane t owns row t and adds its
// LDS segment left to right, so the per-row rounding sequence is that of a scalar CSR loop
// (bit-identical to the oracle).  Rows longer than the LDS tile are handled by looping
// over chunks with the running sum kept in a register.
//
// Epi interface:   double xin(double xj)            value actually multiplied (e.g. s*y[j])
//                  void   row(int64_t r, double s, double *acc)   consume the row result
// ---------------------------------------------------------------------------------------
template <class Epi, int NACC>
__device__ __forceinline__ void mk_spmv_tiles(const MkCsrView &A, const double *__restrict__ x, Epi &epi,
                                              double *prod, double (&acc)[NACC]) {
    constexpr int PER = MK_SPMV_TILE / MK_BLOCK;   // 8 nonzeros per lane per chunk
    const int tid = threadIdx.x;
    for (int64_t tile = blockIdx.x; tile < A.ntiles; tile += gridDim.x) {
        const int64_t r0 = tile * MK_ROWS_PER_TILE;
        const int64_t rend = (r0 + MK_ROWS_PER_TILE < A.nrows) ? r0 + MK_ROWS_PER_TILE : A.nrows;
        const int64_t r = r0 + tid;
        const int p_lo = A.indptr[r0];
        const int p_hi = A.indptr[rend];
        int my_lo = p_hi, my_hi = p_hi;
        if (r < rend) {
            my_lo = A.indptr[r];
            my_hi = A.indptr[r + 1];
        }
        double sum = 0.0;
        for (int base = p_lo; base < p_hi; base += MK_SPMV_TILE) {
            const int cnt = (p_hi - base < MK_SPMV_TILE) ? p_hi - base : MK_SPMV_TILE;
            // ---- pass 1: coalesced stream of the chunk, products into LDS
            int col[PER];
            double val[PER];
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                int j = k * MK_BLOCK + tid;
                j = (j < cnt) ? j : cnt - 1;            // clamp: loads stay unconditional
                col[k] = A.indices[base + j];
                val[k] = A.data[base + j];
            }
            double xv[PER];
#pragma unroll
            for (int k = 0; k < PER; ++k) xv[k] = x[col[k]];
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int j = k * MK_BLOCK + tid;
                if (j < cnt) prod[j] = val[k] * epi.xin(xv[k]);
            }
            __syncthreads();
            // ---- pass 2: one lane per row, left-to-right sum of its segment
            const int lo = ((my_lo > base) ? my_lo : base) - base;
            const int hi = ((my_hi < base + cnt) ? my_hi : base + cnt) - base;
            const int len = hi - lo;
            double t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = (k < len) ? prod[lo + k] : 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k < len) sum += t[k];
            for (int k = 8; k < len; ++k) sum += prod[lo + k];
            __syncthreads();
        }
        if (r < rend) epi.row(r, sum, acc);
    }
}

// Gate: a test on a global sum that must be settled before the product may start (e.g. the loop
// condition on ||r||).  Every workgroup evaluates it identically from the previous kernel's partial
// sums; `open` returns whether to run the product and may request a halt through *stop.
struct MkNoGate {
    __device__ bool open(double *, bool, bool *) { return true; }
};

template <class Epi, class Gate>
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
    const bool go = gate.open(s4, lead, &stop);
    if (lead) halt.out(stop);
    if (!go) return;
    epi.prologue(s4);
    double acc[Epi::NACC > 0 ? Epi::NACC : 1];
#pragma unroll
    for (int d = 0; d < (Epi::NACC > 0 ? Epi::NACC : 1); ++d) acc[d] = 0.0;
    mk_spmv_tiles(A, x, epi, prod, acc);
#pragma unroll
    for (int d = 0; d < Epi::NACC; ++d) {
        const double tot = mk_block_sum(acc[d], s4);
        if (threadIdx.x == 0) partials[(Epi::SLOT0 + d) * MK_MAXP + blockIdx.x] = tot;
    }
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
template <class Op>
__global__ __launch_bounds__(MK_BLOCK) void mk_stream_kernel(Op op, int64_t n, MkHalt halt,
                                                             double *__restrict__ partials) {
    __shared__ double s4[4];
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
    const int64_t S = (int64_t)gridDim.x * MK_BLOCK;
    const int64_t g = (int64_t)blockIdx.x * MK_BLOCK + threadIdx.x;
    const int64_t npair = n >> 1;
    for (int64_t q = g; q < npair; q += S) op.pair(2 * q, acc);
    if ((n & 1) && g == (npair % S)) op.one(n - 1, acc);
#pragma unroll
    for (int d = 0; d < Op::NACC; ++d) {
        const double tot = mk_block_sum(acc[d], s4);
        if (threadIdx.x == 0) partials[(Op::SLOT0 + d) * MK_MAXP + blockIdx.x] = tot;
    }
}

__device__ __forceinline__ double2 mk_ld2(const double *p, int64_t i) {
    return *reinterpret_cast<const double2 *>(p + i);
}
__device__ __forceinline__ void mk_st2(double *p, int64_t i, double2 v) {
    *reinterpret_cast<double2 *>(p + i) = v;
}

#endif  // __HIPCC__
