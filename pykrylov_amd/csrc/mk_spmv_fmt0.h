// mk_spmv_fmt0.h -- plain CSR: coalesced stream of (column, value), x gathered through L1/L2 (every tile: mk_tile_gather)
#pragma once
// (included by mk_device.h: one SpMV tile loop per storage format behind the same Epi / Gate / row_x interface)

template <bool PROG, class Epi, int NACC>
__device__ __forceinline__ void mk_spmv_tiles_fmt0(const MkCsrView &A, const double *__restrict__ x, Epi &epi,
        double *prod, double *xw, double (&acc)[NACC]) {
    const int tid = threadIdx.x;
    const MkTileRange trange = mk_tile_range(A);
    int64_t pos = trange.pos;
    const int64_t stride = trange.stride, end = trange.end;
    __shared__ int sptr[MK_BLOCK + 1];
    auto load_meta = [&](int64_t p, MkTileMeta &m) { mk_load_meta(A, p, end, m); };
    (void)xw;
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
}
