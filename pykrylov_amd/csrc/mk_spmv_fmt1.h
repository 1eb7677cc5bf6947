// mk_spmv_fmt1.h -- windowed tiles with raw values: x windows in LDS, uint16 slots, products staged through LDS
#pragma once
// (included by mk_device.h: one SpMV tile loop per storage format behind the same Epi / Gate / row_x interface)

template <bool PROG, class Epi, int NACC>
__device__ __forceinline__ void mk_spmv_tiles_fmt1(const MkCsrView &A, const double *__restrict__ x, Epi &epi,
        double *prod, double *xw, double (&acc)[NACC]) {
    const int tid = threadIdx.x;
    const MkTileRange trange = mk_tile_range(A);
    int64_t pos = trange.pos;
    const int64_t stride = trange.stride, end = trange.end;
    __shared__ int sptr[MK_BLOCK + 1];
    auto load_meta = [&](int64_t p, MkTileMeta &m) { mk_load_meta(A, p, end, m); };
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
