// mk_spmv_fmt3.h -- plain CSR, tile resident in LDS, gathers ordered by column phase (x longer than an L2)
#pragma once
// (included by mk_device.h: one SpMV tile loop per storage format behind the same Epi / Gate / row_x interface)

template <bool PROG, class Epi, int NACC>
__device__ __forceinline__ void mk_spmv_tiles_fmt3(const MkCsrView &A, const double *__restrict__ x, Epi &epi,
        double *prod, double *xw, double (&acc)[NACC]) {
    const int tid = threadIdx.x;
    const MkTileRange trange = mk_tile_range(A);
    int64_t pos = trange.pos;
    const int64_t stride = trange.stride, end = trange.end;
    (void)xw;
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
            const int c1 = (k + 1 < A.rt_k) ? A.rt_c0 + (k + 1) * A.rt_w : 0x7fffffff;
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
}
