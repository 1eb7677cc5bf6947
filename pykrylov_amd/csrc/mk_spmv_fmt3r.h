// mk_spmv_fmt3r.h -- format 3 with a SECOND tile per workgroup whose rows (<= W entries) are held in registers (round 4)
#pragma once
// (included by mk_device.h: one SpMV tile loop per storage format behind the same Epi / Gate / row_x interface)
//
// The column-phase kernel of mk_spmv_fmt3.h needs as many ROUNDS over x as the tiles need at 2048 resident workgroups:
// BASELINE config 3 (3 907 tiles) pulls x through the fabric 2 rounds x 8 XCDs = 16 times (201 MB per product for 84 MB of
// data; the kernel runs at the fabric's rate for that traffic, profiles/r03_pmc_and_trace_summary.txt).  Holding BOTH tiles of
// a workgroup in registers with a static phase loop was priced first (tools/ubench/spmv_cb.hip `reg`: 42-53 us, slower: three
// staged ingest steps per workgroup skew the phases of a CU's workgroups by 5 ... 36 us and the slices of x are no longer
// shared).  This variant keeps the proven structure -- tile A's (column, value) stream in LDS, walked with a cursor -- and
// adds tile B = the tile the workgroup would visit NEXT (pos + stride: same tile-to-workgroup map, same visiting order, so
// the partial sums of fused dots do not change by a bit): B's stream passes through the same LDS buffer first and lane t
// keeps row t's <= W (column, value) pairs in registers; in every phase the gathers of B's entries in the slice are issued
// (exec-masked buffer loads, one 32-bit offset register each) BEFORE the cursor walk of A starts and consumed after it.
// One round per pair of tiles: x crosses the fabric half as often.
template <int W, bool PROG, class Epi, int NACC>
__device__ __forceinline__ void mk_spmv_tiles_fmt3r(const MkCsrView &A, const double *__restrict__ x, Epi &epi,
        double *prod, double *xw, double (&acc)[NACC]) {
    const int tid = threadIdx.x;
    const MkTileRange trange = mk_tile_range(A);
    int64_t pos = trange.pos;
    const int64_t stride = trange.stride, end = trange.end;
    (void)xw;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cap = A.rt_cap;
    double *lv = prod;                                   // [cap] values, then [cap] columns
    int *lc = reinterpret_cast<int *>(prod + cap);
    constexpr unsigned NONE = 0xffffffffu;
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(x), 0, 0x7fffffff, 0x00020000);
    auto ingest = [&](int64_t tile, int &cur, int &fin, double &sum0) {      // the tile's stream -> LDS (DMA), row bounds
        const int64_t r0 = tile * MK_ROWS_PER_TILE;
        const int64_t rend = (r0 + MK_ROWS_PER_TILE < A.nrows) ? r0 + MK_ROWS_PER_TILE : A.nrows;
        const int64_t r = r0 + tid;
        const int p_lo = mk_sload(A.indptr + r0), p_hi = mk_sload(A.indptr + rend);
        const int base = p_lo & ~3, cnt = p_hi - base;   // cnt <= cap (builder)
        const int last = (cnt > 0) ? ((cnt - 1) & ~3) : 0;
        for (int c0 = wv * 256; c0 < cnt; c0 += 4 * 256) {
            int j = c0 + 4 * lane;
            j = j < last ? j : last;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(A.indices + base + j),
                                             (__attribute__((address_space(3))) void *)(lc + c0), 16, 0, 0);
        }
        const int lastv = (cnt > 0) ? ((cnt - 1) & ~1) : 0;
        for (int c0 = wv * 128; c0 < cnt; c0 += 4 * 128) {
            int j = c0 + 2 * lane;
            j = j < lastv ? j : lastv;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(A.data + base + j),
                                             (__attribute__((address_space(3))) void *)(lv + c0), 16, 0, 0);
        }
        cur = fin = 0;
        sum0 = 0.0;
        if (r < rend) {
            cur = A.indptr[r] - base;
            fin = A.indptr[r + 1] - base;
            if (A.sum_in) sum0 = A.sum_in[r];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    pos += (int64_t)A.step0 * 2 * stride;                  // (a product split into one launch per step: mk_spmv_launch_blocks)
    for (int step = 0; pos < end && (A.nsteps == 0 || step < A.nsteps); pos += 2 * stride, ++step) {
        // ---- tile B (the later one) first: through LDS into registers
        const int64_t posb = pos + stride;
        const bool has_b = posb < end;                   // (workgroup uniform)
        unsigned ob[W];
        double vb[W];
        double sumb = 0.0;
#pragma unroll
        for (int j = 0; j < W; ++j) {
            ob[j] = NONE;
            vb[j] = 0.0;
        }
        if (has_b) {
            int cur, fin;
            ingest(mk_tile_at(A, posb), cur, fin, sumb);
#pragma unroll
            for (int j = 0; j < W; ++j) {
                const bool has = cur + j < fin;
                const int idx = has ? cur + j : 0;
                const unsigned cc = (unsigned)lc[idx] << 3;
                const double vv = lv[idx];
                ob[j] = has ? cc : NONE;
                vb[j] = has ? vv : 0.0;
            }
            __syncthreads();                             // tile A's copies overwrite the buffer
        }
        // ---- tile A stays in LDS
        int cur, fin;
        double sum;
        ingest(mk_tile_at(A, pos), cur, fin, sum);
        for (int k = 0; k < A.rt_k; ++k) {
            const int c1 = (k + 1 < A.rt_k) ? (k + 1) * A.rt_w : 0x7fffffff;
            const unsigned o_lo = (unsigned)(k * A.rt_w) << 3;
            const unsigned o_hi = (k + 1 < A.rt_k) ? (unsigned)c1 << 3 : NONE;
            double xb[W];
#pragma unroll
            for (int j = 0; j < W; ++j) {                // B's gathers of this slice: in flight during A's cursor walk
                xb[j] = 0.0;
                if (ob[j] >= o_lo && ob[j] < o_hi) {
                    const mk_u2 w = __builtin_bit_cast(mk_u2, __builtin_amdgcn_raw_buffer_load_b64(xres, (int)ob[j], 0, 0));
                    xb[j] = __builtin_bit_cast(double, w);
                }
            }
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
#pragma unroll
            for (int j = 0; j < W; ++j) {
                const bool in = ob[j] >= o_lo && ob[j] < o_hi;
                const double t = sumb + vb[j] * epi.xin(xb[j]);
                sumb = in ? t : sumb;
            }
        }
        {
            const int64_t r = mk_tile_at(A, pos) * MK_ROWS_PER_TILE + tid;
            if (r < A.nrows) {
                if constexpr (MkHasPre<Epi>::value) epi.pre(r);
                if constexpr (PROG) sum = mk_rowprog(A, sum, x, r, epi);
                epi.row(r, sum, acc);
            }
        }
        if (has_b) {
            const int64_t r = mk_tile_at(A, posb) * MK_ROWS_PER_TILE + tid;
            if (r < A.nrows) {
                if constexpr (MkHasPre<Epi>::value) epi.pre(r);
                if constexpr (PROG) sumb = mk_rowprog(A, sumb, x, r, epi);
                epi.row(r, sumb, acc);
            }
        }
        __syncthreads();                                 // the next pair's copies overwrite this LDS
    }
}
