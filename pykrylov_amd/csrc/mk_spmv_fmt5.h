// mk_spmv_fmt5.h -- windowed tiles + row patterns + streamed values (variable-coefficient stencils)
#pragma once
// (included by mk_device.h: one SpMV tile loop per storage format behind the same Epi / Gate / row_x interface)
//
// fmt 5.  The pattern idea of fmt 4 for matrices whose VALUES are all different (no dictionary): a row is still
// described by one byte -- the number of its pattern, here the sequence of its x positions relative to the lane,
// {slot - t} -- but the values are streamed, 8 bytes per nonzero, from a copy of `data` laid out for this kernel
// (mk_format.hip, "sliced ELL" over the 256-row tiles): tile T owns 256 * w_T doubles, w_T = longest row of the tile;
// entry k of row t sits at [(k >> 1) * 512 + 2 t + (k & 1)] (pairs: one 16-byte load per lane, 1 KiB per wave
// instruction) and, for an odd w_T, the last column at [(w_T - 1) * 256 + t]; rows shorter than w_T are padded with
// +0.0, whose pattern entries point at the lane's zero cell (w_T <= 8: a windowed tile holds at most 2048 nonzeros).  What a tile ingests: its x windows (global_load_lds),
// 256 pattern bytes and 256 * w_T values straight into registers -- no column indices, no row pointers, no slots, no
// product staging: 8 B per nonzero + 1 B per row instead of CSR's 12 B per nonzero + 4 B per row.  The row phase is
// fmt 4's: lane t walks row t LEFT TO RIGHT (same products, same order, same bits as every other format).
template <bool PROG, bool NT, class Epi, int NACC>
__device__ __forceinline__ void mk_spmv_tiles_fmt5(const MkCsrView &A, const double *__restrict__ x, Epi &epi,
        double *prod, double *xw, double (&acc)[NACC]) {
    const int tid = threadIdx.x;
    const MkTileRange trange = mk_tile_range(A);
    int64_t pos = trange.pos;
    const int64_t stride = trange.stride, end = trange.end;
    __shared__ int sptr[MK_BLOCK + 1];
    __shared__ int splen[256];
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // behind the windows (and whatever the gather path of a tile without windows may overwrite): 256 zeros, then the
    // pattern table -- per entry the byte offset of its x value relative to the lane's own cell
    const int wtop = 128 * A.wchunks + 2;
    double *zeros = xw + ((!A.allwin && wtop < MK_PROD_LDS) ? MK_PROD_LDS : wtop);
    int *otab = reinterpret_cast<int *>(zeros + MK_BLOCK);
    const int zoff = (int)((zeros - xw) * (int)sizeof(double));
    zeros[tid] = 0.0;
    for (int e = tid; e < A.npat * A.pmax; e += MK_BLOCK) {
        const int pnum = e / A.pmax, k = e - pnum * A.pmax;
        otab[e] = (k < (int)A.plen[pnum]) ? 8 * (int)(short)(A.pat[e] & 0xffffu) : zoff;
    }
    splen[tid] = (tid < A.npat) ? ((int)A.plen[tid] | ((int)A.plen[256 + tid] << 8)) : (255 << 8);
    // this wave's window descriptor and the tile's value block (scalar loads, issued one tile ahead)
    struct Desc {
        mk_i4 g;
        unsigned nvw;
        mk_i2 sd;                                            // {start of the value block / 256, w_T}
    };
    auto load_desc = [&](int64_t p, Desc &d) {
        d.g = mk_i4{0, 0, 0, 0};
        d.nvw = 0;
        d.sd = mk_i2{0, 0};
        if (p < end) {
            const int64_t t = mk_tile_at(A, p);
            d.g = mk_sload(reinterpret_cast<const mk_i4 *>(A.wg + (t * 4 + wv) * 4));
            d.nvw = mk_sload(A.wn + t * 4 + wv);
            d.sd = mk_sload(reinterpret_cast<const mk_i2 *>(A.sdesc + 2 * t));
        }
    };
    // a row's values: registers, (w >> 1) 16-byte loads + one 8-byte load for an odd width (w is tile uniform: scalar
    // branches).  NT: non-temporal loads -- the values are read once per product and should not push the x windows out
    // of the L2 (with tile order 4 the fabric reads of a 512^3 product fall from 10.4 to 9.3 GB, 1.06 x the format's
    // bytes; the kernel is not faster for it, profiles/r03_*).  Prefetching the next tile's values into a second set of
    // registers while the rows are walked was tried too (87 registers, 5 workgroups per CU): +1 %, not kept.
    auto load_vals = [&](const mk_i2 sd, double (&v)[8]) {
        const int w = sd.y;
        const double *vb = A.sval + (int64_t)sd.x * MK_ROWS_PER_TILE;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v[2 * q] = 0.0;
            v[2 * q + 1] = 0.0;
            if (2 * q + 1 < w) {
                const mk_d2 pr = NT ? __builtin_nontemporal_load(reinterpret_cast<const mk_d2 *>(vb + q * 512 + 2 * tid))
                                    : *reinterpret_cast<const mk_d2 *>(vb + q * 512 + 2 * tid);
                v[2 * q] = pr.x;
                v[2 * q + 1] = pr.y;
            } else if (2 * q < w) {
                v[2 * q] = NT ? __builtin_nontemporal_load(vb + 2 * q * 256 + tid) : vb[2 * q * 256 + tid];
            }
        }
    };
    MkTileMeta cur;
    Desc dcur, dnxt;
    load_desc(pos, dcur);

    for (; pos < end; pos += stride) {
        const int64_t tile = mk_tile_at(A, pos);
        const int64_t r0 = tile * MK_ROWS_PER_TILE;
        const int64_t rend = (r0 + MK_ROWS_PER_TILE < A.nrows) ? r0 + MK_ROWS_PER_TILE : A.nrows;
        const int64_t r = r0 + tid;
        constexpr bool ROWX = !PROG && MkHasRowX<Epi>::value;
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
            const unsigned id = (r < rend) ? (unsigned)A.pid[r] : 0u;       // one byte per row
            double v[8];
            load_vals(dcur.sd, v);
            load_desc(pos + stride, dnxt);                   // next tile's descriptors go in flight
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();

            const int lo = (int)id * A.pmax;
            const int len = (r < rend) ? (splen[id] & 0xff) : 0;
            const int kdiag = splen[id] >> 8;
            const char *cell = reinterpret_cast<const char *>(xw + tid);        // this lane's own cell
            const int *po = otab + lo;
            const mk_i4 o0 = *reinterpret_cast<const mk_i4 *>(po), o1 = *reinterpret_cast<const mk_i4 *>(po + 4);
            const int off[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
            double xk[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) xk[k] = epi.xin(*reinterpret_cast<const double *>(cell + off[k]));
#pragma unroll
            for (int k = 0; k < 8; ++k) sum += v[k] * xk[k];
            if constexpr (ROWX) {                            // x[r] for the epilogue: the diagonal entry's cell
                if (kdiag < len) xr_cur = epi.xin(*reinterpret_cast<const double *>(cell + po[kdiag]));
                else if (r < rend) xr_cur = epi.xin(x[r]);
            }
            __syncthreads();                                 // the next tile's copies overwrite this LDS
        } else {
            if constexpr (ROWX) {
                if (r < rend) xr_cur = epi.xin(x[r]);
            }
            mk_load_meta(A, pos, end, cur);                  // (tiles without windows are rare: their row pointers now)
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
        dcur = dnxt;
    }
}
