// mk_spmv_fmtw.h -- "wide" windowed tiles: rows of up to 32 entries, tiles of up to 8192 nonzeros, 32 window chunks
#pragma once
// (included by mk_device.h: one SpMV tile loop per storage format behind the same Epi / Gate / row_x interface)
//
// fmt 6 / 7 / 8.  The row-walk kernels of fmt 4 and fmt 5 keep a row's 8 entries in registers and a tile's windows in
// 16 chunks: enough for 5- and 7-point stencils, not for 9- / 19- / 27-point ones, block-structured problems or meshes
// (a 27-point tile has 6912 nonzeros in 27 chunks).  This kernel is the same design without those limits -- lane t
// walks row t LEFT TO RIGHT against x windows staged in LDS by global_load_lds, no column index, no row pointer, no
// product staging, same products in the same order as every other format -- in three storage modes (A.wmode):
//
//   0 (fmt 6)  per nonzero a uint16 LDS slot and the value, both streamed in tile-sliced ELL order: 10 B per nonzero.
//              Any matrix whose tiles have column sets that fit 32 chunks; no pattern, no dictionary needed.
//   1 (fmt 7)  per ROW one pattern byte (the row's x positions relative to its lane, table in LDS), values streamed:
//              8 B per nonzero + 1 B per row.  Variable-coefficient stencils of any width <= 32.
//   2 (fmt 8)  per ROW one pattern byte, the pattern's words carry {relative slot, value code}: 1 B per row.
//              Constant-coefficient stencils (HPCG's operator: 27 entries per row from 2 distinct values).
//
// Layouts (mk_format.hip): values of tile T at sval + 256 * start_T, entry k of row t at [(k >> 1) * 512 + 2 t + (k & 1)]
// (pairs: one 16-byte load per lane) and, for an odd width, the last column at [(w - 1) * 256 + t]; slots at
// sslot + 1024 * sstart_T, entry k of row t at [(k >> 2) * 1024 + 4 t + (k & 3)] (8-byte loads), 0xffff = padding.
// Padding multiplies +0.0 (or a masked value) with the lane's own zero cell in LDS: the product is +-0.0 and leaves a
// running sum that started at +0.0 unchanged, bit for bit.
// DICT: the instantiation for mode 2, which holds no values in registers (twice the occupancy of the streaming one).
template <bool DICT, bool PROG, class Epi, int NACC>
__device__ __forceinline__ void mk_spmv_tiles_wide(const MkCsrView &A, const double *__restrict__ x, Epi &epi,
        double *prod, double *xw, double (&acc)[NACC]) {
    const int tid = threadIdx.x;
    const MkTileRange trange = mk_tile_range(A);
    int64_t pos = trange.pos;
    const int64_t stride = trange.stride, end = trange.end;
    __shared__ int sptr[MK_BLOCK + 1];
    __shared__ int splen[256];
    __shared__ double sdict[256];
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mode = DICT ? 2 : (A.wmode & 1);               // (kernel uniform, like everything read from A)
    const int cpw = A.wper;                                  // window chunks per wave: 4 (16 per tile) or 8 (32 per tile)
    // behind the windows (and whatever the gather path of a tile without windows may overwrite): 256 zeros, then the
    // pattern table as the builder stores it -- per entry {slot - lane : 16 | value code : 8}
    const int wtop = 128 * A.wchunks + 2;
    double *zeros = xw + ((!A.allwin && wtop < MK_PROD_LDS) ? MK_PROD_LDS : wtop);
    uint32_t *wtab = reinterpret_cast<uint32_t *>(zeros + MK_BLOCK);
    const int zcell = (int)(zeros - xw) + tid;               // this lane's zero cell (index into xw)
    zeros[tid] = 0.0;
    if (mode >= 1) {
        for (int e = tid; e < A.npat * A.pmax; e += MK_BLOCK) wtab[e] = A.pat[e];
        splen[tid] = (tid < A.npat) ? ((int)A.plen[tid] | ((int)A.plen[256 + tid] << 8)) : (255 << 8);
    }
    sdict[tid] = (mode == 2 && tid < A.ndict) ? A.dict[tid] : 0.0;
    const bool two = (A.ndict <= 2);                         // value picked in registers instead of read from LDS
    const double d0 = (mode == 2) ? A.dict[0] : 0.0, d1 = (mode == 2) ? A.dict[A.ndict > 1 ? 1 : 0] : 0.0;
    // this wave's window descriptor and the tile's blocks (scalar loads, issued one tile ahead)
    struct Desc {
        mk_i4 g0, g1;
        mk_i2 n;
        mk_i4 sd;                                            // {start of the value block / 256, width, start of the slot block / 1024, -}
    };
    auto load_desc = [&](int64_t p, Desc &d) {
        d.g0 = d.g1 = d.sd = mk_i4{0, 0, 0, 0};
        d.n = mk_i2{0, 0};
        if (p < end) {
            const int64_t t = mk_tile_at(A, p);
            const int32_t *gp = A.wg + (t * 4 + wv) * cpw;
            const uint32_t *np = A.wn + (t * 4 + wv) * (cpw >> 2);
            d.g0 = mk_sload(reinterpret_cast<const mk_i4 *>(gp));
            d.n.x = (int)mk_sload(np);
            if (cpw == 8) {
                d.g1 = mk_sload(reinterpret_cast<const mk_i4 *>(gp + 4));
                d.n.y = (int)mk_sload(np + 1);
            }
            if (mode <= 1) d.sd = mk_sload(reinterpret_cast<const mk_i4 *>(A.sdesc + 4 * t));
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
        double sum = 0.0;
        [[maybe_unused]] double xr_cur = 0.0;
        if (dcur.g0.x & 1) {
            const int gs[8] = {dcur.g0.x & ~1, dcur.g0.y, dcur.g0.z, dcur.g0.w, dcur.g1.x, dcur.g1.y, dcur.g1.z, dcur.g1.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int hc = (i < cpw) ? (int)(((unsigned)(i < 4 ? dcur.n.x : dcur.n.y) >> (8 * (i & 3))) & 0xffu) : 0;
                if (hc > 0) {
                    const int l2 = (lane < hc) ? lane : hc - 1;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(x + gs[i] + 2 * l2),
                                                     (__attribute__((address_space(3))) void *)(xw + (wv + 4 * i) * 128),
                                                     16, 0, 0);
                }
            }
            const int w = dcur.sd.y;
            unsigned id = 0;
            if (mode >= 1) id = (r < rend) ? (unsigned)A.pid[r] : 0u;           // one byte per row
            [[maybe_unused]] double v[DICT ? 1 : 32];
            [[maybe_unused]] mk_u2 sl[DICT ? 1 : 8];
            if constexpr (!DICT) {
            if (mode <= 1) {                                 // the tile's values: (w >> 1) 16-byte loads + one 8-byte load
                const double *vb = A.sval + (int64_t)dcur.sd.x * MK_ROWS_PER_TILE;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    v[2 * q] = 0.0;
                    v[2 * q + 1] = 0.0;
                    if (2 * q + 1 < w) {
                        const mk_d2 pr = *reinterpret_cast<const mk_d2 *>(vb + q * 512 + 2 * tid);
                        v[2 * q] = pr.x;
                        v[2 * q + 1] = pr.y;
                    } else if (2 * q < w) {
                        v[2 * q] = vb[2 * q * 256 + tid];
                    }
                }
            }
            if (mode == 0) {                                 // ... and their LDS slots, four to a load
                const uint16_t *sb = A.sslot + (int64_t)dcur.sd.z * 1024;
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    sl[h] = mk_u2{0xffffffffu, 0xffffffffu};
                    if (4 * h < w) sl[h] = *reinterpret_cast<const mk_u2 *>(sb + h * 1024 + 4 * tid);
                }
            }
            }
            load_desc(pos + stride, dnxt);                   // next tile's descriptors go in flight
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();

            if (!DICT && mode == 0) {
              if constexpr (!DICT) {
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    if (4 * h < w) {                         // (tile uniform)
                        const unsigned s4[4] = {sl[h].x & 0xffffu, sl[h].x >> 16, sl[h].y & 0xffffu, sl[h].y >> 16};
                        double xk[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) xk[j] = epi.xin(xw[(s4[j] == 0xffffu) ? zcell : (int)s4[j]]);
#pragma unroll
                        for (int j = 0; j < 4; ++j) sum += v[4 * h + j] * xk[j];
                    }
                }
              }
                if constexpr (ROWX) {
                    if (r < rend) xr_cur = epi.xin(x[r]);
                }
            } else {
                const int lo = (int)id * A.pmax;
                const int len = (r < rend) ? (splen[id] & 0xff) : 0;
                const int kdiag = splen[id] >> 8;
                const uint32_t *pw = wtab + lo;
                if constexpr (!DICT) {
#pragma unroll
                    for (int h = 0; h < 8; ++h) {
                        if (4 * h < w) {                     // (tile uniform; entries past the row's end meet +0.0 values)
                            const mk_u4 wd = *reinterpret_cast<const mk_u4 *>(pw + 4 * h);
                            const unsigned w4[4] = {wd.x, wd.y, wd.z, wd.w};
                            double xk[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                xk[j] = epi.xin(xw[(4 * h + j < len) ? tid + (int)(short)(w4[j] & 0xffffu) : zcell]);
#pragma unroll
                            for (int j = 0; j < 4; ++j) sum += v[4 * h + j] * xk[j];
                        }
                    }
                } else {
                    for (int k0 = 0; k0 < len; k0 += 4) {    // (per lane: lanes with shorter rows sit out)
                        const mk_u4 wd = *reinterpret_cast<const mk_u4 *>(pw + k0);
                        const unsigned w4[4] = {wd.x, wd.y, wd.z, wd.w};
                        double xk[4], vk[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const bool in = (k0 + j < len);
                            xk[j] = epi.xin(xw[in ? tid + (int)(short)(w4[j] & 0xffffu) : zcell]);
                            const double dv = two ? ((w4[j] >> 16) ? d1 : d0) : sdict[(w4[j] >> 16) & 0xffu];
                            vk[j] = in ? dv : 0.0;
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) sum += vk[j] * xk[j];
                    }
                }
                if constexpr (ROWX) {                        // x[r] for the epilogue: the diagonal entry's cell
                    if (kdiag < len) xr_cur = epi.xin(xw[tid + (int)(short)(pw[kdiag] & 0xffffu)]);
                    else if (r < rend) xr_cur = epi.xin(x[r]);
                }
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
