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
//   1 (fmt 7)  per ROW one pattern byte (the row's x positions relative to its lane), values streamed: 8 B per nonzero
//              + 1 B per row.  Variable-coefficient stencils of any width <= 32.
//   2 (fmt 8)  per ROW one pattern byte, a pattern's entries carry {offset, value}: 1 B per row.  Constant-coefficient
//              stencils (HPCG's operator: 27 entries per row from 2 distinct values).
//
// Layouts (mk_format.hip): values of tile T at sval + 256 * start_T, entry k of row t at [(k >> 1) * 512 + 2 t + (k & 1)]
// (pairs: one 16-byte load per lane) and, for an odd width, the last column at [(w - 1) * 256 + t]; slots at
// sslot + 1024 * sstart_T, entry k of row t at [(k >> 2) * 1024 + 4 t + (k & 3)] (8-byte loads).
// LDS: 256 zeros FIRST, then the windows -- lane t's zero cell sits 256 doubles below its own cell whatever the
// matrix, so padding needs no mask anywhere: padded pattern entries carry the relative slot -256, padded slots the
// lane's number (real slots are stored + 256), padded values are +0.0.  A padded product is +-0.0 * 0.0 and leaves a
// running sum that started at +0.0 unchanged, bit for bit.
//
// Pattern tables.  A tile's rows almost all follow ONE pattern, so a wave walks the distinct pattern numbers among its
// lanes (one, or two in a wave that holds a boundary row) and takes a pattern's entries through the SCALAR cache --
// mode 2: {byte offset from the lane's cell, value}, per entry one address add, one LDS read, one multiply with a
// scalar operand, one add; mode 1: the offsets alone (two rounds, then lanes that are left read a word table in LDS:
// waves whose rows follow many patterns must not pay a round each).  (The first version of mode 2 read {slot, code}
// words from an LDS table per lane and picked the value from a second table: 15 vector instructions per entry, 323 us
// for the 256^3 HPCG operator against 200 now)
// fmt 8: a pattern entry as mk_format.hip builds it -- {byte offset of the x value from the lane's own cell, 0, value} --
// is read as four ints, four entries as sixteen (one s_load_dwordx16)
typedef int mk_i16 __attribute__((ext_vector_type(16)));

// DICT: the instantiation for mode 2, which holds no values in registers (twice the occupancy of the streaming one).
// NT: values and slots, read once per product, are loaded non-temporally (mk_stream_nt: matrices beyond the caches).
template <bool DICT, bool NT, bool PROG, class Epi, int NACC>
__device__ __forceinline__ void mk_spmv_tiles_wide(const MkCsrView &A, const double *__restrict__ x, Epi &epi,
        double *prod, double *smem, double (&acc)[NACC]) {
    const int tid = threadIdx.x;
    const MkTileRange trange = mk_tile_range(A);
    int64_t pos = trange.pos;
    const int64_t stride = trange.stride, end = trange.end;
    __shared__ int sptr[MK_BLOCK + 1];
    __shared__ int splen[DICT ? 1 : 256];
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mode = DICT ? 2 : (A.wmode & 1);               // (kernel uniform, like everything read from A)
    const int cpw = A.wper;                                  // window chunks per wave: 4 (16 per tile) or 8 (32 per tile)
    // 256 zeros | windows (or the gather path's products of a tile without windows) | pattern words (mode 1)
    double *zeros = smem;
    double *xw = smem + MK_BLOCK;
    const int wtop = 128 * A.wchunks + 2;
    uint32_t *wtab = reinterpret_cast<uint32_t *>(xw + ((!A.allwin && wtop < MK_PROD_LDS) ? MK_PROD_LDS : wtop));
    zeros[tid] = 0.0;
    if constexpr (!DICT) {
        if (mode == 1) {
            for (int e = tid; e < A.npat * A.pmax; e += MK_BLOCK) wtab[e] = A.pat[e];
            splen[tid] = (tid < A.npat) ? ((int)A.plen[tid] | ((int)A.plen[256 + tid] << 8)) : (255 << 8);
        }
    }
    const mk_i4 *ptab = reinterpret_cast<const mk_i4 *>(A.ptab);
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
            int id = 0;
            if (mode >= 1) id = (r < rend) ? (int)A.pid[r] : 0;                 // one byte per row (loading it one tile
                                                                                // ahead was tried: no change, 203 us)
            [[maybe_unused]] double v[DICT ? 1 : 32];
            [[maybe_unused]] mk_u2 sl[DICT ? 1 : 8];
            if constexpr (!DICT) {                           // the tile's values: (w >> 1) 16-byte loads + one 8-byte load
                const double *vb = A.sval + (int64_t)dcur.sd.x * MK_ROWS_PER_TILE;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
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
                if (mode == 0) {                             // ... and their LDS slots, four to a load
                    const uint16_t *sb = A.sslot + (int64_t)dcur.sd.z * 1024;
#pragma unroll
                    for (int h = 0; h < 8; ++h) {
                        sl[h] = mk_u2{0u, 0u};
                        if (4 * h < w)
                            sl[h] = NT ? __builtin_nontemporal_load(reinterpret_cast<const mk_u2 *>(sb + h * 1024 + 4 * tid))
                                       : *reinterpret_cast<const mk_u2 *>(sb + h * 1024 + 4 * tid);
                    }
                }
            }
            load_desc(pos + stride, dnxt);                   // next tile's descriptors go in flight
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();

            const char *cell = reinterpret_cast<const char *>(xw + tid);        // this lane's own cell
            if constexpr (DICT) {
                int kdiag = 255, len_mine = 0;
                bool pend = true;
                while (pend) {                               // one round per distinct pattern among the wave's lanes
                    const int id0 = __builtin_amdgcn_readfirstlane(id);
                    if (id == id0) {
                        // (inside this branch the compiler knows id == id0 and would address the table with the
                        // per-lane id -- vector loads; read the scalar again from the lanes that are left)
                        const int idu = __builtin_amdgcn_readfirstlane(id);
                        const int info = mk_sload(A.pinfo + idu);
                        const int len0 = info & 0xff;
                        const mk_i16 *pq = reinterpret_cast<const mk_i16 *>(ptab + (int64_t)idu * A.pmax);
                        for (int k0 = 0; k0 < len0; k0 += 8) {   // (eight entries per round: both loads in flight together)
                            const mk_i16 qa = mk_sload(pq + (k0 >> 2)), qb = mk_sload(pq + (k0 >> 2) + 1);
                            const int qo[8] = {qa.s0, qa.s4, qa.s8, qa.sc, qb.s0, qb.s4, qb.s8, qb.sc};
                            const double qv[8] = {__hiloint2double(qa.s3, qa.s2), __hiloint2double(qa.s7, qa.s6),
                                                  __hiloint2double(qa.sb, qa.sa), __hiloint2double(qa.sf, qa.se),
                                                  __hiloint2double(qb.s3, qb.s2), __hiloint2double(qb.s7, qb.s6),
                                                  __hiloint2double(qb.sb, qb.sa), __hiloint2double(qb.sf, qb.se)};
                            double xk[8];
#pragma unroll
                            for (int j = 0; j < 8; ++j) xk[j] = epi.xin(*reinterpret_cast<const double *>(cell + qo[j]));
#pragma unroll
                            for (int j = 0; j < 8; ++j) sum += qv[j] * xk[j];
                        }
                        kdiag = info >> 8;
                        len_mine = len0;
                        if constexpr (ROWX) {
                            if (kdiag < len0) {
                                const mk_i4 d = mk_sload(ptab + (int64_t)idu * A.pmax + kdiag);
                                xr_cur = epi.xin(*reinterpret_cast<const double *>(cell + d.x));
                            }
                        }
                        pend = false;
                    }
                }
                if constexpr (ROWX) {
                    if (kdiag >= len_mine && r < rend) xr_cur = epi.xin(x[r]);
                }
            } else if (mode == 0) {
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    if (4 * h < w) {                         // (tile uniform)
                        const unsigned s4[4] = {sl[h].x & 0xffffu, sl[h].x >> 16, sl[h].y & 0xffffu, sl[h].y >> 16};
                        double xk[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) xk[j] = epi.xin(zeros[s4[j]]);
#pragma unroll
                        for (int j = 0; j < 4; ++j) sum += v[4 * h + j] * xk[j];
                    }
                }
                if constexpr (ROWX) {
                    if (r < rend) xr_cur = epi.xin(x[r]);
                }
            } else {
                // Rows of a tile mostly share one pattern: up to two rounds take a pattern's offsets through the scalar
                // cache for all lanes that follow it (one address add per entry); lanes left after that -- waves whose
                // rows follow many patterns -- read their words from the LDS table.
                const int *poff = reinterpret_cast<const int *>(A.ptab);
                const int pstride = A.pmax < 16 ? 16 : A.pmax;           // (64-byte rows: one s_load_dwordx16 each)
                int kdiag = 255, len = 0;
                bool pend = true;
#pragma unroll 1
                for (int round = 0; round < 2 && __builtin_amdgcn_ballot_w64(pend) != 0; ++round) {
                    if (pend) {
                        const int id0 = __builtin_amdgcn_readfirstlane(id);
                        if (id == id0) {
                            const int idu = __builtin_amdgcn_readfirstlane(id);     // (see the dictionary mode above)
                            const int info = mk_sload(A.pinfo + idu);
                            const mk_i16 *po = reinterpret_cast<const mk_i16 *>(poff + (int64_t)idu * pstride);
                            const mk_i16 oa = mk_sload(po);
                            mk_i16 ob = oa;
                            if (w > 16) ob = mk_sload(po + 1);
#pragma unroll
                            for (int h = 0; h < 8; ++h) {
                                if (4 * h < w) {             // (tile uniform; entries past the row's end: zero cell, +0.0 values)
                                    double xk[4];
#pragma unroll
                                    for (int j = 0; j < 4; ++j) {
                                        const int o = (h < 4) ? oa[(4 * h + j) & 15] : ob[(4 * h + j) & 15];
                                        xk[j] = epi.xin(*reinterpret_cast<const double *>(cell + o));
                                    }
#pragma unroll
                                    for (int j = 0; j < 4; ++j) sum += v[4 * h + j] * xk[j];
                                }
                            }
                            len = info & 0xff;
                            kdiag = info >> 8;
                            if constexpr (ROWX) {
                                if (kdiag < len) xr_cur = epi.xin(*reinterpret_cast<const double *>(cell + mk_sload(poff + (int64_t)idu * pstride + kdiag)));
                            }
                            pend = false;
                        }
                    }
                }
                if (pend) {
                    const int lo = id * A.pmax;
                    len = splen[id] & 0xff;
                    kdiag = splen[id] >> 8;
                    const uint32_t *pw = wtab + lo;
#pragma unroll
                    for (int h = 0; h < 8; ++h) {
                        if (4 * h < w) {
                            const mk_u4 wd = *reinterpret_cast<const mk_u4 *>(pw + 4 * h);
                            const unsigned w4[4] = {wd.x, wd.y, wd.z, wd.w};
                            double xk[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                xk[j] = epi.xin(*reinterpret_cast<const double *>(cell + 8 * (int)(short)(w4[j] & 0xffffu)));
#pragma unroll
                            for (int j = 0; j < 4; ++j) sum += v[4 * h + j] * xk[j];
                        }
                    }
                    if constexpr (ROWX) {
                        if (kdiag < len) xr_cur = epi.xin(*reinterpret_cast<const double *>(cell + 8 * (int)(short)(pw[kdiag] & 0xffffu)));
                    }
                }
                if constexpr (ROWX) {                        // (rows without a diagonal entry: x[r] from memory)
                    if (kdiag >= len && r < rend) xr_cur = epi.xin(x[r]);
                }
            }
            __syncthreads();                                 // the next tile's copies overwrite this LDS
        } else {
            if constexpr (ROWX) {
                if (r < rend) xr_cur = epi.xin(x[r]);
            }
            mk_load_meta(A, pos, end, cur);                  // (tiles without windows are rare: their row pointers now)
            load_desc(pos + stride, dnxt);
            sum = mk_tile_gather(A, x, epi, xw, sptr, cur);  // (products staged where the windows would be)
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
