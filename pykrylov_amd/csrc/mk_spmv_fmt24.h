// mk_spmv_fmt24.h -- windowed tiles + value dictionary (fmt 2: one packed word per nonzero; fmt 4: one pattern byte per row)
#pragma once
// (included by mk_device.h: one SpMV tile loop per storage format behind the same Epi / Gate / row_x interface)

template <int FMT, bool PROG, class Epi, int NACC>
__device__ __forceinline__ void mk_spmv_tiles_fmt24(const MkCsrView &A, const double *__restrict__ x, Epi &epi,
        double *prod, double *xw, double (&acc)[NACC]) {
    const int tid = threadIdx.x;
    const MkTileRange trange = mk_tile_range(A);
    int64_t pos = trange.pos;
    const int64_t stride = trange.stride, end = trange.end;
    __shared__ int sptr[MK_BLOCK + 1];
    auto load_meta = [&](int64_t p, MkTileMeta &m) { mk_load_meta(A, p, end, m); };
    constexpr bool PAT = (FMT == 4);
    // ---- windowed tiles with a value dictionary: ROW PHASE ONLY.  One 32-bit word per nonzero {slot | code};
    // the words and the x windows of a tile go straight to LDS with global_load_lds (no VGPR round trip, no
    // per-nonzero staging work); after one barrier lane t walks row t left to right: word, x and value from LDS
    // (consecutive rows read consecutive words / x entries: conflict free for the usual odd row lengths).
    // Measured against the product-staging design of fmt 1 with the codes: 512^3 1.50 -> 1.32 ms, 2-D n = 1e6
    // 13.1 -> 9.1 us (tools/ubench/spmv_win2.hip, w3 vs w7).
    // fmt 4 (PAT) goes one step further: a row is described by ONE BYTE, the number of its pattern -- the sequence
    // of its words relative to the lane, {slot - t, code} -- and the pattern table (<= 8 KB) sits in LDS for the
    // whole kernel.  The per-nonzero stream and the row pointers are not read at all: what a tile ingests is its
    // x windows and 256 bytes.
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // fmt 2: a tile's packed words behind its windows.  fmt 4: the pattern table, which lives as long as the kernel
    // and therefore sits behind everything the gather path of a tile without windows may overwrite
    // (unless every tile has windows: then the gather path never runs, A.allwin)
    const int wtop = 128 * A.wchunks + 2;
    uint32_t *spk = reinterpret_cast<uint32_t *>(xw + ((PAT && !A.allwin && wtop < MK_PROD_LDS) ? MK_PROD_LDS : wtop));
    __shared__ double sdict[PAT ? 1 : 256];              // (fmt 4 keeps the values in its pattern table)
    if constexpr (!PAT) sdict[tid] = (tid < A.ndict) ? A.dict[tid] : 0.0;   // (read after a barrier below)
    [[maybe_unused]] __shared__ int splen[PAT ? 256 : 1];
    // fmt 4: behind the windows (and whatever the gather path may overwrite) 256 zeros, then the pattern table in
    // the form the row phase consumes with the fewest instructions -- per entry {byte offset of its x value
    // relative to the lane's own cell, the VALUE itself}: one 16-byte LDS read, one add, one 8-byte LDS read, one
    // multiply, one add.  Entries past a pattern's end point at the lane's zero cell with value +0.0: their
    // product is +-0.0 and leaves the running sum (never -0.0, it started at +0.0) unchanged, so nothing is masked.
    struct PatEntry {
        int off, pad;
        double val;
    };
    [[maybe_unused]] PatEntry *ftab = nullptr;
    [[maybe_unused]] int zoff = 0;
    if constexpr (PAT) {
        double *zeros = reinterpret_cast<double *>(spk);
        ftab = reinterpret_cast<PatEntry *>(zeros + MK_BLOCK);
        zoff = (int)((zeros - xw) * (int)sizeof(double));
        zeros[tid] = 0.0;
        for (int e = tid; e < A.npat * A.pmax; e += MK_BLOCK) {
            const int pnum = e / A.pmax, k = e - pnum * A.pmax;
            const uint32_t w = A.pat[e];
            PatEntry en;
            en.pad = 0;
            if (k < (int)A.plen[pnum]) {
                en.off = 8 * (int)(short)(w & 0xffffu);
                en.val = A.dict[w >> 16];
            } else {
                en.off = zoff;
                en.val = 0.0;
            }
            ftab[e] = en;
        }
        splen[tid] = (tid < A.npat) ? ((int)A.plen[tid] | ((int)A.plen[256 + tid] << 8)) : (255 << 8);
    }
    const double d0 = A.dict[0], d1 = A.dict[A.ndict > 1 ? 1 : 0];
    const bool two = A.ndict <= 2;                       // value picked in registers instead of read from LDS
    // this wave's window descriptor of a tile (scalar loads, issued one tile ahead like the row pointers)
    struct Desc {
        mk_i4 g;
        unsigned nvw;
    };
    auto load_desc = [&](int64_t p, Desc &d) {
        d.g = mk_i4{0, 0, 0, 0};
        d.nvw = 0;
        if (p < end) {
            const int64_t t = mk_tile_at(A, p);
            d.g = mk_sload(reinterpret_cast<const mk_i4 *>(A.wg + (t * 4 + wv) * 4));
            d.nvw = mk_sload(A.wn + t * 4 + wv);
        }
    };
    MkTileMeta cur, nxt;
    Desc dcur, dnxt;
    if constexpr (!PAT) load_meta(pos, cur);
    load_desc(pos, dcur);
    for (; pos < end; pos += stride) {
        const int64_t tile = mk_tile_at(A, pos);
        const int64_t r0 = tile * MK_ROWS_PER_TILE;
        const int64_t rend = (r0 + MK_ROWS_PER_TILE < A.nrows) ? r0 + MK_ROWS_PER_TILE : A.nrows;
        const int64_t r = r0 + tid;
        constexpr bool ROWX = PAT && !PROG && MkHasRowX<Epi>::value;
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
            int lo = 0, len = 0;
            [[maybe_unused]] int kdiag = 255;
            if constexpr (PAT) {
                const unsigned id = (r < rend) ? (unsigned)A.pid[r] : 0u;   // one byte per row
                load_desc(pos + stride, dnxt);           // next tile's descriptor goes in flight
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                lo = (int)id * A.pmax;
                len = (r < rend) ? (splen[id] & 0xff) : 0;
                kdiag = splen[id] >> 8;
                const char *cell = reinterpret_cast<const char *>(xw + tid);    // this lane's own cell
                const PatEntry *pe = ftab + lo;
                PatEntry en[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) en[k] = pe[k];
                double xk[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) xk[k] = epi.xin(*reinterpret_cast<const double *>(cell + en[k].off));
#pragma unroll
                for (int k = 0; k < 8; ++k) sum += en[k].val * xk[k];
                for (int k = 8; k < len; ++k)
                    sum += pe[k].val * epi.xin(*reinterpret_cast<const double *>(cell + pe[k].off));
                if constexpr (ROWX) {                    // x[r] for the epilogue: the diagonal entry's cell
                    if (kdiag < len) xr_cur = epi.xin(*reinterpret_cast<const double *>(cell + pe[kdiag].off));
                    else if (r < rend) xr_cur = epi.xin(x[r]);
                }
                __syncthreads();                         // the next tile's copies overwrite this LDS
                if constexpr (PROG) {
                    if (r < rend) sum = mk_rowprog(A, sum, x, r, epi);
                }
                if constexpr (ROWX) {
                    if (r < rend) epi.row_x(r, sum, xr_cur, acc);
                } else {
                    if (r < rend) epi.row(r, sum, acc);
                }
                cur = nxt;
                dcur = dnxt;
                continue;
            } else {
                const int p_lo = cur.p_lo, p_hi = cur.p_hi, my_lo = cur.my_lo;
                const int base = p_lo & ~3, cnt = p_hi - base;         // 0 < cnt <= MK_SPMV_TILE + 3 (builder)
#pragma unroll
                for (int c = 0; c < 3; ++c) {                            // 256 words per wave-level copy
                    const int c0 = (wv + 4 * c) * 256;
                    if (c0 < cnt)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(A.pk + base + c0 + 4 * lane),
                                                         (__attribute__((address_space(3))) void *)(spk + c0), 16, 0, 0);
                }
                load_meta(pos + stride, nxt);            // next tile's row pointers and descriptor go in flight
                load_desc(pos + stride, dnxt);
                sptr[tid] = my_lo;
                if (tid == 0) sptr[MK_BLOCK] = p_hi;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                const int my_hi = sptr[tid + 1];
                lo = my_lo - base;
                len = my_hi - my_lo;
            }
            if constexpr (!PAT) {
            auto slot_of = [&](unsigned w, int) -> unsigned { return w & 0xffffu; };
            unsigned wk[8];
            double xk[8], vk[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) wk[k] = spk[lo + k];
            if (two) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    xk[k] = epi.xin(xw[slot_of(wk[k], k)]);
                    vk[k] = (wk[k] >> 16) ? d1 : d0;
                }
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    xk[k] = epi.xin(xw[slot_of(wk[k], k)]);
                    vk[k] = sdict[wk[k] >> 16];
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const double t = vk[k] * xk[k];
                sum += (k < len) ? t : 0.0;              // (+0.0 never changes a running sum that started at +0.0)
            }
            for (int k = 8; k < len; ++k) {
                const unsigned w = spk[lo + k];
                sum += sdict[w >> 16] * epi.xin(xw[slot_of(w, k)]);
            }
            __syncthreads();                             // the next tile's copies overwrite this LDS
            }
        } else {
            if constexpr (ROWX) {
                if (r < rend) xr_cur = epi.xin(x[r]);
            }
            if constexpr (PAT) load_meta(pos, cur);      // (tiles without windows are rare: their row pointers now)
            else load_meta(pos + stride, nxt);
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
        cur = nxt;
        dcur = dnxt;
    }
}
