// mk_spmv_fmt9l.h -- the brick march on LINEAR bricks (round 6): storage formats 9 / 10 / 11 for lines that fill 128-row bricks badly
#pragma once
// (included by mk_device.h behind mk_spmv_fmt9.h, whose helpers, ring geometry and epilogue hooks it shares)
//
// mk_spmv_fmt9.h cuts a plane into bricks of 4 LINES x 128 rows, so that a row's +-L neighbours are the same columns of the
// adjacent lines of an LDS image.  Lines of L = 200, 300, 400 rows fill such bricks to 78 %, L = 130 to 51 % (measured,
// profiles/r06_march_sizes.txt: CG on 200^3 loses 17 % against the windowed format), and a 5-point matrix has no +-L entries at
// all but pays for the two halo lines.  Here a brick is 512 CONSECUTIVE rows of the plane, whatever the line length -- every
// lane has rows except in a plane's last brick -- and the in-plane neighbours come from a FLAT LDS image of the plane over
// [b0 - Lh, b0 + 512 + Lh): entry i of the plane at F[i - b0 + 512]; row i's neighbours are F[.. - 1], F[.. + 1], F[.. - Lh],
// F[.. + Lh] (Lh = L, or 2 for a 5-point matrix: only the +-1 edges are needed).  The halo -- Lh entries below the brick, Lh above --
// is ONE 16-byte load per side and lane for Lh <= 512 (lanes past the halo's end load their own row again: an L1 hit, stored
// into the unused part of the image), so longer lines keep the line bricks, which they fill to >= 80 %.  Everything else is
// mk_spmv_fmt9.h's: the ring of R = 6 register slots for the planes z - 1, z, z + 1 of the lane's own two rows, the double-
// buffered image, one barrier per plane, the unrolled single-basic-block loop with unconditional clamped loads, the masked
// terms of absent entries, row sums left to right in column order, the fused CG hooks, the symmetric format's lower values
// from the neighbouring rows' upper ones (flat images of the +L and +1 values beside the image of x).  Rows that do not exist
// (in-plane index >= P: the plane's last brick) and planes past a chunk's end are discarded as in the general geometry
// (GEN: dump stores, +0.0 dot terms; epilogue hook row2_m), pairs start at any 8-byte boundary.  Only epilogues that may meet
// format 11 (plain products and CG) have these kernels.
//
// Fused dots: lane t of a workgroup owns the rows z P + b0 + 2 t and + 1 of its items (b0 = 512 x brick number) and adds their
// terms plane by plane, row by row (oracle/gpu_order.py `pencil_partials`, gen == 3).

constexpr int MK_PENL_F = 1536;                              // doubles per buffer of the flat image of x: [below 512 | own 512 | above 512]
constexpr int MK_PENL_LDS = 2 * MK_PENL_F;
constexpr int MK_PENL_V = 2048;                              // SYM, per buffer: +L values [below 512 | own 512], +1 values [edge 2 | own 512 | dump 510]
constexpr int MK_PENL_LDS_SYM = MK_PENL_LDS + 2 * MK_PENL_V;

template <bool PROG, bool STREAM, bool SYM, class Epi, int NACC>
__device__ __forceinline__ void mk_spmv_tiles_fmt9l(const MkCsrView &A, const double *__restrict__ x, Epi &epi,
                                                    double *smem, double (&acc)[NACC]) {
    constexpr int R = MK_PEN_R, H = MK_PEN_H, FB = MK_PENL_F, VB = MK_PENL_V;
    static_assert(!SYM || STREAM, "the symmetric march streams its values");
    static_assert(MkHasRow2M<Epi>::value || !MkHasRowXPf<Epi>::value, "linear bricks: plain products and CG only");
    constexpr bool ROWX = !PROG && MkHasRowX<Epi>::value;
    constexpr bool FUSE = MkHasFuse<Epi>::value;
    constexpr int FNT = [] { if constexpr (MkHasFuse<Epi>::value) return (int)Epi::FUSE_NT; else return 0; }();
    const int tid = threadIdx.x;
    const int64_t P = A.pen_P;
    const int Lh = A.pen_lh, Le = Lh + (Lh & 1);             // halo length (rounded up to a pair)
    const int nz = A.pen_nz, bpp = A.pen_bpp, zc = A.pen_zc;
    const int nch1 = (A.pen_zb - A.pen_za + zc - 1) / zc, nch2 = (A.pen_yb - A.pen_ya + zc - 1) / zc;
    const bool xdeal = A.pen_per > 0 && (gridDim.x & 7) == 0;
    const int64_t items = (int64_t)(xdeal ? 8 * A.pen_per : bpp) * (nch1 + nch2);
    const int64_t xtop = A.pen_xtop;
    double *gdump = A.pen_dump + (int64_t)blockIdx.x * 512 + 2 * tid;
    const int64_t off_lo = A.pen_xlo >= 0 ? A.pen_xlo : (int64_t)0;
    const int64_t off_hi = A.pen_xhi >= 0 ? A.pen_xhi : (int64_t)(nz - 1) * P;
    const uint8_t *pid = A.pid;
    [[maybe_unused]] unsigned *ptl = reinterpret_cast<unsigned *>(smem + MK_PENL_LDS);
    if constexpr (!STREAM) {
        for (int e = tid; e < 16 * A.npat; e += MK_BLOCK) ptl[e] = reinterpret_cast<const unsigned *>(A.ptab)[e];
        __syncthreads();
    }
    [[maybe_unused]] double va[7], vb[7];
    [[maybe_unused]] unsigned ma[7], mb[7], pprev = 0xffffffffu;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        va[k] = vb[k] = 0.0;
        ma[k] = mb[k] = 0u;
    }
    constexpr int VD = 2;
    [[maybe_unused]] mk_d2 vr[STREAM ? VD : 1][7];
    const bool hal = 2 * tid < Le;                           // this lane's halo pairs exist
    const int fo = 512 + 2 * tid;                            // own rows in the image
    double *cdst = smem + fo;
    double *bdst = smem + ((512 - Le + 2 * tid) & 511);      // below-halo pair (lanes past the halo: the image's unused head)
    double *adst = smem + 1024 + 2 * tid;                    // above-halo pair
    [[maybe_unused]] double *vimg = smem + MK_PENL_LDS;
    [[maybe_unused]] double *vl_own = vimg + 512 + 2 * tid, *vl_halo = vimg + ((512 - Le + 2 * tid) & 511);
    [[maybe_unused]] double *vw_own = vimg + 1024 + 2 + 2 * tid, *vw_edge = vimg + 1024 + (tid == 0 ? 1 : 516 + tid);
    [[maybe_unused]] const double *sv4 = SYM ? A.sval + A.nrows : nullptr, *sv5 = SYM ? A.sval + 2 * A.nrows : nullptr,
                                  *sv6 = SYM ? A.sval + 3 * A.nrows : nullptr;
    [[maybe_unused]] mk_d2 vlo{0.0, 0.0};

    for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
        int bi, chunk;
        if (xdeal) {
            const int per = A.pen_per;
            const int64_t q = item >> 3;
            chunk = (int)(q / per);
            bi = (int)(item & 7) * per + (int)(q % per);
            if (bi >= bpp) continue;
        } else {
            bi = (int)(item % bpp);
            chunk = (int)(item / bpp);
        }
        const int zlim = chunk < nch1 ? A.pen_zb : A.pen_yb;
        const int z0 = chunk < nch1 ? A.pen_za + chunk * zc : A.pen_ya + (chunk - nch1) * zc, z1 = (z0 + zc < zlim) ? z0 + zc : zlim;
        const int64_t b0 = (int64_t)bi * 512;
        const int64_t c = b0 + 2 * tid;                       // this lane's rows c, c + 1 (in-plane index)
        const bool oka = c < P, okb = c + 1 < P;
        // halo pairs of this lane: Le entries below the brick, Le above (lanes past the halo's end: their own rows again)
        const int64_t hb = hal ? b0 - Le + 2 * tid : c, ha = hal ? b0 + 512 + 2 * tid : c;
        auto at = [&](const double *v, int64_t i) -> mk_d2 { return mk_pen_ld2<false>(v, i < 0 ? (int64_t)0 : i, xtop); };
        auto plane = [&](int p) -> mk_d2 {
            const int64_t o = p < 0 ? off_lo : (p > nz - 1 ? off_hi : (int64_t)p * P);
            return mk_pen_ld2<false>(x, o + c, xtop);
        };
        [[maybe_unused]] auto plane_of = [&](const double *v, int p) -> mk_d2 {
            const int64_t o = p < 0 ? off_lo : (p > nz - 1 ? off_hi : (int64_t)p * P);
            return mk_pen_ld2<false>(v, o + c, xtop);
        };
        [[maybe_unused]] auto plane_of_x = [&](const double *v, int p) -> mk_d2 {
            const int64_t o = p < 0 ? off_lo : (p > nz - 1 ? off_hi : (int64_t)p * P);
            return mk_pen_ld2<(FNT & 1) != 0>(v, o + c, xtop);
        };
        mk_d2 hbr[H], har[H];                                 // x at the halo pairs
        mk_d2 ring[R];
        unsigned pidr[H];
        [[maybe_unused]] mk_d2 hvr[SYM ? H : 1];              // SYM: +L values of the rows below the brick
        [[maybe_unused]] double evr[SYM ? H : 1];             // SYM: +1 value of the row before the brick
        [[maybe_unused]] mk_d2 hbq[FUSE ? H : 1], haq[FUSE ? H : 1];   // fuse: r at the halo pairs
        [[maybe_unused]] mk_d2 rr[FUSE ? H : 1], xx[FUSE ? H : 1];
        auto halo = [&](int p, int d) {
            p = p > nz - 1 ? nz - 1 : p;
            const int64_t o = (int64_t)p * P;
            hbr[d] = at(x, o + hb);
            har[d] = at(x, o + ha);
            if constexpr (FUSE) {
                hbq[d] = at(epi.fuse_r, o + hb);
                haq[d] = at(epi.fuse_r, o + ha);
            }
            pidr[d] = *reinterpret_cast<const mk_u16u *>(pid + o + c);
            if constexpr (SYM) {
                const int64_t j = o + hb;                     // (the value arrays have slack behind them; not in front)
                hvr[d] = *reinterpret_cast<const mk_d2u *>(sv5 + (j < 0 ? (int64_t)0 : j));
                const int64_t e = o + b0 - 1;
                evr[d] = sv4[e < 0 ? (int64_t)0 : e];
            }
        };
        [[maybe_unused]] auto transform = [&](mk_d2 &pv, const mk_d2 rv, const mk_d2 xv, int pl, bool lv) {
            if constexpr (FUSE) {
                const mk_d2 po = pv;
                pv.x = epi.fuse_pnew(po.x, rv.x);
                pv.y = epi.fuse_pnew(po.y, rv.y);
                const bool own = pl >= z0 && pl < z1;
                double *dump = epi.fuse_dump + (int64_t)blockIdx.x * 1024 + 2 * tid;
                mk_d2 xn;
                xn.x = epi.fuse_xnew(xv.x, po.x);
                xn.y = epi.fuse_xnew(xv.y, po.y);
                double *pt = own ? epi.fuse_p + (int64_t)pl * P + c : nullptr;
                pt = (pl == -1 && A.pen_xlo >= 0) ? epi.fuse_p + off_lo + c : pt;
                pt = (pl == nz && A.pen_xhi >= 0) ? epi.fuse_p + off_hi + c : pt;
                pt = lv ? pt : nullptr;
                double *xt = (own && lv) ? epi.fuse_x + (int64_t)pl * P + c : nullptr;
                mk_d2u *pd = reinterpret_cast<mk_d2u *>((pt && okb) ? pt : dump);
                mk_d2u *xd = reinterpret_cast<mk_d2u *>((xt && okb) ? xt : dump + 512);
                if constexpr (FNT & 4) __builtin_nontemporal_store(pv, pd);
                else *pd = pv;
                if constexpr (FNT & 2) __builtin_nontemporal_store(xn, xd);
                else *xd = xn;
                if (oka && !okb) {
                    if (pt) *pt = pv.x;
                    if (xt) *xt = xn.x;
                }
            }
        };
        [[maybe_unused]] auto halo_val = [&](mk_d2 pv, mk_d2 rv) -> mk_d2 {
            if constexpr (FUSE) {
                mk_d2 t;
                t.x = epi.fuse_pnew(pv.x, rv.x);
                t.y = epi.fuse_pnew(pv.y, rv.y);
                return t;
            } else {
                return pv;
            }
        };
        [[maybe_unused]] auto vals = [&](int p, int sl) {
            p = p > nz - 1 ? nz - 1 : p;
            if constexpr (SYM) {
#pragma unroll
                for (int k = 3; k < 7; ++k)
                    vr[sl][k] = __builtin_nontemporal_load(reinterpret_cast<const mk_d2u *>(A.sval + (int64_t)(k - 3) * A.nrows + (int64_t)p * P + c));
            } else if constexpr (STREAM) {
#pragma unroll
                for (int k = 0; k < 7; ++k)
                    vr[sl][k] = __builtin_nontemporal_load(reinterpret_cast<const mk_d2u *>(A.sval + (int64_t)k * A.nrows + (int64_t)p * P + c));
            }
        };
        auto step = [&](int zz, int b, const mk_d2 xm_, const mk_d2 xc_, const mk_d2 xp_, const mk_d2 hbv, const mk_d2 hav, unsigned pp,
                        const mk_d2 (&vv)[7], [[maybe_unused]] const mk_d2 hvv, [[maybe_unused]] double evv, auto &&reload, auto &&after,
                        bool live) {
            const int bo = b * FB;
            mk_d2 xm, xc, xp;
            xm.x = epi.xin(xm_.x); xm.y = epi.xin(xm_.y);
            xc.x = epi.xin(xc_.x); xc.y = epi.xin(xc_.y);
            xp.x = epi.xin(xp_.x); xp.y = epi.xin(xp_.y);
            [[maybe_unused]] const int vbo = b * VB;
            if constexpr (SYM) {
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    ma[k] = (unsigned)__builtin_amdgcn_sbfe((int)pp, k, 1);
                    mb[k] = (unsigned)__builtin_amdgcn_sbfe((int)pp, 8 + k, 1);
                }
#pragma unroll
                for (int k = 3; k < 7; ++k) {
                    va[k] = vv[k].x;
                    vb[k] = vv[k].y;
                }
                va[0] = mk_pen_sel(vlo.x, ma[0]);
                vb[0] = mk_pen_sel(vlo.y, mb[0]);
                vb[2] = mk_pen_sel(vv[4].x, mb[2]);
                vlo = vv[6];
                *reinterpret_cast<mk_d2 *>(vl_own + vbo) = vv[5];
                *reinterpret_cast<mk_d2 *>(vw_own + vbo) = vv[4];
                *reinterpret_cast<mk_d2 *>(vl_halo + vbo) = hvv;
                vw_edge[vbo] = evv;
            } else if constexpr (STREAM) {
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    va[k] = vv[k].x;
                    vb[k] = vv[k].y;
                    ma[k] = (unsigned)__builtin_amdgcn_sbfe((int)pp, k, 1);
                    mb[k] = (unsigned)__builtin_amdgcn_sbfe((int)pp, 8 + k, 1);
                }
            } else if (__builtin_amdgcn_ballot_w64(pp != pprev) != 0) {
                const mk_u4 *ta = reinterpret_cast<const mk_u4 *>(ptl + 16 * (pp & 0xffu));
                const mk_u4 *tb = reinterpret_cast<const mk_u4 *>(ptl + 16 * (pp >> 8));
                const mk_u4 a0 = ta[0], a1 = ta[1], a2 = ta[2], a3 = ta[3], b0_ = tb[0], b1 = tb[1], b2 = tb[2], b3 = tb[3];
                const unsigned wa[16] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, a3.x, a3.y, a3.z, a3.w};
                const unsigned wb[16] = {b0_.x, b0_.y, b0_.z, b0_.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    va[k] = __hiloint2double((int)wa[2 * k + 1], (int)wa[2 * k]);
                    vb[k] = __hiloint2double((int)wb[2 * k + 1], (int)wb[2 * k]);
                    ma[k] = ((wa[14] >> k) & 1u) ? 0xffffffffu : 0u;
                    mb[k] = ((wb[14] >> k) & 1u) ? 0xffffffffu : 0u;
                }
                pprev = pp;
            }
            *reinterpret_cast<mk_d2 *>(cdst + bo) = xc;
            {
                mk_d2 t;
                t.x = epi.xin(hbv.x); t.y = epi.xin(hbv.y);
                *reinterpret_cast<mk_d2 *>(bdst + bo) = t;
                t.x = epi.xin(hav.x); t.y = epi.xin(hav.y);
                *reinterpret_cast<mk_d2 *>(adst + bo) = t;
            }
            reload();
            __syncthreads();
            const double *row = cdst + bo;
            const double lox = row[-Lh], loy = row[1 - Lh], upx = row[Lh], upy = row[1 + Lh];
            const double we = row[-1], ea = row[2];
            if constexpr (SYM) {
                const double *vl = vl_own + vbo - Lh;          // the +L values of the rows r - L, r + 1 - L
                va[1] = mk_pen_sel(vl[0], ma[1]);
                vb[1] = mk_pen_sel(vl[1], mb[1]);
                va[2] = mk_pen_sel(vw_own[vbo - 1], ma[2]);    // a(c, c - 1) = row c - 1's +1 value
            }
            const int64_t r = (int64_t)zz * P + c;
            const double na[7] = {xm.x, lox, we, xc.x, xc.y, upx, xp.x}, nb[7] = {xm.y, loy, xc.x, xc.y, ea, upy, xp.y};
            double sa = 0.0, sb = 0.0;
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                sa = mk_pen_term(sa, va[k], ma[k], na[k]);
                sb = mk_pen_term(sb, vb[k], mb[k], nb[k]);
            }
            const bool la = oka && live, lb = okb && live;
            if constexpr (PROG) {
                if (la) sa = mk_rowprog(A, sa, x, r, epi);
                if (lb) sb = mk_rowprog(A, sb, x, r + 1, epi);
            }
            if constexpr (MkHasRow2M<Epi>::value) {
                mk_d2 s2;
                s2.x = sa;
                s2.y = sb;
                epi.row2_m(r, s2, xc, la, lb, gdump, acc);
            } else if constexpr (ROWX) {
                if (la) epi.row_x(r, sa, xc.x, acc);
                if (lb) epi.row_x(r + 1, sb, xc.y, acc);
            } else {
                if (la) {
                    if constexpr (MkHasPre<Epi>::value) epi.pre(r);
                    epi.row(r, sa, acc);
                }
                if (lb) {
                    if constexpr (MkHasPre<Epi>::value) epi.pre(r + 1);
                    epi.row(r + 1, sb, acc);
                }
            }
            after();
        };
        if constexpr (SYM) {
            const double *src = z0 > 0 ? sv6 + (int64_t)(z0 - 1) * P + c : (A.pen_xlo >= 0 ? A.sval + 4 * A.nrows + c : sv6 + c);
            vlo = *reinterpret_cast<const mk_d2u *>(src);
        }
        // whole rounds, the last one masked past z1; chunks of one or two planes (a slab's boundary launch): one plane after the other
        const int zfull = z1 - z0 > 2 ? z0 + ((z1 - z0 + R - 1) / R) * R : z0;
        if (zfull > z0) {
            [[maybe_unused]] mk_d2 rm1{0.0, 0.0}, r00{0.0, 0.0}, x00{0.0, 0.0};
            if constexpr (FUSE) {
                rm1 = plane_of(epi.fuse_r, z0 - 1);
                r00 = plane_of(epi.fuse_r, z0);
                x00 = plane_of_x(epi.fuse_x, z0);
            }
#pragma unroll
            for (int d = 0; d < R - 1; ++d) {
                ring[d] = plane(z0 - 1 + d);
                if (d < H) halo(z0 + d, d);
                if (d < VD) vals(z0 + d, d);
                if constexpr (FUSE) {
                    if (d < H) {
                        rr[(1 + d) % H] = plane_of(epi.fuse_r, z0 + 1 + d);
                        xx[(1 + d) % H] = plane_of_x(epi.fuse_x, z0 + 1 + d);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0)
            if constexpr (FUSE) {
                transform(ring[0], rm1, rm1, z0 - 1, true);
                transform(ring[1], r00, x00, z0, true);
            }
            int z = z0;
            do {
#pragma unroll
                for (int d = 0; d < R; ++d) {
                    const int zz = z + d;
                    const bool live = zz < z1;
                    if constexpr (FUSE) transform(ring[(d + 2) % R], rr[(d + 1) % H], xx[(d + 1) % H], zz + 1, live);
                    step(zz, d & 1, ring[d], ring[(d + 1) % R], ring[(d + 2) % R], halo_val(hbr[d % H], hbq[FUSE ? d % H : 0]),
                         halo_val(har[d % H], haq[FUSE ? d % H : 0]), pidr[d % H], vr[STREAM ? d % VD : 0], hvr[SYM ? d % H : 0],
                         evr[SYM ? d % H : 0], [&]() {
                             __builtin_amdgcn_sched_barrier(0);
                             ring[(d + R - 1) % R] = plane(zz + R - 2);
                             halo(zz + H, d % H);
                             if constexpr (FUSE) {
                                 rr[(d + 1) % H] = plane_of(epi.fuse_r, zz + 1 + H);
                                 xx[(d + 1) % H] = plane_of_x(epi.fuse_x, zz + 1 + H);
                             }
                         }, [&]() { vals(zz + VD, d % VD); }, live);
                }
                z += R;
            } while (z < zfull);
        }
        for (int zz = zfull; zz < z1; ++zz) {                 // (chunks of one or two planes)
            mk_d2 xm = plane(zz - 1), xc = plane(zz), xp = plane(zz + 1);
            halo(zz, 0);
            if constexpr (FUSE) {
                const mk_d2 ra = plane_of(epi.fuse_r, zz - 1), rb = plane_of(epi.fuse_r, zz), rc = plane_of(epi.fuse_r, zz + 1);
                const mk_d2 xb = plane_of_x(epi.fuse_x, zz), xcn = plane_of_x(epi.fuse_x, zz + 1);
                if (zz == z0) {
                    transform(xm, ra, ra, zz - 1, true);
                    transform(xc, rb, xb, zz, true);
                } else {                                      // (formed again, written before: x is not touched twice)
                    xm.x = epi.fuse_pnew(xm.x, ra.x); xm.y = epi.fuse_pnew(xm.y, ra.y);
                    xc.x = epi.fuse_pnew(xc.x, rb.x); xc.y = epi.fuse_pnew(xc.y, rb.y);
                }
                transform(xp, rc, xcn, zz + 1, true);
            }
            vals(zz, 0);
            step(zz, (zz - zfull) & 1, xm, xc, xp, halo_val(hbr[0], hbq[0]), halo_val(har[0], haq[0]), pidr[0], vr[0], hvr[0], evr[0],
                 [] {}, [] {}, true);
        }
        __syncthreads();                                     // the next item's first plane image overwrites this LDS
    }
}
