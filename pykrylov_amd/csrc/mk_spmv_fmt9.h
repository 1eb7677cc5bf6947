// mk_spmv_fmt9.h -- z-marching bricks for 7-point-class matrices (one pattern byte per row + a value dictionary)
#pragma once
// (included by mk_device.h: one SpMV loop per storage format behind the same Epi / Gate / row_x interface)
//
// fmt 9 (round 5; any grid side since round 6, see GENERAL GEOMETRY below -- this paragraph describes whole aligned bricks: L a
// multiple of 128, P a multiple of 4 L).  A matrix whose column offsets all lie in {0, +-1, +-L, +-P} with nrows a multiple of
// P -- the 7-point stencil of an nx x ny x nz grid (L = nx, P = nx ny), any boundary
// treatment, any band matrix of that shape -- and whose values come from <= 256 distinct bit patterns is stored as ONE
// BYTE PER ROW (the number of the row's pattern) plus a table of 64 bytes per pattern: {7 values in column order, +0.0
// where the row has no entry, a 7-bit presence mask}.  Same data volume as fmt 4; what changes is how x is read.
//
// fmt 4 ingests, per 256-row tile, every x window the tile touches (own, +-line, +-plane): each x entry is requested five
// times, crosses the fabric about twice (counter traffic 2.03 x the format's bytes at 512^3), and a tile's chain
// copies -> barrier -> row walk exposes the copy latency (0.75 ms for 2.3 GB).  Here a workgroup owns a BRICK -- 4 lines
// x 128 rows, two adjacent rows per lane -- and marches through `zc` planes:
//   * the x entries at the lane's own two rows of the planes z-1, z, z+1 ride in a ring of R = 6 register slots: every
//     entry is loaded ONCE per product, 16 bytes per lane, R - 3 planes ahead of its first use, always into the slot
//     whose plane went out of use one step EARLIER (a slot reloaded while its old value is still needed makes the
//     compiler rotate the ring through register copies at the loop's back edge, and a copy of a register with a load in
//     flight waits for that load: the pipeline drained once per round);
//   * the in-plane neighbours (+-1, +-L) come from a double-buffered LDS image of the brick's current plane with a one-row
//     halo (2 x 128 halo rows + 8 edge rows per plane: one 8-byte load per lane, R planes ahead; they are L2 hits --
//     the neighbouring bricks' own rows);
//   * one barrier per plane, 16-byte stores of the product.
// The loop is unrolled R times so that every ring index is static, every load is unconditional (addresses clamped into
// the vector) and no branch in the loop diverges: only then does the compiler count the loads in flight instead of
// draining them (vmcnt(0)) once per round -- tools/ubench/pencil2.hip, where this structure moves the compulsory 16 N
// bytes of the 512^3 product at 5.0 TB/s (0.43 ms) against 3.6 TB/s for round 4's unpipelined walker.
//
// Row sums: LEFT TO RIGHT in column order (-P, -L, -1, 0, +1, +L, +P), every product and every add rounds separately: the
// bits of the scalar CSR loop.  The loop body is ONE basic block -- a waterfall over the wave's distinct patterns (fmt 8's
// way) puts inner loops into the body, the register allocator then splits the ring slots' live ranges around them and
// rotates the ring through copies at the back edge, each of which waits for a load issued a step earlier (measured in the
// ISA: vmcnt(4) once per round).  So a lane keeps its two rows' patterns in registers -- seven values and seven 32-bit
// AND masks each, reloaded from an LDS copy of the table only when a pattern byte of the wave changes (a wave-uniform
// `if` that a stencil takes at the first and the last plane) -- and an entry the row does not have costs one v_and: its
// value is +0.0 and the HIGH word of the x candidate is masked to zero, which leaves a tiny non-negative finite number, so
// the product is exactly +0.0 whatever the candidate held (Inf, NaN included) and adding it leaves the running sum --
// which started at +0.0 and therefore is never -0.0 -- unchanged bit for bit.  Everything is index based (row r's
// neighbours are the vector entries r +- 1, r +- L, r +- P): no grid geometry is assumed beyond the three strides.
//
// Fused dots: lane (w, l) of a workgroup owns the rows z P + c, z P + c + 1 (c = brick base + w L + 2 l) of its items
// (items blockIdx, blockIdx + grid, ...; item i = brick (i % 8) bpp / 8 + (i / 8) % (bpp / 8) of chunk (i / 8) / (bpp / 8)
// when the bricks per plane bpp and the grid are multiples of 8 -- an XCD-contiguous deal --, else brick i % bpp of chunk
// i / bpp), and adds their terms plane by plane,
// row c before row c + 1 (the tests restate this order on the host to demand bit equality).

// Optional epilogue hook set "fuse" (CG, mk_cg.hip): the product's input vector is FORMED on the fly from the previous
// pass -- p = beta p_old - r (cg.py:150-151) -- and the deferred x += alpha p_old (cg.py:130) rides along: the own rows
// of every plane are loaded from p_old, r and x, transformed once when the plane enters the ring's front, and written
// out (p to a second buffer: the neighbouring bricks still read p_old for their halo rows and form the same p
// redundantly); the halo rows are formed from p_old and r on the fly.  One pass over x, p, r less per CG iteration.
template <class Epi, class = void>
struct MkHasFuse : std::false_type {};
template <class Epi>
struct MkHasFuse<Epi, std::void_t<decltype(std::declval<Epi &>().fuse_r)>> : std::true_type {};

// GENERAL GEOMETRY (round 6; template flag GEN).  The march above needs L % 128 == 0, P % 4L == 0: bricks tile the plane and every
// pair of rows starts at a 16-byte boundary.  GEN lifts both: a plane of P rows is cut into lines of L rows (the last one may be
// short), a line into ceil(L / 128) bricks, lines are grouped in fours -- the last brick of a line and the last group of a
// plane are partly EMPTY.  A lane whose row does not exist (in-line position >= L, or in-plane index >= P) still loads at its
// natural index -- which is what makes the +-1 neighbours across line ends come out right, index based as everything here --
// clamped into the vector, computes a row sum from whatever it found, and DISCARDS it: its stores go to a dump row, its term of
// a fused dot is replaced by +0.0 (selects, no branch: the loop body stays one basic block).  Pairs start at any 8-byte
// boundary (odd L or P): 16-byte accesses through an 8-byte-aligned type; a pair that starts at the vector's LAST entry reads
// one entry of the slack every device buffer has behind it.  The same masking runs a chunk's LEFTOVER planes (planes % R != 0) as one more
// pipelined round whose planes past the end are discarded, instead of one unpipelined plane after the other -- which is why
// a slab's boundary launch takes this kernel on any geometry (mk_device.h).  Only epilogues that may meet format 11
// (SYM_MARCH: plain products and CG) have GEN instantiations; they provide the masked pair hook
//     void row2_m(int64_t r, mk_d2 s, mk_d2 xr, bool oka, bool okb, double *dump, double *acc)
// = row_x(r, s.x, xr.x, acc) if oka; row_x(r + 1, s.y, xr.y, acc) if okb (okb implies oka).
typedef mk_d2 mk_d2u __attribute__((aligned(8)));
typedef uint16_t mk_u16u __attribute__((aligned(1)));
template <class Epi, class = void>
struct MkHasRow2M : std::false_type {};
template <class Epi>
struct MkHasRow2M<Epi, std::void_t<decltype(std::declval<Epi &>().row2_m((int64_t)0, mk_d2{}, mk_d2{}, false, false, (double *)nullptr,
                                                                         (double *)nullptr))>> : std::true_type {};

// the pair v[idx], v[idx + 1] of an input vector, idx clamped to `top` = the vector's last entry: a pair that STARTS there
// reads one entry past the end -- every device buffer of this library has 16 bytes of slack behind it (mk_malloc, alloc_vec;
// a slice of a longer vector has its neighbour there) and the second half is a row that does not exist, so it is discarded.
// (No select on the loaded value: a fix-up after the load would pin a wait for it right behind the load.)
template <bool NT>
__device__ __forceinline__ mk_d2 mk_pen_ld2(const double *v, int64_t idx, int64_t top) {
    const mk_d2u *p = reinterpret_cast<const mk_d2u *>(v + (idx > top ? top : idx));
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

// one term of a row sum: s + v * x, x with its high word ANDed by m (0xffffffff: the row has the entry; 0: it has not, v = +0.0)
__device__ __forceinline__ double mk_pen_term(double s, double v, unsigned m, double xk) {
    return s + v * __hiloint2double((int)((unsigned)__double2hiint(xk) & m), __double2loint(xk));
}

#ifndef MK_PEN_R_DEF
#define MK_PEN_R_DEF 6
#endif
#ifndef MK_PEN_H_DEF
#define MK_PEN_H_DEF 3
#endif
#ifndef MK_PEN_OCC
#define MK_PEN_OCC 2                                         // workgroups per CU the register budget is cut for (tools/variants)
#endif
constexpr int MK_PEN_R = MK_PEN_R_DEF;                                  // ring slots = unroll factor (own rows: prefetch depth R - 3 planes)
constexpr int MK_PEN_H = MK_PEN_H_DEF;                                  // slots of the halo / pattern-byte rings (depth H planes; R % H == 0)
constexpr int MK_PEN_RS = 132;                               // LDS row: [0] pad, [1] west edge, [2..129] rows, [130] east edge, [131] pad
constexpr int MK_PEN_LDS = 3 * 6 * MK_PEN_RS + MK_BLOCK;     // doubles: two plane images of 6 rows + the dump rows (lanes without an
                                                             // edge row store there, at the same buffer offset as the others)

// SYM (storage format 11): format 10 for matrices that are SYMMETRIC bit for bit -- what CG runs on.  Only the diagonal and the
// upper offsets (+1, +L, +P) are stored, 32 B per row instead of 56: a(r, r-1) is row r-1's +1 value -- the left lane's, or
// for a wave's first lane the west edge row's; a(r, r-L) is row r-L's +L value -- the lane one brick line below, or for the
// first line the halo line's; a(r, r-P) is the +P value the lane itself loaded one plane earlier.  The first two travel
// through an LDS image of the plane's values beside the image of x (written before the step's barrier, read after it), the
// third stays in two registers.  A lower value is ANDed with the row's presence mask: the product's mask trick needs the VALUE
// of an absent entry to be +0.0, and a clamped edge load may have fetched a stranger's.  The first plane of a rank's slab
// takes its -P values (the neighbour's +P values) from an array of one plane behind the four value arrays.
constexpr int MK_PEN_VB = 6 * 128 + 4 * MK_PEN_RS;           // doubles per buffer of the value image: 5 lines of +L values (halo
                                                             // line, four brick lines) + a dump line, 4 lines of +1 values
constexpr int MK_PEN_LDS_SYM = MK_PEN_LDS + 2 * MK_PEN_VB;

// v where the mask is all ones, +0.0 where it is zero
__device__ __forceinline__ double mk_pen_sel(double v, unsigned m) {
    return __hiloint2double((int)((unsigned)__double2hiint(v) & m), (int)((unsigned)__double2loint(v) & m));
}

// STREAM (storage format 10): the same march for matrices of the class WITHOUT a value dictionary (variable coefficients):
// the byte per row is the row's 7-bit presence mask itself and the values are streamed from seven arrays in column-position-
// major order, sval[k * nrows + r] = the value at offset k of row r or +0.0 (one 16-byte non-temporal load per position and
// lane and plane, two planes ahead): 56 B per row of values -- what fmt 5 streams -- with every x entry loaded once.
template <bool PROG, bool STREAM, bool SYM, bool GEN, class Epi, int NACC>
__device__ __forceinline__ void mk_spmv_tiles_fmt9(const MkCsrView &A, const double *__restrict__ x, Epi &epi,
                                                   double *smem, double (&acc)[NACC]) {
    constexpr int R = MK_PEN_R, RS = MK_PEN_RS, BUF = 6 * MK_PEN_RS;
    static_assert(!SYM || STREAM, "the symmetric march streams its values");
    static_assert(MK_PEN_R % 2 == 0 && MK_PEN_R % MK_PEN_H == 0, "ring geometry");
    static_assert(MK_PEN_R == 6, "mk_pen_split (mk_device.h) cuts a slab's boundary planes in rounds of 6");
    constexpr bool ROWX = !PROG && MkHasRowX<Epi>::value;
    constexpr bool FUSE = MkHasFuse<Epi>::value;
    constexpr int FNT = [] { if constexpr (MkHasFuse<Epi>::value) return (int)Epi::FUSE_NT; else return 0; }();
    // epilogue operands loaded at the top of a step (mk_device.h: row_pf / row_x_pf)
    constexpr bool XPF = !PROG && MkHasRowXPf<Epi>::value;
    constexpr bool RPF = !XPF && MkHasRowPf<Epi>::value;
    static_assert(!GEN || !(XPF || RPF), "general geometry: plain products and CG only (no prefetched epilogue operands)");
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int64_t L = A.pen_L, P = A.pen_P;
    const int nz = A.pen_nz, bx = A.pen_bx, bpp = A.pen_bpp, zc = A.pen_zc;
    // this launch's planes: [pen_za, pen_zb) and [pen_ya, pen_yb) in chunks of zc (a whole product: [0, nz) and nothing)
    const int nch1 = (A.pen_zb - A.pen_za + zc - 1) / zc, nch2 = (A.pen_yb - A.pen_ya + zc - 1) / zc;
    // GEN: the XCD-contiguous deal for ANY number of bricks per plane: XCD k takes the bricks [k per, (k + 1) per), per =
    // ceil(bpp / 8); a chunk has 8 per item slots, the ones past the plane's last brick are empty (pen_per == 0: planes of
    // fewer than 64 bricks, dealt round robin)
    const bool xdeal = GEN && A.pen_per > 0 && (gridDim.x & 7) == 0;
    const int64_t items = (int64_t)(xdeal ? 8 * A.pen_per : bpp) * (nch1 + nch2);
    const int64_t last = A.nrows - 1;
    [[maybe_unused]] const int64_t xtop = A.pen_xtop;         // GEN: the input vector's last entry (loads are clamped to it)
    [[maybe_unused]] double *gdump = GEN ? A.pen_dump + (int64_t)blockIdx.x * 512 + 2 * tid : nullptr;
    // a rank's slab (mk_csr_localize mode 0): the planes below plane 0 / above plane nz - 1 are the neighbours', received
    // behind the own rows of the input vector; on one device the march clamps into the grid (their entries are masked).
    // Localising renumbers the columns in place and keeps every row's STORAGE order -- the order of the global columns --
    // so the scalar loop adds the entry in the plane below first as everywhere: slot 0, the same bits as on one device.
    const int64_t off_lo = A.pen_xlo >= 0 ? A.pen_xlo : (int64_t)0;
    const int64_t off_hi = A.pen_xhi >= 0 ? A.pen_xhi : (int64_t)(nz - 1) * P;
    const uint8_t *pid = A.pid;
    // the pattern table, 16 words per pattern, behind the plane images (read-only after this copy)
    [[maybe_unused]] unsigned *ptl = reinterpret_cast<unsigned *>(smem + MK_PEN_LDS);
    if constexpr (!STREAM) {
        for (int e = tid; e < 16 * A.npat; e += MK_BLOCK) ptl[e] = reinterpret_cast<const unsigned *>(A.ptab)[e];
        __syncthreads();
    }
    [[maybe_unused]] double va[7], vb[7];                    // this lane's two rows: values and AND masks of the 7 candidates
    [[maybe_unused]] unsigned ma[7], mb[7], pprev = 0xffffffffu;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        va[k] = vb[k] = 0.0;
        ma[k] = mb[k] = 0u;
    }
    constexpr int VD = 2;                                    // STREAM: planes of values in flight (slot = plane parity)
    [[maybe_unused]] mk_d2 vr[STREAM ? VD : 1][7];
    double *cdst = smem + (1 + w) * RS + 2 + 2 * l;
    double *hdst = smem + (tid < 128 ? 0 : 5) * RS + 2 + (tid & 127);
    double *edst = tid < 8 ? smem + (1 + (tid & 3)) * RS + ((tid & 4) ? 130 : 1) : smem + 2 * BUF + tid;
    // SYM: the value image (per buffer: +L values [6][128] -- line 0 the halo line below the brick, 1..4 the brick's lines, 5 a
    // dump line -- then +1 values [4][RS] with the west edge row's at [1])
    [[maybe_unused]] double *vimg = smem + MK_PEN_LDS;
    [[maybe_unused]] double *vl_own = vimg + (1 + w) * 128 + 2 * l, *vl_halo = vimg + (tid < 128 ? 0 : 5) * 128 + (tid & 127);
    [[maybe_unused]] double *vw_own = vimg + 6 * 128 + w * RS + 2 + 2 * l;
    [[maybe_unused]] double *vw_edge = tid < 4 ? vimg + 6 * 128 + tid * RS + 1 : vimg + 5 * 128 + (tid & 127);
    [[maybe_unused]] const double *sv4 = SYM ? A.sval + A.nrows : nullptr, *sv5 = SYM ? A.sval + 2 * A.nrows : nullptr,
                                  *sv6 = SYM ? A.sval + 3 * A.nrows : nullptr;     // (+1, +L, +P values; the diagonal's lead the block)
    [[maybe_unused]] mk_d2 vlo{0.0, 0.0};                    // SYM: the +P values of the plane before = this plane's -P values

    for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
        // item -> (brick, chunk).  Workgroup b runs on XCD b % 8 and every XCD has its own L2: with bricks dealt round robin the
        // halo rows of a brick -- the own rows of its neighbours in y -- always belong to another XCD and cross the fabric a
        // second time (fused CG kernel at 512^3: 4.98 GB read for 3.35 GB of operands, profiles/r05_*).  So XCD k takes the
        // k-th contiguous eighth of a plane's bricks (whole brick rows when the plane has a multiple of 8 of them).
        int bi, chunk;
        if constexpr (GEN) {
            if (xdeal) {
                const int per = A.pen_per;
                const int64_t q = item >> 3;
                chunk = (int)(q / per);
                bi = (int)(item & 7) * per + (int)(q % per);
                if (bi >= bpp) continue;                      // (an empty slot: workgroup uniform)
            } else {
                bi = (int)(item % bpp);
                chunk = (int)(item / bpp);
            }
        } else if ((bpp & 7) == 0 && (gridDim.x & 7) == 0) {
            const int per = bpp >> 3;
            const int64_t q = item >> 3;
            chunk = (int)(q / per);
            bi = (int)(item & 7) * per + (int)(q % per);
        } else {
            bi = (int)(item % bpp);
            chunk = (int)(item / bpp);
        }
        const int zlim = chunk < nch1 ? A.pen_zb : A.pen_yb;
        const int z0 = chunk < nch1 ? A.pen_za + chunk * zc : A.pen_ya + (chunk - nch1) * zc, z1 = (z0 + zc < zlim) ? z0 + zc : zlim;
        const int64_t b0 = (int64_t)(bi / bx) * 4 * L + (int64_t)(bi % bx) * 128;    // the brick's first row in plane 0
        const int64_t c = b0 + (int64_t)w * L + 2 * l;                                // this lane's rows c, c + 1 (in-plane index)
        // GEN: do this lane's rows exist?  (in-line position below L, in-plane index below P; okb implies oka)
        [[maybe_unused]] const int cx = (bi % bx) * 128 + 2 * l;
        [[maybe_unused]] const bool oka = !GEN || (cx < L && c < P), okb = !GEN || (cx + 1 < L && c + 1 < P);
        // halo row of this lane: lanes 0..127 the line below the brick, 128..255 the line above; lanes 0..7 also the row
        // west / east of brick line tid & 3 (lanes >= 8 load lane (tid & 7)'s edge row again and drop it into the dump row)
        // (a 5-point matrix marched line by line has no +-L entries: its halo-line loads go to the lane's own row -- an L1 hit)
        const int64_t hc = (GEN && A.pen_nol) ? c : (tid < 128 ? b0 - L + tid : b0 + 4 * L + (tid - 128));
        const int64_t ec = b0 + (int64_t)(tid & 3) * L + ((tid & 4) ? 128 : -1);
        auto plane = [&](int p) -> mk_d2 {                    // the own rows of plane p (clamped into the grid: values of a
            // plane that does not exist are never multiplied; a slab's neighbour planes come from the received entries)
            const int64_t o = p < 0 ? off_lo : (p > nz - 1 ? off_hi : (int64_t)p * P);
            if constexpr (GEN) return mk_pen_ld2<false>(x, o + c, xtop);
            else return *reinterpret_cast<const mk_d2 *>(x + o + c);
        };
        auto clampr = [&](int64_t r) { return r < 0 ? (int64_t)0 : (r > last ? last : r); };
        constexpr int H = MK_PEN_H;
        double hreg[H], ereg[H];
        mk_d2 ring[R];
        unsigned pidr[H];
        [[maybe_unused]] double hvr[SYM ? H : 1], evr[SYM ? H : 1];     // SYM: +L values of the halo line below, +1 values of the west edge rows
        [[maybe_unused]] double hr[FUSE ? H : 1], er[FUSE ? H : 1];      // fuse: r at the halo rows
        [[maybe_unused]] mk_d2 rr[FUSE ? H : 1], xx[FUSE ? H : 1];      // fuse: r and x at the own rows of the plane to transform next
        auto halo = [&](int p, int d) {                       // plane p at the halo rows + the pattern bytes of the own rows
            p = p > nz - 1 ? nz - 1 : p;
            hreg[d] = x[clampr((int64_t)p * P + hc)];
            ereg[d] = x[clampr((int64_t)p * P + ec)];
            if constexpr (FUSE) {
                hr[d] = epi.fuse_r[clampr((int64_t)p * P + hc)];
                er[d] = epi.fuse_r[clampr((int64_t)p * P + ec)];
            }
            if constexpr (GEN) pidr[d] = *reinterpret_cast<const mk_u16u *>(pid + (int64_t)p * P + c);   // (the array has slack behind it)
            else pidr[d] = *reinterpret_cast<const uint16_t *>(pid + (int64_t)p * P + c);
            if constexpr (SYM) {
                hvr[d] = sv5[clampr((int64_t)p * P + hc)];
                evr[d] = sv4[clampr((int64_t)p * P + ec)];
            }
        };
        [[maybe_unused]] auto plane_of = [&](const double *v, int p) -> mk_d2 {     // (r of a slab's neighbour planes: received)
            const int64_t o = p < 0 ? off_lo : (p > nz - 1 ? off_hi : (int64_t)p * P);
            if constexpr (GEN) return mk_pen_ld2<false>(v, o + c, xtop);
            else return *reinterpret_cast<const mk_d2 *>(v + o + c);
        };
        [[maybe_unused]] auto plane_of_x = [&](const double *v, int p) -> mk_d2 {   // x: read by its owner only
            const int64_t o = p < 0 ? off_lo : (p > nz - 1 ? off_hi : (int64_t)p * P);
            if constexpr (GEN) return mk_pen_ld2<(FNT & 1) != 0>(v, o + c, xtop);
            else if constexpr (FNT & 1) return __builtin_nontemporal_load(reinterpret_cast<const mk_d2 *>(v + o + c));
            else return *reinterpret_cast<const mk_d2 *>(v + o + c);
        };
        // fuse: raw p_old of plane `pl` in pv with its r (and x) -> p (in pv) and x; both written out when the plane is one of
        // this chunk's own (anything else lands in the workgroup's dump rows: every store of the loop is unconditional)
        [[maybe_unused]] auto transform = [&](mk_d2 &pv, const mk_d2 rv, const mk_d2 xv, int pl, bool store, [[maybe_unused]] bool lv = true) {
            if constexpr (FUSE) {
                const mk_d2 po = pv;
                pv.x = epi.fuse_pnew(po.x, rv.x);
                pv.y = epi.fuse_pnew(po.y, rv.y);
                if (store) {                                 // (compile-time constant at every call site of the pipelined loop)
                    const bool own = pl >= z0 && pl < z1;    // (workgroup uniform: a scalar select of the address)
                    double *dump = epi.fuse_dump + (int64_t)blockIdx.x * 1024 + 2 * tid;
                    mk_d2 xn;
                    xn.x = epi.fuse_xnew(xv.x, po.x);
                    xn.y = epi.fuse_xnew(xv.y, po.y);
                    if constexpr (GEN) {
                        // the same targets; a lane writes its pair where both rows exist, a row alone where only the first does
                        // (odd L or P: divergent, rare), nothing otherwise; `lv`: the step is one of the chunk's own (not a
                        // discarded plane of the leftover round)
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
                        return;
                    }
                    // a slab's neighbour planes: the same p the neighbour forms for its own rows (same beta, same p_old and r
                    // bits) is kept behind the own rows of the new p buffer -- the next pass's p_old there; x is the neighbour's
                    double *pd = own ? epi.fuse_p + (int64_t)pl * P + c : dump;
                    pd = (pl == -1 && A.pen_xlo >= 0) ? epi.fuse_p + off_lo + c : pd;
                    pd = (pl == nz && A.pen_xhi >= 0) ? epi.fuse_p + off_hi + c : pd;
                    if constexpr (FNT & 4) __builtin_nontemporal_store(pv, reinterpret_cast<mk_d2 *>(pd));
                    else *reinterpret_cast<mk_d2 *>(pd) = pv;
                    mk_d2 *xd = reinterpret_cast<mk_d2 *>(own ? epi.fuse_x + (int64_t)pl * P + c : dump + 512);
                    if constexpr (FNT & 2) __builtin_nontemporal_store(xn, xd);
                    else *xd = xn;
                }
            }
        };
        [[maybe_unused]] auto halo_val = [&](double pv, double rv) -> double {
            if constexpr (FUSE) return epi.fuse_pnew(pv, rv);
            else return pv;
        };
        // one plane: slots (xm, xc, xp) = planes zz-1, zz, zz+1 at the own rows; hv / ev / pp = halo rows and pattern bytes of plane zz
        [[maybe_unused]] auto vals = [&](int p, int sl) {     // STREAM: the 14 values of the own rows of plane p
            if constexpr (SYM) {                             // diagonal, +1, +L, +P: slots 3 .. 6
                p = p > nz - 1 ? nz - 1 : p;
#pragma unroll
                for (int k = 3; k < 7; ++k) {                 // (GEN: any 8-byte boundary; the arrays have slack behind them)
                    if constexpr (GEN) vr[sl][k] = __builtin_nontemporal_load(reinterpret_cast<const mk_d2u *>(A.sval + (int64_t)(k - 3) * A.nrows + (int64_t)p * P + c));
                    else vr[sl][k] = __builtin_nontemporal_load(reinterpret_cast<const mk_d2 *>(A.sval + (int64_t)(k - 3) * A.nrows + (int64_t)p * P + c));
                }
            } else if constexpr (STREAM) {
                p = p > nz - 1 ? nz - 1 : p;
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    if constexpr (GEN) vr[sl][k] = __builtin_nontemporal_load(reinterpret_cast<const mk_d2u *>(A.sval + (int64_t)k * A.nrows + (int64_t)p * P + c));
                    else vr[sl][k] = __builtin_nontemporal_load(reinterpret_cast<const mk_d2 *>(A.sval + (int64_t)k * A.nrows + (int64_t)p * P + c));
                }
            }
        };
        auto step = [&](int zz, int bo, const mk_d2 xm_, const mk_d2 xc_, const mk_d2 xp_, double hv, double ev, unsigned pp,
                        const mk_d2 (&vv)[7], [[maybe_unused]] double hvv, [[maybe_unused]] double evv, auto &&reload, auto &&after,
                        [[maybe_unused]] bool live = true) {
            [[maybe_unused]] double oa[4], ob[4];             // the epilogue's own-row operands of rows c, c + 1 (<= 4 vectors)
            if constexpr (XPF || RPF) {
                static_assert(Epi::NPF <= 4, "at most four prefetched epilogue operands");
#pragma unroll
                for (int j = 0; j < Epi::NPF; ++j) {
                    const mk_d2 o2 = *reinterpret_cast<const mk_d2 *>(epi.pf_vec(j) + (int64_t)zz * P + c);
                    oa[j] = o2.x;
                    ob[j] = o2.y;
                }
            }
            mk_d2 xm, xc, xp;
            xm.x = epi.xin(xm_.x); xm.y = epi.xin(xm_.y);
            xc.x = epi.xin(xc_.x); xc.y = epi.xin(xc_.y);
            xp.x = epi.xin(xp_.x); xp.y = epi.xin(xp_.y);
            [[maybe_unused]] const int vbo = (bo / BUF) * MK_PEN_VB;
            if constexpr (SYM) {                             // own values now, the three lower ones after the barrier
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
                va[0] = mk_pen_sel(vlo.x, ma[0]);            // a(r, r - P) = the +P value of the same row one plane below
                vb[0] = mk_pen_sel(vlo.y, mb[0]);
                vb[2] = mk_pen_sel(vv[4].x, mb[2]);          // a(c + 1, c) = row c's +1 value
                vlo = vv[6];
                *reinterpret_cast<mk_d2 *>(vl_own + vbo) = vv[5];
                *reinterpret_cast<mk_d2 *>(vw_own + vbo) = vv[4];
                vl_halo[vbo] = hvv;
                vw_edge[vbo] = evv;
            } else if constexpr (STREAM) {                   // the byte IS the mask; the values arrived with the plane
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    va[k] = vv[k].x;
                    vb[k] = vv[k].y;
                    ma[k] = (unsigned)__builtin_amdgcn_sbfe((int)pp, k, 1);
                    mb[k] = (unsigned)__builtin_amdgcn_sbfe((int)pp, 8 + k, 1);
                }
            } else if (__builtin_amdgcn_ballot_w64(pp != pprev) != 0) {  // (wave uniform) a row of this wave follows another pattern now
                const mk_u4 *ta = reinterpret_cast<const mk_u4 *>(ptl + 16 * (pp & 0xffu));
                const mk_u4 *tb = reinterpret_cast<const mk_u4 *>(ptl + 16 * (pp >> 8));
                const mk_u4 a0 = ta[0], a1 = ta[1], a2 = ta[2], a3 = ta[3], b0 = tb[0], b1 = tb[1], b2 = tb[2], b3 = tb[3];
                const unsigned wa[16] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, a3.x, a3.y, a3.z, a3.w};
                const unsigned wb[16] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
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
            hdst[bo] = epi.xin(hv);
            edst[bo] = epi.xin(ev);
            reload();
            __syncthreads();
            const double *row = cdst + bo;
            const mk_d2 lo = *reinterpret_cast<const mk_d2 *>(row - RS), up = *reinterpret_cast<const mk_d2 *>(row + RS);
            const double we = row[-1], ea = row[2];
            if constexpr (SYM) {
                const mk_d2 vl = *reinterpret_cast<const mk_d2 *>(vl_own + vbo - 128);    // the line below this lane's
                va[1] = mk_pen_sel(vl.x, ma[1]);
                vb[1] = mk_pen_sel(vl.y, mb[1]);
                va[2] = mk_pen_sel(vw_own[vbo - 1], ma[2]);  // a(c, c - 1) = row c - 1's +1 value
            }
            const int64_t r = (int64_t)zz * P + c;
            const double na[7] = {xm.x, lo.x, we, xc.x, xc.y, up.x, xp.x}, nb[7] = {xm.y, lo.y, xc.x, xc.y, ea, up.y, xp.y};
            double sa = 0.0, sb = 0.0;
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                sa = mk_pen_term(sa, va[k], ma[k], na[k]);
                sb = mk_pen_term(sb, vb[k], mb[k], nb[k]);
            }
            if constexpr (GEN) {                              // rows that do not exist, planes past the chunk's end: discarded
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
                return;
            }
            if constexpr (PROG) {
                sa = mk_rowprog(A, sa, x, r, epi);
                sb = mk_rowprog(A, sb, x, r + 1, epi);
            }
            if constexpr (XPF) {
                epi.row_x_pf(r, sa, xc.x, oa, acc);
                epi.row_x_pf(r + 1, sb, xc.y, ob, acc);
            } else if constexpr (RPF) {
                epi.row_pf(r, sa, oa, acc);
                epi.row_pf(r + 1, sb, ob, acc);
            } else if constexpr (ROWX) {
                epi.row_x(r, sa, xc.x, acc);
                epi.row_x(r + 1, sb, xc.y, acc);
            } else {
                if constexpr (MkHasPre<Epi>::value) epi.pre(r);
                epi.row(r, sa, acc);
                if constexpr (MkHasPre<Epi>::value) epi.pre(r + 1);
                epi.row(r + 1, sb, acc);
            }
            after();
        };
        if constexpr (SYM) {                                  // the -P values of the chunk's first plane
            const double *src = z0 > 0 ? sv6 + (int64_t)(z0 - 1) * P + c : (A.pen_xlo >= 0 ? A.sval + 4 * A.nrows + c : sv6 + c);
            if constexpr (GEN) vlo = *reinterpret_cast<const mk_d2u *>(src);
            else vlo = *reinterpret_cast<const mk_d2 *>(src);
        }
        // planes of the pipelined rounds; the rest one by one below (GEN: one more round, its planes past z1 discarded)
        // (GEN, a chunk of one or two planes -- a slab's boundary launch: one plane after the other, one memory round trip
        //  each, instead of a whole round's pipeline fill for them)
        const int zfull = GEN ? (z1 - z0 > 2 ? z0 + ((z1 - z0 + R - 1) / R) * R : z0) : z0 + ((z1 - z0) / R) * R;
        if (zfull > z0) {
            [[maybe_unused]] mk_d2 rm1{0.0, 0.0}, r00{0.0, 0.0}, x00{0.0, 0.0};
            if constexpr (FUSE) {                             // r (and x) of the two planes the ring starts with
                rm1 = plane_of(epi.fuse_r, z0 - 1);
                r00 = plane_of(epi.fuse_r, z0);
                x00 = plane_of_x(epi.fuse_x, z0);
            }
#pragma unroll
            for (int d = 0; d < R - 1; ++d) {                 // (issue order = consumption order; slot R - 1 is loaded by step 0)
                ring[d] = plane(z0 - 1 + d);
                if (d < H) halo(z0 + d, d);
                if (d < VD) vals(z0 + d, d);
                if constexpr (FUSE) {
                    if (d < H) {                              // plane z0 + 1 + d -> slot (1 + d) % H
                        rr[(1 + d) % H] = plane_of(epi.fuse_r, z0 + 1 + d);
                        xx[(1 + d) % H] = plane_of_x(epi.fuse_x, z0 + 1 + d);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // one full wait at entry: the loop header's wait count is the stricter of the entry edge and the back edge,
            // and the entry edge as the compiler models it would drain the pipeline on every round
            __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0)
            if constexpr (FUSE) {
                transform(ring[0], rm1, rm1, z0 - 1, true);   // (the chunk below owns plane z0 - 1: formed, written only where
                                                              //  it is a slab's neighbour plane -- else into the dump rows)
                transform(ring[1], r00, x00, z0, true);
            }
            int z = z0;
            do {
#pragma unroll
                for (int d = 0; d < R; ++d) {
                    const int zz = z + d;
                    [[maybe_unused]] const bool live = !GEN || zz < z1;
                    if constexpr (FUSE) transform(ring[(d + 2) % R], rr[(d + 1) % H], xx[(d + 1) % H], zz + 1, true, live);
                    step(zz, (d & 1) * BUF, ring[d], ring[(d + 1) % R], ring[(d + 2) % R], halo_val(hreg[d % H], hr[FUSE ? d % H : 0]),
                         halo_val(ereg[d % H], er[FUSE ? d % H : 0]), pidr[d % H], vr[STREAM ? d % VD : 0], hvr[SYM ? d % H : 0],
                         evr[SYM ? d % H : 0], [&]() {
                             __builtin_amdgcn_sched_barrier(0);   // (the slots' last uses stay ABOVE their reloads)
                             ring[(d + R - 1) % R] = plane(zz + R - 2);   // the slot of plane zz - 2: dead since the last step
                             halo(zz + H, d % H);
                             if constexpr (FUSE) {            // plane zz + 1 + H into the slot plane zz + 1 just left
                                 rr[(d + 1) % H] = plane_of(epi.fuse_r, zz + 1 + H);
                                 xx[(d + 1) % H] = plane_of_x(epi.fuse_x, zz + 1 + H);
                             }
                         }, [&]() { vals(zz + VD, d % VD); }, live);   // (values: their slot is free once the row sums are formed)
                }
                z += R;
            } while (z < zfull);
        }
        for (int zz = zfull; zz < z1; ++zz) {                 // (<= R - 1 planes of the last chunk when nz is not a multiple of R)
            mk_d2 xm = plane(zz - 1), xc = plane(zz), xp = plane(zz + 1);
            halo(zz, 0);
            if constexpr (FUSE) {
                // planes zz - 1 and zz are formed again from p_old and r (x was updated when they were written: not touched);
                // a chunk without pipelined rounds writes its first plane here; plane zz + 1 is formed and written now
                const mk_d2 ra = plane_of(epi.fuse_r, zz - 1), rb = plane_of(epi.fuse_r, zz), rc = plane_of(epi.fuse_r, zz + 1);
                const mk_d2 xb = plane_of_x(epi.fuse_x, zz), xcn = plane_of_x(epi.fuse_x, zz + 1);
                if (zz == z0) transform(xm, ra, ra, zz - 1, true);   // (written only as a slab's neighbour plane, see above)
                else transform(xm, ra, ra, zz - 1, false);
                if (zz == z0) transform(xc, rb, xb, zz, true);
                else transform(xc, rb, rb, zz, false);
                transform(xp, rc, xcn, zz + 1, true);
            }
            vals(zz, 0);
            step(zz, ((zz - zfull) & 1) * BUF, xm, xc, xp, halo_val(hreg[0], hr[0]), halo_val(ereg[0], er[0]), pidr[0], vr[0], hvr[0], evr[0],
                 [] {}, [] {});
        }
        __syncthreads();                                     // the next item's first plane image overwrites this LDS
    }
}
