"""NumPy restatement of the device's fixed summation trees.   TEST INFRASTRUCTURE ONLY.

The device never sums in NumPy's (pairwise / BLAS) order, which is the only source of
difference between the GPU solvers and the oracle (all other arithmetic rounds identically).
Restating the tree lets tests demand BIT equality instead of a tolerance:

stream kernels (csrc/mk_device.h `mk_stream_kernel`):
    lane g of S = grid*256 lanes adds elements 2q, 2q+1 for q = g, g+S, ... in order (the odd
    tail element goes to lane (n//2) % S); then per wave64 a shuffle-down tree 32,16,...,1; the 4
    wave sums of a workgroup are added in order; workgroup partials are added by `mk_total`:
    lane t takes partials t, t+256, ... then the same workgroup tree.
"""
import numpy as np

BLOCK = 256
MAXP = 512          # default grid cap of the streaming kernels (MK_GRID_STREAM)
SPMV_CAP = 2048     # SpMV grid cap for cache-resident matrices (2 x MK_GRID_SPMV; XCD-chunked tile order);
                    # matrices beyond ~200 MB use 1024 workgroups and a plain round-robin (csrc/mk_device.h)


def _wave_tree(v):
    """v: (..., 64) -> lane-0 result of the shuffle-down tree with offsets 32..1."""
    v = v.copy()
    for off in (32, 16, 8, 4, 2, 1):
        shifted = np.concatenate([v[..., off:], v[..., -off:] * 0 + v[..., -off:]], axis=-1)
        # lanes >= 64-off read their own value; only lane 0's chain matters
        v = v + shifted
    return v[..., 0]


def _block_sum(vals):
    """vals: (nblocks, 256) per-lane values -> (nblocks,) workgroup sums in device order."""
    w = _wave_tree(vals.reshape(vals.shape[0], 4, 64))          # (nblocks, 4)
    return ((w[:, 0] + w[:, 1]) + w[:, 2]) + w[:, 3]


def total(partials):
    """mk_total: lane t adds slots t, t+256, ... sequentially, then the workgroup tree."""
    p = np.asarray(partials, dtype=np.float64)
    padded = np.zeros(((len(p) + BLOCK - 1) // BLOCK) * BLOCK)
    padded[:len(p)] = p
    lanes = np.zeros(BLOCK)
    for chunk in padded.reshape(-1, BLOCK):
        lanes = lanes + chunk
    return float(_block_sum(lanes[None, :])[0])


def grid_stream(n):
    g = max(1, (n + 2 * BLOCK - 1) // (2 * BLOCK))
    return min(g, MAXP)


def stream_partials(a, b):
    """Per-workgroup partial sums of sum(a*b) as mk_stream_kernel<MkOpDot> produces them."""
    n = len(a)
    grid = grid_stream(n)
    S = grid * BLOCK
    prod = np.asarray(a, dtype=np.float64) * np.asarray(b, dtype=np.float64)
    npair = n // 2
    acc = np.zeros(S)
    steps = (npair + S - 1) // S if npair else 0
    for s in range(steps):
        q0 = s * S
        cnt = min(S, npair - q0)
        seg = prod[2 * q0: 2 * (q0 + cnt)].reshape(cnt, 2)
        acc[:cnt] = acc[:cnt] + seg[:, 0]
        acc[:cnt] = acc[:cnt] + seg[:, 1]
    if n & 1:
        g = npair % S
        acc[g] = acc[g] + prod[n - 1]
    return _block_sum(acc.reshape(grid, BLOCK))


def stream_dot(a, b):
    if len(a) == 0:
        return 0.0
    return total(stream_partials(a, b))


def grid_spmv(ntiles):
    """Grid of the SpMV kernels for matrices with at most 1024 tiles (every format caps the grid at >= 1024
    workgroups); beyond that the grid depends on the storage format -- ask the library (mk_csr_launch_info)."""
    g = min(max(1, ntiles), SPMV_CAP)
    if g >= 8:
        g -= g % 8
    return g


def spmv_tile_order(ntiles, grid=None, tile_map=1):
    """For every workgroup the list of tiles it processes, in order (csrc/mk_device.h `mk_spmv_tiles`).
    tile_map 0: round robin.  1: XCD b % 8 owns a contiguous eighth of the tiles, dealt round-robin to its
    workgroups (cache-resident matrices).  2: every step of the grid is cut into eight XCD-contiguous blocks.
    (3, S, _): stripes of S consecutive tiles dealt round-robin to the XCDs, each XCD's tiles dealt to its workgroups.
    (4, S, P): as 3, but an XCD walks its strip through all ntiles / P planes (P tiles apart) before its next strip.
    Maps 1 to 4 fall back to 0 when the grid is not a multiple of 8."""
    if grid is None:
        grid = grid_spmv(ntiles)
    x8 = grid % 8 == 0
    order = []
    for b in range(grid):
        if tile_map == 1 and x8:
            per = grid // 8
            chunk = (ntiles + 7) // 8
            c0 = (b % 8) * chunk
            order.append(range(c0 + b // 8, min(c0 + chunk, ntiles), per))
        elif tile_map == 2 and x8:
            order.append(range((b % 8) * (grid // 8) + b // 8, ntiles, grid))
        elif isinstance(tile_map, tuple) and tile_map[0] == 3 and x8:   # stripes of S tiles dealt round-robin to the XCDs
            S, e = tile_map[1], b % 8
            nst = (ntiles + S - 1) // S
            cnt = (nst - e + 7) // 8 if nst > e else 0
            end = cnt * S - ((nst * S - ntiles) if (cnt > 0 and (nst - 1) % 8 == e) else 0)
            order.append([((p // S) * 8 + e) * S + p % S for p in range(b // 8, end, grid // 8)])
        elif isinstance(tile_map, tuple) and tile_map[0] == 4 and x8:   # ... an XCD walks a strip through all planes first
            S, P, e = tile_map[1], tile_map[2], b % 8
            NP = ntiles // P
            order.append([((p // S) % NP) * P + (((p // S) // NP) * 8 + e) * S + p % S
                          for p in range(b // 8, ntiles // 8, grid // 8)])
        else:
            order.append(range(b, ntiles, grid))
    return order


def pencil_partials(w, y, grid, L, P, nz, zc, chunks, gen=0, per=-1):
    """Brick march (csrc/mk_spmv_fmt9.h; storage formats 9 / 10 / 11): workgroup b takes the (brick, chunk) items b, b + grid,
    ...  A plane of P rows is cut into lines of L rows, a line into bx = ceil(L / 128) bricks, lines are grouped in fours:
    bpp = bx * ceil(ceil(P / L) / 4) bricks per plane, brick j starting at row (j // bx) 4 L + (j % bx) 128 of a plane.  With
    `per` > 0 and a grid that is a multiple of 8 the items are dealt XCD-contiguously: item i is brick (i % 8) per +
    (i // 8) % per of chunk (i // 8) // per -- 8 per item slots per chunk, a slot whose brick number is >= bpp is empty --,
    otherwise brick i % bpp of chunk i // bpp (per < 0: round 5's rule, per = bpp / 8 when that is whole).  Lane (wave v,
    lane l) owns the rows z P + c and z P + c + 1, c = brick start + v L + 2 l, and adds their terms plane by plane through
    the chunk, row c first; on a general geometry (`gen`) a row that does not exist -- in-line position >= L or in-plane
    index >= P -- adds +0.0, which leaves the running sum unchanged bit for bit."""
    pad = 4 * L + 256
    prod = np.zeros((nz, P + pad))
    prod[:, :P] = (np.asarray(w, dtype=np.float64) * np.asarray(y, dtype=np.float64)).reshape(nz, P)
    bx = (L + 127) // 128
    bpp = bx * (((P + L - 1) // L + 3) // 4)
    if per < 0:
        per = bpp // 8 if bpp % 8 == 0 else 0
    xdeal = per > 0 and grid % 8 == 0
    items = (8 * per if xdeal else bpp) * chunks
    lane_c = (np.arange(4)[:, None] * L + 2 * np.arange(64)[None, :]).reshape(-1)       # lane t = 64 v + l
    lane_x = np.tile(2 * np.arange(64), 4)                                              # position in the brick's 128 columns
    acc = np.zeros((grid, BLOCK))
    for first in range(0, items, grid):                      # round k of every workgroup (vectorised over workgroups)
        it = np.arange(first, min(first + grid, items))
        wg = np.arange(len(it))
        if xdeal:                                            # XCD-contiguous deal (workgroup b runs on XCD b % 8)
            bi, ch = (it % 8) * per + (it // 8) % per, (it // 8) // per
            keep = bi < bpp
            it, wg, bi, ch = it[keep], wg[keep], bi[keep], ch[keep]
        else:
            bi, ch = it % bpp, it // bpp
        if len(it) == 0:
            continue
        b0 = (bi // bx) * 4 * L + (bi % bx) * 128
        cols = b0[:, None] + lane_c[None, :]                 # (workgroups, 256): in-plane index of row c
        cx = (bi % bx)[:, None] * 128 + lane_x[None, :]
        oka = (cx < L) & (cols < P)
        okb = (cx + 1 < L) & (cols + 1 < P)
        cols = np.minimum(cols, P + pad - 2)
        for zi in range(zc):
            z = ch * zc + zi
            live = z < nz
            if not live.any():
                break
            rows = wg[live]
            ta = np.where(oka[live], prod[z[live][:, None], cols[live]], 0.0)
            tb = np.where(okb[live], prod[z[live][:, None], cols[live] + 1], 0.0)
            acc[rows] = acc[rows] + ta
            acc[rows] = acc[rows] + tb
    return _block_sum(acc)


def spmv_partials(w, y, ntiles, grid=None, tile_map=1):
    """Per-workgroup partial sums of sum(w*y) when the dot is fused into the SpMV kernel: lane t of a
    workgroup owns row 256*tile + t of each of its tiles and adds w[r]*y[r] in tile order."""
    if isinstance(tile_map, tuple) and tile_map[0] == "pencil":
        return pencil_partials(w, y, grid, *tile_map[1:])
    n = len(w)
    prod = np.zeros(ntiles * BLOCK)
    prod[:n] = np.asarray(w, dtype=np.float64) * np.asarray(y, dtype=np.float64)
    prod = prod.reshape(ntiles, BLOCK)
    order = spmv_tile_order(ntiles, grid, tile_map)
    acc = np.zeros((len(order), BLOCK))
    steps = max(len(t) for t in order) if order else 0
    for k in range(steps):                                   # vectorised over workgroups: step k of every workgroup
        idx = np.array([t[k] if k < len(t) else -1 for t in order])
        live = idx >= 0
        acc[live] = acc[live] + prod[idx[live]]
    return _block_sum(acc)


def launch_geometry(op):
    """(grid, tile_map) of the SpMV kernels of a device operator, from the library (TEST helper: lets the
    emulation follow the real grid of large matrices, which depends on the storage format)."""
    import ctypes
    from pykrylov_amd import _lib
    g, m = ctypes.c_int32(), ctypes.c_int32()
    _lib.check(_lib.init().mk_csr_launch_info(op.handle, ctypes.byref(g), ctypes.byref(m)))
    info = (ctypes.c_int64 * 12)()
    _lib.check(_lib.init().mk_csr_march_info(op.handle, info, 12))
    if info[0]:                                               # storage formats 9 / 10 / 11: brick march instead of 256-row tiles
        return g.value, ("pencil", info[1], info[2], info[3], info[7], info[8], info[9], info[10])
    if m.value in (3, 4):                                     # (these orders have parameters)
        o, s, p = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        _lib.check(_lib.init().mk_csr_tile_order(op.handle, ctypes.byref(o), ctypes.byref(s), ctypes.byref(p), None))
        return g.value, (o.value, s.value, p.value)
    return g.value, m.value


class GpuDots(object):
    """`dot_impl` for krylov_ref.Reductions reproducing the device's summation order: call sites
    listed in `spmv_sites` are fused into an SpMV kernel (row-owner order), all others are streaming
    kernels.  `geometry` = (grid, tile_map) of the operator's SpMV launches (default: small-matrix rule)."""

    def __init__(self, n, spmv_sites, geometry=None):
        self.ntiles = (n + BLOCK - 1) // BLOCK
        self.spmv_sites = set(spmv_sites)
        self.grid, self.tile_map = geometry if geometry else (None, 1)

    def __call__(self, a, b, site):
        if site in self.spmv_sites:
            return total(spmv_partials(a, b, self.ntiles, self.grid, self.tile_map))
        return stream_dot(a, b)


class ExactDots(object):
    """`dot_impl` for krylov_ref.Reductions with ORDER-INDEPENDENT inner products: products and sums are formed in
    extended precision (x87 long double, 64-bit significand; NumPy's pairwise summation) and rounded to double once.
    The result is within 1e-18 (relative, for sums of like-signed terms; ~1e-18 x condition otherwise) of the exactly
    rounded inner product -- four orders of magnitude below the 1e-12 parity bar -- so a history computed with it is
    the anchor both the device's tree order and np.dot's BLAS order can be measured against."""

    def __init__(self):
        assert np.finfo(np.longdouble).nmant >= 63, "needs x87 extended precision"

    def __call__(self, a, b, site):
        a = np.asarray(a, dtype=np.float64).astype(np.longdouble)
        if b is a:
            return float(np.sum(a * a))
        return float(np.sum(a * np.asarray(b, dtype=np.float64).astype(np.longdouble)))


SPMV_SITES = {
    "cg": ["cg.pAp"],
    "bicgstab": ["bicgstab.r0v", "bicgstab.ts", "bicgstab.tt", "bicgstab.r0t"],
}
SPMV_SITES["cgs"] = ["cgs.sigma", "cgs.r", "cgs.rho"]
SPMV_SITES["tfqmr"] = ["tfqmr.sigma", "tfqmr.w2", "tfqmr.rho"]
SPMV_SITES["minres"] = ["minres.alfa"]
SPMV_SITES["symmlq"] = ["symmlq.alfa"]
SPMV_SITES["symmlq"] = ["symmlq.alfa1", "symmlq.alfa", "symmlq.rnorm"]
for _s in ("lsqr", "lsmr", "craig", "craigmr"):
    SPMV_SITES[_s] = [_s + ".beta", _s + ".alpha", _s + ".alpha0"]
