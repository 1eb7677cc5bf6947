"""The brick march (storage formats 9 / 10, csrc/mk_spmv_fmt9.h) on ONE RANK'S SLAB of a z-partitioned 7-point matrix:
columns localised to [own | plane below | plane above] (mk_csr_localize mode 0), the neighbours' planes taken from the
received entries (a row keeps the storage order of its global columns, so the sums are those of one device).  One
process plays rank `rank` of `nr` through the host-staged transport in a loopback (the planes it would send come back as
the planes it would receive: a z-periodic slab), as tools/slab_budget.py does for the 512^3 run.

  * the product of the slab is BIT-identical to the oracle's scalar loop over the localised CSR arrays;
  * under an overlapped halo exchange the product runs as interior planes + boundary planes (two launches): CG on the
    slab follows the windowed formats' run to rounding (the rows are the same bits, the dots group differently)."""
import ctypes

import numpy as np
import pytest

from oracle import csr_ref

pytestmark = pytest.mark.gpu


def loopback_world(nr, rank, ranges, plane):
    from pykrylov_amd import _lib, dist

    class World(dist.World):
        def allgather_object(self, obj):
            assert isinstance(obj, tuple) and len(obj) == 4, obj    # (c0, c1, halo_lo, halo_hi) of partition_poisson3d
            return [(c0, c1, plane if r > 0 else 0, plane if r < nr - 1 else 0) for r, (c0, c1) in enumerate(ranges)]

    def view(ptr, count):
        return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_double)), shape=(count,))

    def allreduce(buf, count):
        return 0

    def exchange(send, send_count, send_off, recv, recv_count, recv_off):
        nb = [r for r in (rank - 1, rank + 1) if 0 <= r < nr]
        for dst in nb:                                       # what I would receive from `dst`: the plane I send the other way
            src = [r for r in nb if r != dst] or [dst]
            cnt = recv_count[dst]
            assert cnt == send_count[src[0]] == plane
            view(recv, recv_off[dst] + cnt)[recv_off[dst]:] = view(send, send_off[src[0]] + cnt)[send_off[src[0]]:]
        return 0

    def allgather(send, count, recv):
        return 1

    w = World(rank, nr, None)
    w._cbs = (_lib.HOST_ALLREDUCE_FN(allreduce), _lib.HOST_EXCHANGE_FN(exchange), _lib.HOST_ALLGATHER_FN(allgather))
    return w


def build_slab(nx, ny, nz, nr, rank, varcoef, fmt):
    from pykrylov_amd import _lib, dist
    lib = _lib.init(0)
    ranges = dist.row_ranges(nx * ny * nz, nr, align=nx * ny)
    world = loopback_world(nr, rank, ranges, nx * ny)
    _lib.check(lib.mk_comm_init_host(nr, rank, *world._cbs))
    op, _ = dist.partition_poisson3d(world, nx, ny, nz, mode="halo", varcoef_seed=5 if varcoef else None)
    _lib.check(lib.mk_csr_set_format(op.handle, fmt))
    return lib, world, op


def fmt_of(lib, op):
    from pykrylov_amd import _lib
    fmt = ctypes.c_int32()
    _lib.check(lib.mk_csr_format_info(op.handle, ctypes.byref(fmt), None, None, None, None))
    return fmt.value


def local_oracle(op):
    ip, ix, dat = op.to_csr_arrays()
    return csr_ref.RefCsr(ip, ix, dat, (int(op.shape[0]), int(op.shape[1])))


# (nx, ny): whole aligned bricks, and a general geometry (round 6: lines of 100 rows, planes of 9 lines -- partly empty bricks,
# the general-geometry kernels on the slab, its boundary planes run one by one)
GRIDS = [(128, 8), (100, 9)]


@pytest.mark.parametrize("nx,ny", GRIDS)
@pytest.mark.parametrize("varcoef,fmt", [(False, 9), (True, 10), (True, 11)])
@pytest.mark.parametrize("nr,rank,planes", [(4, 0, 7), (4, 2, 7), (4, 3, 7), (3, 1, 20), (2, 0, 2)])
def test_slab_product_bit_exact(varcoef, fmt, nr, rank, planes, nx, ny):
    from pykrylov_amd import _lib
    lib, world, op = build_slab(nx, ny, planes * nr, nr, rank, varcoef, fmt)
    try:
        n_local, ncols = int(op.shape[0]), int(op.shape[1])
        assert n_local == planes * nx * ny and ncols == n_local + nx * ny * ((rank > 0) + (rank < nr - 1))
        A = local_oracle(op)
        rng = np.random.default_rng(11 + rank)
        for trial in range(2):
            x = rng.standard_normal(ncols)
            if trial:
                x[::7] = 0.0
                x[5::11] *= 1e300
            xd = _lib.DeviceArray.from_numpy(x)
            yd = _lib.DeviceArray(n_local)
            op.spmv_device(xd.ptr, yd.ptr)
            assert fmt_of(lib, op) == fmt
            assert np.array_equal(yd.to_numpy(), A.matvec(x))
    finally:
        op.free()
        lib.mk_comm_destroy()


@pytest.mark.parametrize("nx,ny", GRIDS)
@pytest.mark.parametrize("varcoef,fmt,fmt_ref", [(False, 9, 4), (True, 10, 5), (True, 11, 5)])
@pytest.mark.parametrize("nr,rank,planes", [(4, 1, 20), (4, 0, 14), (4, 3, 13), (4, 2, 7), (4, 1, 23)])
def test_cg_on_the_slab_two_launch_product(varcoef, fmt, fmt_ref, nr, rank, planes, nx, ny):
    """planes = 20 / 14 / 13: interior + boundary launches (both neighbours, upper only, lower only); 7: too few planes to
    split, the messages are waited for first."""
    from pykrylov_amd import _lib
    from pykrylov_amd.generic import DeviceRun
    res = {}
    for f in (fmt, fmt_ref):
        lib, world, op = build_slab(nx, ny, planes * nr, nr, rank, varcoef, f)
        try:
            ni, nb = ctypes.c_int64(), ctypes.c_int64()
            _lib.check(lib.mk_csr_overlap_info(op.handle, ctypes.byref(ni), ctypes.byref(nb)))
            assert ni.value > 0 and nb.value > 0                # the exchange is overlapped: products run in two parts
            n_local = int(op.shape[0])
            rng = np.random.default_rng(5)
            rhs = _lib.DeviceArray.from_numpy(rng.standard_normal(n_local))
            run = DeviceRun(op, _lib.MK_CG, rhs, None, abstol=0.0, reltol=0.0, matvec_max=1 << 60, check_curvature=1)
            run.setup()
            assert fmt_of(lib, op) == f
            assert run.iterate(25) == 25
            run.finish()
            res[f] = (run.history(), run.x())
            run.close()
        finally:
            op.free()
            lib.mk_comm_destroy()
    (h, x), (h0, x0) = res[fmt], res[fmt_ref]
    assert len(h) == len(h0) and np.all(np.isfinite(h))
    assert np.max(np.abs(h - h0) / h0) <= 1e-12
    assert np.linalg.norm(x - x0) <= 1e-12 * np.linalg.norm(x0)


@pytest.mark.parametrize("nx,ny", GRIDS)
@pytest.mark.parametrize("varcoef,fmt", [(False, 9), (True, 10), (True, 11)])
@pytest.mark.parametrize("nr,rank,planes", [(4, 1, 20), (4, 0, 14), (4, 3, 13), (4, 2, 7), (2, 1, 3), (4, 1, 23)])
def test_fused_cg_passes_on_the_slab_change_no_bit(varcoef, fmt, nr, rank, planes, nx, ny, monkeypatch):
    """CG on a slab of the march runs FUSED passes too (csrc/mk_cg.hip): the neighbours' planes of p are formed on the spot
    from their p_old (kept behind the own rows of the p buffers) and their r, so it is r's boundary planes that travel.
    Same operations on the same values, same launches and workgroup shares: history, iterate, residual vector and search
    direction equal the three-kernel pass on the same slab bit for bit -- through interior + boundary launches (20 / 14 /
    13 planes; 23: an interior of 21 planes, three whole rounds and a masked one) and through the whole-slab launch (7 / 3 planes).
    Round 6: only the slab's first and last plane wait for the messages (run one plane at a time), on both geometries."""
    from pykrylov_amd import _lib
    from pykrylov_amd.generic import DeviceRun
    res = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("MK_CG_FUSE", fuse)
        lib, world, op = build_slab(nx, ny, planes * nr, nr, rank, varcoef, fmt)
        try:
            n_local = int(op.shape[0])
            rng = np.random.default_rng(5)
            rhs = _lib.DeviceArray.from_numpy(rng.standard_normal(n_local))
            guess = _lib.DeviceArray.from_numpy(rng.standard_normal(n_local))
            for g in (None, guess):
                run = DeviceRun(op, _lib.MK_CG, rhs, g, abstol=0.0, reltol=0.0, matvec_max=1 << 60, check_curvature=1)
                run.setup()
                f = ctypes.c_int32(-1)
                _lib.check(lib.mk_solver_fused(run.handle, ctypes.byref(f)))
                assert bool(f.value) == (fuse == "1") and fmt_of(lib, op) == fmt
                assert run.iterate(5) == 5
                mid = run.x()                                   # the iterate between passes: the pending update formed aside
                assert run.iterate(16) == 16
                run.finish()
                res[fuse, g is None] = (run.history(), mid, run.x(), run.vector(0)[:n_local], run.vector(1)[:n_local])
                run.close()
        finally:
            op.free()
            lib.mk_comm_destroy()
    for key in ((True,), (False,)):
        a, b = res[("1",) + key], res[("0",) + key]
        assert np.all(np.isfinite(a[0])) and len(a[0]) == 22
        for u, v in zip(a, b):
            assert np.array_equal(u, v)
