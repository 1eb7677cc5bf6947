"""Round-4 additions to the boundary and the scattered-product paths: row-range download, per-product kernel timing, the
vector arena, automatic column blocks, the pair-of-tiles kernel -- each through the C ABI, products bit for bit against the
oracle's scalar CSR loop."""
import ctypes

import numpy as np
import pytest

from oracle import csr_ref, krylov_ref as kr

pytestmark = pytest.mark.gpu


def op_from(A, **kw):
    from pykrylov_amd import CsrOperator
    return CsrOperator(A.indptr, A.indices, A.data, A.shape, **kw)


def fmt_of(op):
    from pykrylov_amd import _lib
    fmt, chunks = ctypes.c_int32(), ctypes.c_int32()
    _lib.check(_lib.init().mk_csr_format_info(op.handle, ctypes.byref(fmt), None, ctypes.byref(chunks), None, None))
    return fmt.value, chunks.value


def test_download_rows_matches_whole_download_and_rejects_bad_ranges():
    from pykrylov_amd import _lib
    A = csr_ref.random_diagdom(5000, seed=4)
    op = op_from(A)
    ip, ix, dv = op.to_csr_arrays()
    for a, b in ((0, 5000), (0, 0), (4999, 5000), (17, 1234), (2500, 2500)):
        p, i, v = op.csr_rows(a, b)
        assert np.array_equal(p, ip[a:b + 1] - ip[a]) and np.array_equal(i, ix[ip[a]:ip[b]]) and np.array_equal(v, dv[ip[a]:ip[b]])
    lib = _lib.init()
    buf = np.empty(8, dtype=np.int32)
    for a, b in ((-1, 3), (3, 2), (0, 5001)):
        assert lib.mk_csr_download_rows(op.handle, a, b, buf.ctypes.data, None, None) == -2        # MK_ERR_ARG
    op.free()


def test_time_product_every_loop_and_refuses_products_a_loop_does_not_have():
    """mk_solver_time_product: product 0 of every square loop, product 1 where the pass has two; MK_ERR_ARG otherwise."""
    from pykrylov_amd import _lib, gallery
    from pykrylov_amd.generic import DeviceRun
    lib = _lib.init()
    op = gallery.poisson2d(64)
    n = op.shape[0]
    rhs = op * np.ones(n)
    two = {_lib.MK_BICGSTAB, _lib.MK_CGS, _lib.MK_TFQMR}
    for kind, kw in ((_lib.MK_CG, dict(abstol=0.0, reltol=0.0, matvec_max=50)),
                     (_lib.MK_BICGSTAB, dict(abstol=0.0, reltol=1e-3, matvec_max=50)),
                     (_lib.MK_CGS, dict(abstol=0.0, reltol=1e-3, matvec_max=50)),
                     (_lib.MK_TFQMR, dict(abstol=0.0, reltol=1e-3, matvec_max=50)),
                     (_lib.MK_MINRES, dict(itnlim=50, rtol=0.0, etol=0.0, window=5)),
                     (_lib.MK_SYMMLQ, dict(matvec_max=50, rtol=0.0))):
        run = DeviceRun(op, kind, rhs, None, **kw)
        run.setup()
        run.iterate(3)
        assert run.time_product(0, 5) > 0.0
        avg = ctypes.c_double()
        rc1 = lib.mk_solver_time_product(run.handle, 1, 5, ctypes.byref(avg))
        assert (rc1 == 0 and avg.value > 0) if kind in two else rc1 == -2, (kind, rc1)
        assert lib.mk_solver_time_product(run.handle, 2, 5, ctypes.byref(avg)) == -2
        run.close()
    A = csr_ref.random_diagdom(3000, seed=2)
    T = op_from(A)
    At = T.T
    run = DeviceRun(T, _lib.MK_LSQR, A.matvec(np.ones(3000)), None, transpose=At, itnlim=20, damp=0.0, atol=0.0, btol=0.0,
                    conlim=0.0, etol=0.0, window=5)
    run.setup()
    run.iterate(2)
    assert run.time_product(0, 3) > 0 and run.time_product(1, 3) > 0
    run.close()
    T.free()
    op.free()


def test_vector_arena_gives_the_same_bits_and_is_guarded():
    from pykrylov_amd import CG, _lib, gallery
    lib = _lib.init()
    op = gallery.poisson3d_varcoef(24, seed=3)
    n = op.shape[0]
    rhs = op * np.ones(n)
    s = CG(op)
    s.solve(rhs)
    want = (s.nMatvec, np.array(s.residHistory), s.x.copy())
    _lib.check(lib.mk_arena_reserve(64 << 20))
    from pykrylov_amd.generic import DeviceRun
    run = DeviceRun(op, _lib.MK_CG, rhs, None, abstol=1e-8, reltol=1e-6, matvec_max=2 * n, check_curvature=1)
    res = run.run()                                             # (its vectors come from the arena)
    assert lib.mk_arena_reserve(1 << 20) == -3                  # MK_ERR_STATE: vectors of the arena are in use
    assert res.nMatvec == want[0] and np.array_equal(run.history(), want[1]) and np.array_equal(run.x(), want[2])
    run.close()
    s2 = CG(op)
    s2.solve(rhs)                                               # the arena is reusable once its vectors are back
    assert np.array_equal(s2.x, want[2])
    _lib.check(lib.mk_arena_reserve(0))
    s3 = CG(op)
    s3.solve(rhs)
    assert np.array_equal(s3.x, want[2])
    op.free()


def ragged_short_rows(m, ncols, seed):
    """rows of 0 .. 5 entries at scattered columns (sorted, distinct), some rows empty"""
    rng = np.random.default_rng(seed)
    length = rng.integers(0, 6, size=m)
    length[rng.integers(0, m, size=m // 50)] = 0
    w = ncols // 5
    cols = (rng.integers(0, w, size=(m, 5)) + np.arange(5)[None, :] * w)
    keep = np.arange(5)[None, :] < length[:, None]
    indptr = np.concatenate([[0], np.cumsum(length)])
    return csr_ref.RefCsr(indptr, cols[keep], rng.standard_normal(int(length.sum())), (m, ncols))


def test_pair_kernel_ragged_rows_bit_exact():
    """More tiles than resident workgroups, rows of <= 5 entries, x longer than an L2: format 3 takes its tiles in pairs, the
    second in registers (mk_spmv_fmt3r.h).  Ragged and empty rows, an odd tile count, a last tile that is not full."""
    from pykrylov_amd import _lib
    m, ncols = 2049 * 256 + 77, 700001
    A = ragged_short_rows(m, ncols, seed=8)
    op = op_from(A)
    x = np.random.default_rng(1).standard_normal(ncols)
    y = op * x
    assert fmt_of(op)[0] == 3
    assert np.array_equal(y, A.matvec(x))
    # the same rows through the gather path: same bits
    lib = _lib.init()
    _lib.check(lib.mk_csr_set_format(op.handle, 0))
    assert fmt_of(op)[0] == 0 and np.array_equal(op * x, y)
    op.free()


def test_pair_kernel_in_a_solver_loop_matches_the_oracle():
    """BiCGSTAB on a random diagonally dominant matrix large enough for the pair kernel (n = 700 000: x = 5.6 MB, 2 735 tiles): the loop's fused dots
    ride in the product epilogues, so counts, residual and iterate agree with the oracle."""
    from pykrylov_amd import BiCGSTAB
    n = 700000
    A = csr_ref.random_diagdom(n, seed=6)
    rhs = A.matvec(np.ones(n))
    ref = kr.bicgstab(A, rhs, reltol=1e-10)
    op = op_from(A)
    s = BiCGSTAB(op, reltol=1e-10)
    s.solve(rhs)
    assert fmt_of(op)[0] == 3
    assert abs(s.nMatvec - ref["nMatvec"]) <= 1 and s.converged
    # (np.dot order against the device's trees: the stopping test at reltol 1e-10 may fall one product apart, and with it the
    # last update of x; the order-independent statement for this matrix class is tests/test_gpu_anchors.py)
    assert np.linalg.norm(s.x - ref["x"]) <= 1e-10 * np.linalg.norm(ref["x"])
    op.free()


def test_automatic_column_blocks_for_long_rows_over_a_long_x():
    """>= 12 entries per row gathered from >= 16 MB of x (the transposed operator of a tall least-squares problem): the product
    runs in column blocks of 8 MiB of x (4 MiB if a tile does not fit LDS) with carried row sums -- the same left-to-right sums, the same bits."""
    from pykrylov_amd import _lib
    lib = _lib.init()
    m, ncols, k = 60000, 2200000, 16
    rng = np.random.default_rng(5)
    w = ncols // k
    cols = (rng.integers(0, w, size=(m, k)) + np.arange(k)[None, :] * w).reshape(-1)
    A = csr_ref.RefCsr(np.arange(m + 1) * k, cols, rng.standard_normal(m * k), (m, ncols))
    op = op_from(A)
    x = rng.standard_normal(ncols)
    y = op * x
    nb = ctypes.c_int32()
    _lib.check(lib.mk_csr_colblocks(op.handle, ctypes.byref(nb)))
    assert nb.value >= 2, nb.value                              # 17.6 MB of x in 8 MiB blocks (4 MiB if a tile outgrows LDS)
    assert np.array_equal(y, A.matvec(x))
    _lib.check(lib.mk_csr_set_colblocks(op.handle, 0))          # off: one launch
    _lib.check(lib.mk_csr_colblocks(op.handle, ctypes.byref(nb)))
    assert nb.value == 0 and np.array_equal(op * x, y)
    # short rows do not trigger it (measured slower there)
    B = ragged_short_rows(300000, ncols, seed=2)
    opb = op_from(B)
    assert np.array_equal(opb * x, B.matvec(x))
    _lib.check(lib.mk_csr_colblocks(opb.handle, ctypes.byref(nb)))
    assert nb.value == 0
    opb.free()
    op.free()


@pytest.mark.parametrize("solver", ["bicgstab", "tfqmr"])
def test_stepped_product_n2300000_bit_exact(solver):
    """8 985 tiles = 2.2 pair steps per workgroup, tile order 0: the product runs as ONE LAUNCH PER STEP (the workgroups are
    re-aligned by the kernel boundary, the per-lane accumulators of the fused dots travel through a carry buffer).  Same
    additions in the same order: counts, residual norms and the iterate bit for bit against the oracle run in the device's
    summation order -- the statement of tests/test_gpu_bitexact_full.py for a matrix large enough to be stepped."""
    from pykrylov_amd import BiCGSTAB, TFQMR, CsrOperator, gallery
    from oracle import gpu_order
    indptr, indices, data, shape = gallery.random_diagdom_csr(2300000, seed=3)
    n = shape[0]
    op = CsrOperator(indptr, indices, data, shape)
    A = csr_ref.RefCsr(indptr, indices, data, shape)
    x = np.random.default_rng(2).standard_normal(n)
    assert np.array_equal(op * x, A.matvec(x))
    rhs = op * np.ones(n)
    geo = gpu_order.launch_geometry(op)
    assert fmt_of(op)[0] == 3 and geo == (2048, 0) and (n + 255) // 256 > 2 * 2048 * 2
    cls, fn = (BiCGSTAB, kr.bicgstab) if solver == "bicgstab" else (TFQMR, kr.tfqmr)
    s = cls(op, reltol=1e-10)
    s.solve(rhs)
    ref = fn(A, rhs, reltol=1e-10, red=kr.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES[solver], geo)))
    assert s.nMatvec == ref["nMatvec"] and s.converged == ref["converged"]
    assert s.residNorm0 == ref["residNorm0"] and s.residNorm == ref["residNorm"]
    assert np.array_equal(s.x, ref["x"])
    op.free()
