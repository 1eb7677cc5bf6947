"""The wide storage formats (csrc/mk_spmv_fmtw.h, mk_format.hip): rows of up to 32 entries, tiles of up to 8192
nonzeros, 32 window chunks -- 6 = slots + values streamed, 7 = row patterns + values streamed, 8 = row patterns + value
dictionary.  As for every format: the product is BIT-identical to the oracle's scalar left-to-right CSR loop, and the
solver loops on top of it are the oracle's loops in the device's summation order, bit for bit."""
import numpy as np
import pytest

from oracle import csr_ref
from test_gpu_formats import MATS, banded, fmt_info, op_with_format

pytestmark = pytest.mark.gpu


def xs(n, rng):
    return (np.ones(n), rng.standard_normal(n), 1e200 * rng.standard_normal(n))


def check_products(op, A, seed=3):
    rng = np.random.default_rng(seed)
    for x in xs(A.shape[1], rng):
        with np.errstate(over="ignore", invalid="ignore"):
            want = A.matvec(x)
        got = op * x
        assert np.array_equal(got.view(np.int64), want.view(np.int64))


def fixed_width_random_band(n, width, spread, rng, distinct=True):
    """Every row holds `width` entries at random columns within +-spread of the diagonal (different for every row: no
    row patterns), the diagonal among them."""
    rows, cols = [], []
    for r in range(n):
        lo, hi = max(0, r - spread), min(n, r + spread + 1)
        c = rng.choice(np.arange(lo, hi), size=min(width, hi - lo) - 1, replace=False)
        c = np.unique(np.concatenate([c[c != r], [r]]))
        rows.append(np.full(len(c), r))
        cols.append(c)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    vals = rng.standard_normal(len(rows)) if distinct else rng.choice([-1.0, 0.5, 2.0], size=len(rows))
    vals = np.where(rows == cols, width + rng.random(len(rows)), vals)
    return csr_ref.from_coo(rows, cols, vals, (n, n))


@pytest.mark.parametrize("seed,expect", [(0, 8), (7, 7)])
@pytest.mark.parametrize("shape", [(256, 6, 5), (64, 16, 6), (128, 12, 7), (512, 4, 3)])
def test_27_point_stencils_get_the_pattern_formats(shape, seed, expect):
    """Grids whose lines divide (or are divided by) the 256-row tiles: few row patterns.  Constant coefficients ->
    dictionary + patterns (one byte per row), variable coefficients -> patterns + streamed values."""
    from pykrylov_amd import gallery
    A = csr_ref.stencil27(*shape, seed=seed)
    op = gallery.stencil27(*shape, seed=seed)
    info = fmt_info(op)
    ntiles = (A.shape[0] + 255) // 256
    assert info["fmt"] == expect, info
    assert info["tiles"] == ntiles and 0 < info["chunks"] <= 32
    if expect == 8:
        assert info["ndict"] == 2 and info["bytes"] < 2 * A.shape[0] + 400 * ntiles            # ~1 B per row + descriptors
    else:
        assert info["bytes"] < 8.5 * A.nnz + A.shape[0] + 400 * ntiles                          # 8 B per nonzero + 1 B per row
    check_products(op, A)
    op.free()


@pytest.mark.parametrize("seed", [0, 7])
@pytest.mark.parametrize("want", [6, 7, 8])
@pytest.mark.parametrize("shape", [(256, 6, 5), (40, 40, 12), (300, 7, 3), (33, 9, 4), (700, 3, 3)])
def test_requests_degrade_and_every_mode_gives_the_scalar_loops_bits(shape, seed, want):
    from pykrylov_amd import _lib, gallery
    A = csr_ref.stencil27(*shape, seed=seed)
    op = gallery.stencil27(*shape, seed=seed)
    _lib.check(op._lib.mk_csr_set_format(op.handle, want))
    info = fmt_info(op)
    assert info["fmt"] in (0, 3, 6, 7, 8) and info["fmt"] <= want, info
    if want == 6 and shape[0] >= 40:
        assert info["fmt"] == 6                              # (slots + values need nothing but a cover and even rows)
    if info["fmt"] == 8:
        assert seed == 0
    check_products(op, A, seed=shape[0])
    op.free()


@pytest.mark.parametrize("name", sorted(MATS))
@pytest.mark.parametrize("want", [6, 7])
def test_wide_kernel_on_the_narrow_matrices(name, want):
    """Asked for explicitly, formats 6 and 7 also serve matrices the 16-chunk cover handles (windows of 4 chunks per
    wave in the same kernel): tiles without windows, empty rows, rectangular shapes, dense rows in between."""
    A, best = MATS[name]
    op = op_with_format(A, want)
    info = fmt_info(op)
    assert info["fmt"] in (0, 3, 6, 7)
    if want == 7 and best >= 1 and name not in ("banded_plus_dense_rows", "empty_rows_and_tiles"):
        assert info["fmt"] >= 6, (name, info)                # (format 6 refuses rows of 5: their slots would pad to 8)
    check_products(op, A)
    op.free()


def test_rows_without_patterns_stream_their_slots():
    rng = np.random.default_rng(11)
    A = fixed_width_random_band(5000, 20, 1200, rng)
    op = op_with_format(A, -1)
    info = fmt_info(op)
    assert info["fmt"] == 6, info
    assert info["bytes"] < 10.6 * A.nnz + 500 * ((A.shape[0] + 255) // 256)
    check_products(op, A)
    T = op.T
    assert np.array_equal(T * np.ones(A.shape[0]), A.rmatvec(np.ones(A.shape[0])))
    op.free()


def test_few_values_without_patterns_fall_back_to_streamed_values():
    """A dictionary alone does not make format 8: without row patterns the values are streamed (format 6)."""
    rng = np.random.default_rng(12)
    A = fixed_width_random_band(4000, 18, 1000, rng, distinct=False)
    op = op_with_format(A, -1)
    assert fmt_info(op)["fmt"] == 6
    check_products(op, A)
    op.free()


def test_tiles_beyond_the_wide_limits_gather_between_wide_tiles():
    """Dense rows (tiles of more than 8192 nonzeros) and a block of scattered columns inside a 27-point matrix."""
    from pykrylov_amd import CsrOperator
    rng = np.random.default_rng(13)
    S = csr_ref.stencil27(64, 12, 6, seed=5)
    n = S.shape[0]
    rows = np.repeat(np.arange(n), np.diff(S.indptr))
    dr = np.repeat([300, 301, 2000], 3000)
    dc = np.concatenate([rng.choice(n, 3000, replace=False) for _ in range(3)])
    sr = np.repeat(np.arange(1024, 1280), 9)
    sc = rng.integers(0, n, size=len(sr))
    A = csr_ref.from_coo(np.concatenate([rows, dr, sr]), np.concatenate([S.indices, dc, sc]),
                         np.concatenate([S.data, rng.standard_normal(len(dr) + len(sr))]), (n, n))
    op = CsrOperator(A.indptr, A.indices, A.data, A.shape)
    info = fmt_info(op)
    assert info["fmt"] in (6, 7) and 0 < info["tiles"] < (n + 255) // 256, info  # (those tiles gather; the others keep their patterns)
    check_products(op, A)
    op.free()


def test_ragged_rows_pad_up_to_half():
    """The slot format pads a tile's rows to its longest one; up to 50 % of padding it still beats the gather path."""
    rng = np.random.default_rng(15)
    n = 6000
    A = fixed_width_random_band(n, 24, 900, rng)
    rows = np.repeat(np.arange(n), np.diff(A.indptr))
    drop = (rows % 2 == 1) & (A.indices != rows) & (rng.random(A.nnz) < 0.5)       # odd rows keep ~12 of 24: 1.33 x
    B = csr_ref.from_coo(rows[~drop], A.indices[~drop], A.data[~drop], (n, n))
    op = op_with_format(B, -1)
    info = fmt_info(op)
    assert info["fmt"] == 6, info
    assert info["bytes"] > 10 * B.nnz                        # (the padding is priced)
    check_products(op, B)
    op.free()


def test_ragged_rows_are_refused():
    """Rows of very different lengths would pad the ELL blocks by more than 50 %: the matrix stays on the CSR path."""
    rng = np.random.default_rng(14)
    n = 6000
    parts = [fixed_width_random_band(n, 24, 900, rng)]
    A = parts[0]
    keep = np.ones(A.nnz, dtype=bool)
    rows = np.repeat(np.arange(n), np.diff(A.indptr))
    drop = (rows % 2 == 1) & (A.indices != rows) & (rng.random(A.nnz) < 0.9)       # odd rows keep ~3 of 24: 1.8 x
    keep &= ~drop
    B = csr_ref.from_coo(rows[keep], A.indices[keep], A.data[keep], (n, n))
    op = op_with_format(B, -1)
    assert fmt_info(op)["fmt"] in (0, 3)
    check_products(op, B)
    op.free()


@pytest.mark.parametrize("seed", [0, 7])
@pytest.mark.parametrize("shape", [(128, 10, 5), (20, 20, 16), (7, 40, 23)])
def test_minres_row_x_hook_in_the_wide_formats(shape, seed, monkeypatch):
    """(short lines: a wave's rows follow three and more patterns -- the rounds of the scalar path and its LDS fallback,
    each with the diagonal cell handed to the epilogue)"""
    from pykrylov_amd import Minres, gallery
    from oracle import gpu_order, krylov_ref as kr
    monkeypatch.setattr(kr, "_sq", lambda a: a * a)
    A = csr_ref.stencil27(*shape, seed=seed)
    n = A.shape[0]
    op = gallery.stencil27(*shape, seed=seed)
    assert fmt_info(op)["fmt"] in ((7 if seed else 8), 6)    # (7 x 40 x 23: too many patterns for the table -> slots)
    rhs = A.matvec(np.linspace(1.0, 2.0, n))
    s = Minres(op)
    s.solve(rhs, show=False, check=False, etol=0.0, rtol=1e-12, itnlim=40)
    ref = kr.minres(A, rhs, check=False, etol=0.0, rtol=1e-12, itnlim=40,
                    red=kr.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES["minres"], gpu_order.launch_geometry(op))))
    assert (s.istop, s.itn) == (ref["istop"], ref["itn"])
    assert np.array_equal(np.array(s.residHistory), ref["residHistory"]) and np.array_equal(s.x, ref["x"])
    op.free()


@pytest.mark.parametrize("solver", ["bicgstab", "cgs", "tfqmr", "minres", "symmlq"])
def test_solvers_on_slot_streams(solver, monkeypatch):
    """Gated products, fused epilogues and the row_x fallback (format 6 has no diagonal cell) on a matrix without
    row patterns."""
    import pykrylov_amd
    from pykrylov_amd import CsrOperator
    from oracle import gpu_order, krylov_ref as kr
    monkeypatch.setattr(kr, "_sq", lambda a: a * a)
    rng = np.random.default_rng(16)
    n = 7000
    A = fixed_width_random_band(n, 16, 900, rng)
    sym = solver in ("minres", "symmlq")
    if sym:
        T = A.transpose()
        rows = np.repeat(np.arange(n), np.diff(A.indptr))
        trows = np.repeat(np.arange(n), np.diff(T.indptr))
        A = csr_ref.from_coo(np.concatenate([rows, trows]), np.concatenate([A.indices, T.indices]),
                             np.concatenate([0.5 * A.data, 0.5 * T.data]), (n, n))
    op = CsrOperator(A.indptr, A.indices, A.data, A.shape, symmetric=sym)
    info = fmt_info(op)
    assert info["fmt"] in (6, 0, 3), info
    rhs = A.matvec(np.ones(n))
    geo = gpu_order.launch_geometry(op)
    if solver == "minres":
        s = pykrylov_amd.Minres(op)
        kw = dict(check=False, rtol=1e-12, itnlim=30, etol=0.0)
        s.solve(rhs, show=False, **kw)
        ref = kr.minres(A, rhs, red=kr.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES[solver], geo)), **kw)
        assert s.itn == ref["itn"] and np.array_equal(s.x, ref["x"])
    elif solver == "symmlq":
        s = pykrylov_amd.Symmlq(op)
        s.solve(rhs, matvec_max=40)
        ref = kr.symmlq(A, rhs, matvec_max=40, red=kr.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES[solver], geo)))
        assert s.nMatvec == ref["nMatvec"] and np.array_equal(s.x, ref["x"])
    else:
        cls = {"bicgstab": pykrylov_amd.BiCGSTAB, "cgs": pykrylov_amd.CGS, "tfqmr": pykrylov_amd.TFQMR}[solver]
        s = cls(op, reltol=1e-10)
        s.solve(rhs)
        ref = getattr(kr, solver)(A, rhs, reltol=1e-10, red=kr.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES[solver], geo)))
        assert s.converged and s.nMatvec == ref["nMatvec"] and np.array_equal(s.x, ref["x"])
    op.free()


@pytest.mark.parametrize("seed", [0, 7])
def test_row_programs_on_wide_formats(seed):
    """alpha * A + D and A - I as row programs of the wide kernel (PROG instantiation), against the reference's
    expression order."""
    from pykrylov_amd import DiagonalOperator, IdentityOperator, gallery
    shape = (64, 16, 5)
    A = csr_ref.stencil27(*shape, seed=seed)
    n = A.shape[0]
    op = gallery.stencil27(*shape, seed=seed)
    rng = np.random.default_rng(5)
    dv = rng.standard_normal(n)
    x = rng.standard_normal(n)
    c1 = 2.5 * op + DiagonalOperator(dv)
    assert np.array_equal(c1 * x, 2.5 * A.matvec(x) + dv * x)
    c2 = op - IdentityOperator(n)
    assert np.array_equal(c2 * x, A.matvec(x) - x)
    op.free()


def test_lsqr_on_a_wide_rectangular_operator():
    """Transpose products (the transposed copy gets its own format) through LSQR."""
    from pykrylov_amd import CsrOperator
    from pykrylov_amd.lls import LSQRFramework
    from oracle import gpu_order, lls_ref
    rng = np.random.default_rng(18)
    S = fixed_width_random_band(4000, 14, 700, rng)
    keep = S.indices < 3600
    rows = np.repeat(np.arange(4000), np.diff(S.indptr))
    A = csr_ref.from_coo(rows[keep], S.indices[keep], S.data[keep], (4000, 3600))
    op = CsrOperator(A.indptr, A.indices, A.data, A.shape)
    b = A.matvec(np.ones(3600)) + 1e-3 * rng.standard_normal(4000)
    s = LSQRFramework(op)
    s.solve(b, itnlim=25, etol=0.0, show=False)
    x, istop, itn = s.x, s.istop, s.itn
    ref = lls_ref.lsqr(A.matvec, A.transpose().matvec, A.shape, b.copy(), itnlim=25, etol=0.0)
    assert itn == ref["itn"] and istop == ref["istop"]
    assert np.linalg.norm(x - ref["x"]) <= 1e-11 * np.linalg.norm(ref["x"])
    op.free()


def test_fuzz_over_grid_shapes():
    """27-point operators on grids of every aspect ratio (lines shorter and longer than a tile, single planes, single
    lines), constant and variable coefficients: whatever format the builder settles on, the scalar loop's bits."""
    from pykrylov_amd import gallery
    rng = np.random.default_rng(21)
    shapes = [(int(a), int(b), int(c)) for a, b, c in zip(rng.integers(2, 90, 10), rng.integers(1, 30, 10), rng.integers(1, 12, 10))]
    shapes += [(256, 3, 2), (255, 2, 2), (257, 2, 1), (512, 2, 2), (1, 1, 300), (2, 129, 3), (1024, 1, 1), (100, 100, 1)]
    seen = set()
    for k, (mx, my, mz) in enumerate(shapes):
        seed = 0 if k % 2 else mx + my
        A = csr_ref.stencil27(mx, my, mz, seed=seed)
        op = gallery.stencil27(mx, my, mz, seed=seed)
        seen.add(fmt_info(op)["fmt"])
        x = rng.standard_normal(A.shape[1])
        assert np.array_equal((op * x).view(np.int64), A.matvec(x).view(np.int64)), (mx, my, mz, seed, fmt_info(op))
        T = op.T
        assert np.array_equal((T * x).view(np.int64), A.rmatvec(x).view(np.int64)), (mx, my, mz, seed)
        op.free()
    assert seen & {7, 8}, seen


@pytest.mark.parametrize("seed", [0, 7])
@pytest.mark.parametrize("order", [(0, 0, 0), (1, 0, 0), (2, 0, 0), (3, 4, 0), (4, 2, 16)])
def test_wide_formats_in_every_tile_order(order, seed):
    """64 x 64 x 12 grid of the 27-point operator: 192 tiles, 16 per plane.  Products and a whole CG solve, bit for bit
    the oracle's in the device's summation order, whichever way the tiles are dealt to the workgroups."""
    from pykrylov_amd import CG, _lib, gallery
    from oracle import gpu_order, krylov_ref as kr
    A = csr_ref.stencil27(64, 64, 12, seed=seed)
    n = A.shape[0]
    op = gallery.stencil27(64, 64, 12, seed=seed)
    _lib.check(_lib.init().mk_csr_set_tile_order(op.handle, order[0], order[1], order[2], -1))
    assert fmt_info(op)["fmt"] == (7 if seed else 8)
    x = np.random.default_rng(2).standard_normal(n)
    assert np.array_equal(op * x, A.matvec(x))
    rhs = A.matvec(np.ones(n))
    s = CG(op)
    s.solve(rhs, matvec_max=40)
    geo = gpu_order.launch_geometry(op)
    ref = kr.cg(A, rhs, matvec_max=40, red=kr.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES["cg"], geo)))
    assert s.nMatvec == ref["nMatvec"]
    assert np.array_equal(np.array(s.residHistory), ref["residHistory"]) and np.array_equal(s.x, ref["x"])
    op.free()
