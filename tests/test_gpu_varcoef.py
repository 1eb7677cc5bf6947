"""Variable-coefficient stencils: the generator (bit-identical to its NumPy twin) and storage format 5 -- row patterns
for the x positions, the values streamed in tile-sliced ELL order (csrc/mk_spmv_fmt5.h, mk_format.hip).  As for every
format the product must be BIT-identical to the scalar left-to-right CSR loop of the oracle."""
import ctypes

import numpy as np
import pytest

from oracle import csr_ref
from test_gpu_formats import MATS, banded, fmt_info, op_with_format

pytestmark = pytest.mark.gpu


def xs(n, rng):
    return (np.ones(n), rng.standard_normal(n), 1e200 * rng.standard_normal(n))


@pytest.mark.parametrize("shape", [(8, 8, 8), (17, 5, 3), (33, 9, 1), (1, 1, 40), (2, 3, 1), (64, 16, 4)])
def test_generator_matches_its_numpy_twin(shape):
    from pykrylov_amd import gallery
    A = csr_ref.poisson3d_varcoef(*shape, seed=7)
    P = csr_ref.poisson3d(*shape)
    assert np.array_equal(A.indptr, P.indptr) and np.array_equal(A.indices, P.indices)     # the sparsity of poisson3d
    op = gallery.poisson3d_varcoef(*shape, seed=7)
    ip, ix, dv = op.to_csr_arrays()
    assert np.array_equal(ip, A.indptr) and np.array_equal(ix, A.indices)
    assert np.array_equal(dv.view(np.int64), A.data.view(np.int64))                       # values to the last bit
    op.free()
    other = gallery.poisson3d_varcoef(*shape, seed=8)
    assert not np.array_equal(other.to_csr_arrays()[2], A.data)
    other.free()


def test_generator_row_ranges():
    """Rows [a, b) of the global matrix (what one rank of a partitioned run generates)."""
    from pykrylov_amd import CsrOperator, _lib
    A = csr_ref.poisson3d_varcoef(12, 7, 5, seed=3)
    lib = _lib.init()
    for a, b in ((0, 420), (100, 333), (419, 420), (5, 5)):
        h = ctypes.c_void_p()
        _lib.check(lib.mk_csr_poisson3d_varcoef(12, 7, 5, 3, a, b, ctypes.byref(h)))
        op = CsrOperator.from_handle(h.value)
        ip, ix, dv = op.to_csr_arrays()
        lo, hi = A.indptr[a], A.indptr[b]
        assert np.array_equal(ip, A.indptr[a:b + 1] - lo) and np.array_equal(ix, A.indices[lo:hi])
        assert np.array_equal(dv, A.data[lo:hi])
        op.free()


def test_matrix_is_symmetric_positive_definite():
    A = csr_ref.poisson3d_varcoef(6, 5, 4)
    D = A.to_dense()
    assert np.array_equal(D, D.T) and np.linalg.eigvalsh(D).min() > 0.1
    assert len(np.unique(A.data)) == (A.nnz - A.shape[0]) // 2 + A.shape[0]                # every coefficient distinct


def test_format5_on_variable_coefficient_stencils():
    """The matrices the format is made for.  Grids whose lines divide the 256-row tiles must get it; ragged ones may
    exceed the pattern table and stay in format 1 -- either way the bits are the scalar loop's."""
    from pykrylov_amd import CsrOperator, IdentityOperator
    rng = np.random.default_rng(0)
    for shape, sure in (((32, 32, 32), True), ((256, 16, 1), True), ((64, 64, 3), True), ((128, 4, 4), True),
                        ((70001, 1, 1), True), ((30, 30, 30), False), ((17, 23, 9), False), ((150, 150, 1), False)):
        A = csr_ref.poisson3d_varcoef(*shape)
        op = CsrOperator(A.indptr, A.indices, A.data, A.shape)
        info = fmt_info(op)
        assert info["fmt"] in ((5,) if sure else (1, 5)), (shape, info)
        if info["fmt"] == 5:
            # 8 B per nonzero (padded to the tile widths) + 1 B per row + descriptors: less than the 10 B per nonzero
            # of format 1, let alone CSR's 12 B per nonzero + 4 B per row
            # (the builder accepts up to 12.5 % of padding; small matrices also pad their last, partial tile)
            assert info["bytes"] < 9 * A.nnz + 16384 + A.shape[0] + 100 * ((A.shape[0] + 255) // 256), (shape, info)
        for x in xs(A.shape[1], rng):
            assert np.array_equal(op * x, A.matvec(x)), (shape, info)
        u = rng.standard_normal(A.shape[0])
        assert np.array_equal(op.T * u, A.rmatvec(u))
        sh = 2.0 * op - 1.5 * IdentityOperator(A.shape[0])                                 # row program on the sums
        assert np.array_equal(sh * u, 2.0 * A.matvec(u) - 1.5 * u)
        op.free()


@pytest.mark.parametrize("name", sorted(MATS))
def test_format5_request_is_bit_exact_on_every_test_matrix(name):
    """Asking for format 5 on matrices of every kind (dictionary matrices take 4 / 2, scattered ones 0 / 3, mixed tiles
    hand over between the pattern path and the gather path inside one launch)."""
    A, best = MATS[name]
    op = op_with_format(A, 5)
    info = fmt_info(op)
    assert info["fmt"] in {2: (2, 4), 1: (1, 5), 0: (0,)}[best], (name, info)
    rng = np.random.default_rng(3)
    for x in xs(A.shape[1], rng):
        assert np.array_equal(op * x, A.matvec(x)), (name, info)
    u = rng.standard_normal(A.shape[0])
    assert np.array_equal(op.T * u, A.rmatvec(u))
    op.free()


def test_format5_widths_even_and_odd():
    """Tile widths 3 .. 7: pairs only (even), pairs + a last single column (odd), and tiles whose rows are shorter
    than the tile's width (boundary rows: padded with +0.0 that points at the zero cell).  Rows of more than 8 entries
    cannot reach the format today (a windowed tile holds at most 2048 nonzeros, so its width is at most 8), and a matrix
    whose padding would exceed 12.5 % stays in format 1."""
    from pykrylov_amd import CsrOperator
    rng = np.random.default_rng(21)
    n = 20000
    cases = {}
    for offs in ((-1, 0, 1), (-300, -1, 1, 300), (-300, -1, 0, 1, 300), (-600, -300, -1, 1, 300, 600),
                 (-512, -256, -1, 0, 1, 256, 512)):
        r, c, v = banded(n, offs, rng)
        cases[len(offs)] = csr_ref.from_coo(r, c, v, (n, n))
    r, c, v = banded(n, (-1, 0, 1), rng)
    long_rows = np.arange(32, n - 32, 16)
    offs = np.array([-9, -7, -5, 5, 7], dtype=np.int64)
    r2 = np.repeat(long_rows, len(offs))
    c2 = (long_rows[:, None] + offs[None, :]).ravel()
    cases[8] = csr_ref.from_coo(np.concatenate([r, r2]), np.concatenate([c, c2]),
                                np.concatenate([v, rng.standard_normal(len(r2))]), (n, n))
    for width, A in sorted(cases.items()):
        assert int(np.max(np.diff(A.indptr))) == width
        op = CsrOperator(A.indptr, A.indices, A.data, A.shape)
        info = fmt_info(op)
        # (width 8: every 16th row is long, the others are padded from 3 to 8 entries -- far beyond 12.5 %: format 1)
        assert info["fmt"] == (1 if width == 8 else 5), (width, info)
        for x in xs(n, rng):
            assert np.array_equal(op * x, A.matvec(x)), width
        assert np.array_equal(op.T * x, A.rmatvec(x)), width
        op.free()


@pytest.mark.parametrize("fmt", [0, 1, 5])
def test_cg_bits_do_not_depend_on_the_format(fmt):
    """CG on a variable-coefficient problem: identical history and iterate in formats 0, 1 and 5, and bit-identical to
    the oracle run in the device's summation order."""
    from pykrylov_amd import CG
    from oracle import gpu_order, krylov_ref as kr
    A = csr_ref.poisson3d_varcoef(32, 32, 8)
    n = A.shape[0]
    rhs = A.matvec(np.ones(n))
    op = op_with_format(A, fmt)
    assert fmt_info(op)["fmt"] == fmt
    s = CG(op)
    s.solve(rhs)
    ref = kr.cg(A, rhs, red=kr.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES["cg"], gpu_order.launch_geometry(op))))
    assert s.converged and s.nMatvec == ref["nMatvec"]
    assert np.array_equal(np.array(s.residHistory), ref["residHistory"]) and np.array_equal(s.x, ref["x"])
    # ... and within 1e-12 of the reference's own summation order (np.dot)
    ref0 = kr.cg(A, rhs)
    assert s.nMatvec == ref0["nMatvec"]
    h, h0 = np.array(s.residHistory), ref0["residHistory"]
    assert np.max(np.abs(h - h0) / np.maximum(h0, 1e-4 * h0[0])) <= 1e-12
    assert np.linalg.norm(s.x - ref0["x"]) <= 1e-12 * np.linalg.norm(ref0["x"])
    op.free()


def test_minres_row_x_hook_in_format5(monkeypatch):
    """MINRES takes s * y[i] from the product's LDS window (row_x): rows with and without a diagonal entry."""
    from pykrylov_amd import CsrOperator, Minres
    from oracle import gpu_order, krylov_ref as kr
    monkeypatch.setattr(kr, "_sq", lambda a: a * a)
    n = 6001
    rng = np.random.default_rng(4)
    i = np.arange(n - 1)
    dg = np.arange(n)[np.arange(n) % 7 != 0]             # (most rows carry a diagonal entry: little padding)
    w = rng.standard_normal(n - 1)
    A = csr_ref.from_coo(np.concatenate([i, i + 1, dg]), np.concatenate([i + 1, i, dg]),
                         np.concatenate([w, w, 3.0 + rng.random(len(dg))]), (n, n))
    op = CsrOperator(A.indptr, A.indices, A.data, A.shape, symmetric=True)
    assert fmt_info(op)["fmt"] == 5
    rhs = A.matvec(np.linspace(1.0, 2.0, n))
    s = Minres(op)
    s.solve(rhs, show=False, check=False, etol=0.0, rtol=1e-12, itnlim=60)
    ref = kr.minres(A, rhs, check=False, etol=0.0, rtol=1e-12, itnlim=60,
                    red=kr.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES["minres"], gpu_order.launch_geometry(op))))
    assert (s.istop, s.itn) == (ref["istop"], ref["itn"]) and s.itn == 60
    assert np.array_equal(np.array(s.residHistory), ref["residHistory"]) and np.array_equal(s.x, ref["x"])
    op.free()


@pytest.mark.parametrize("solver", ["bicgstab", "cgs", "tfqmr"])
def test_nonsymmetric_solvers_in_format5(solver):
    """Gated products and fused epilogues on a banded nonsymmetric matrix with all-distinct values."""
    import pykrylov_amd
    from pykrylov_amd import CsrOperator
    from oracle import gpu_order, krylov_ref as kr
    rng = np.random.default_rng(6)
    n = 12345
    r, c, v = banded(n, (-200, -1, 0, 1, 200), rng)
    v = np.where(r == c, 6.0 + rng.random(len(v)), 0.9 * rng.random(len(v)))               # diagonally dominant
    A = csr_ref.from_coo(r, c, v, (n, n))
    op = CsrOperator(A.indptr, A.indices, A.data, A.shape)
    assert fmt_info(op)["fmt"] == 5
    rhs = A.matvec(np.ones(n))
    cls = {"bicgstab": pykrylov_amd.BiCGSTAB, "cgs": pykrylov_amd.CGS, "tfqmr": pykrylov_amd.TFQMR}[solver]
    s = cls(op, reltol=1e-10)
    s.solve(rhs)
    ref = getattr(kr, solver)(A, rhs, reltol=1e-10, red=kr.Reductions(
        gpu_order.GpuDots(n, gpu_order.SPMV_SITES[solver], gpu_order.launch_geometry(op))))
    assert s.converged and s.nMatvec == ref["nMatvec"] and np.array_equal(s.x, ref["x"])
    op.free()


def test_fuzz_over_grid_shapes():
    from pykrylov_amd import CsrOperator
    rng = np.random.default_rng(13)
    shapes = [(int(a), int(b), int(c)) for a, b, c in zip(rng.integers(3, 70, 12), rng.integers(1, 40, 12), rng.integers(1, 30, 12))]
    shapes += [(256, 3, 2), (255, 2, 2), (257, 2, 1), (512, 2, 2), (1, 1, 300), (2, 129, 5)]
    seen = set()
    for mx, my, mz in shapes:
        A = csr_ref.poisson3d_varcoef(mx, my, mz, seed=mx + my)
        op = CsrOperator(A.indptr, A.indices, A.data, A.shape)
        seen.add(fmt_info(op)["fmt"])
        x = rng.standard_normal(A.shape[1])
        assert np.array_equal(op * x, A.matvec(x)), (mx, my, mz, fmt_info(op))
        assert np.array_equal(op.T * x, A.rmatvec(x)), (mx, my, mz)
        op.free()
    assert 5 in seen


def test_fuzz_random_banded_matrices_with_distinct_values():
    """Random band structures (3 .. 8 offsets, some far, some near, some missing on random row ranges), random sizes with
    ragged last tiles, rectangular shapes, all-distinct values: whichever format the builder settles on (5 where the
    rows follow few patterns with little padding, 1 / 0 / 3 otherwise), products and transposed products keep the
    scalar loop's bits."""
    from pykrylov_amd import CsrOperator
    rng = np.random.default_rng(2027)
    seen = {}
    for case in range(36):
        n = int(rng.integers(300, 9000))
        ncols = n if case % 5 else int(n + rng.integers(-200, 200))
        k = int(rng.integers(3, 9))
        near = rng.choice(np.arange(-12, 13), size=min(k, 5), replace=False)
        far = rng.choice([-1024, -513, -300, -257, 257, 300, 511, 1024, 2000], size=max(0, k - len(near)), replace=False)
        offs = np.unique(np.concatenate([near, far]))
        rows, cols = [], []
        for o in offs:
            r = np.arange(max(0, -o), min(n, ncols - o))
            if case % 3 == 0 and len(r) > 600:                       # a hole: the offset is missing on a range of rows
                lo = int(rng.integers(0, len(r) - 300))
                r = np.concatenate([r[:lo], r[lo + 300:]])
            rows.append(r)
            cols.append(r + o)
        rows, cols = np.concatenate(rows), np.concatenate(cols)
        A = csr_ref.from_coo(rows, cols, rng.standard_normal(len(rows)), (n, ncols))
        op = CsrOperator(A.indptr, A.indices, A.data, A.shape)
        f = fmt_info(op)["fmt"]
        seen[f] = seen.get(f, 0) + 1
        x = rng.standard_normal(ncols)
        assert np.array_equal(op * x, A.matvec(x)), (case, n, ncols, offs, f)
        u = rng.standard_normal(n)
        assert np.array_equal(op.T * u, A.rmatvec(u)), (case, n, ncols, offs, f)
        op.free()
    assert seen.get(5, 0) >= 5, seen
