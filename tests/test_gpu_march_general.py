"""The brick march on ANY grid side (round 6; csrc/mk_spmv_fmt9.h, template flag GEN): lines that are no multiple of 128 rows,
planes that are no multiple of four lines, odd strides (pairs of rows at any 8-byte boundary), 5-point matrices marched line by
line (one far stride), a chunk's leftover planes as one masked round.  Reference sizes this covers: the grid sides of
pykrylov/cg/tests/test_diagdom.py:53 (10, 20, 100, 500 -- `int(sqrt(len(x)))`, pykrylov/gallery/gallery.py:12), none of which
is a multiple of 128.  Everything is compared BIT for bit: products with the oracle's scalar left-to-right CSR loop, CG runs
with the oracle run in the march's summation order (oracle/gpu_order.py `pencil_partials`) and with the three-kernel pass."""
import ctypes

import numpy as np
import pytest

from oracle import csr_ref, gpu_order, krylov_ref
from test_gpu_pencil import _solver_is_fused, banded7, fmt_of, op9, sym_banded

pytestmark = pytest.mark.gpu


def march_info(op):
    from pykrylov_amd import _lib
    info = (ctypes.c_int64 * 12)()
    _lib.check(_lib.init().mk_csr_march_info(op.handle, info, 12))
    keys = ("fmt", "L", "P", "planes", "lines", "bx", "by", "zc", "chunks", "gen", "per", "patterns")
    return dict(zip(keys, list(info)))


def expected_geometry(L, P, two_d=False):
    """(gen, bricks per line, brick rows) the builder must choose (csrc/mk_format.hip pencil_geometry): whole aligned bricks -> 0,
    anything else the general geometry (2)."""
    bx, by = -(-L // 128), -(-(-(-P // L)) // 4)
    return (0 if L % 128 == 0 and P % (4 * L) == 0 else 2), bx, by


# (nx, ny, nz): line lengths below / above 128 and 256, odd lengths, planes of 5 .. 15 lines, plane counts that leave 1 .. 5
# planes over after whole rounds of six; bricks 50 .. 98 % full (partly empty last brick of a line, partly empty last group of
# lines, odd L, odd P)
GRIDS = [(100, 8, 8), (100, 9, 7), (200, 8, 5), (250, 7, 13), (101, 8, 5), (101, 9, 11), (129, 12, 4), (500, 5, 3), (384, 6, 9),
         (90, 13, 8), (250, 8, 5), (500, 8, 3), (128, 15, 5), (250, 15, 4), (255, 8, 5), (255, 15, 3),
         (300, 16, 5), (200, 32, 3), (400, 16, 4), (301, 24, 3), (203, 48, 2)]


@pytest.mark.parametrize("dims", GRIDS)
@pytest.mark.parametrize("fmt", [9, 10, 11])
def test_general_grid_product_bit_exact(dims, fmt):
    A = csr_ref.poisson3d(*dims) if fmt == 9 else csr_ref.poisson3d_varcoef(*dims, seed=3)
    op = op9(A, fmt=fmt)
    rng = np.random.default_rng(3)
    x = rng.standard_normal(A.shape[1])
    y = op * x
    info = march_info(op)
    assert info["fmt"] == fmt, info
    gen, bx, by = expected_geometry(dims[0], dims[0] * dims[1])
    assert gen == 2 and (info["L"], info["P"], info["planes"], info["gen"]) == (dims[0], dims[0] * dims[1], dims[2], gen), info
    assert (info["bx"], info["by"]) == (bx, by), info
    assert np.array_equal(y, A.matvec(x))
    x[::7] = 0.0
    x[5::11] *= 1e300
    assert np.array_equal(op * x, A.matvec(x))


@pytest.mark.parametrize("m", [300, 500, 700, 1000, 1001])
@pytest.mark.parametrize("fmt", [9, 11])
def test_five_point_matrix_is_marched_line_by_line(m, fmt):
    """offsets {0, +-1, +-M}: planes of M rows (one grid line each), cut into pieces of 128 rows that have no +-L entries"""
    A = csr_ref.poisson2d(m)
    if fmt == 11:                                            # (all values different: no dictionary)
        rng = np.random.default_rng(m)
        A = sym_banded(m * m, 128, m, rng, drop=0.0)
        keep = np.abs(A.indices - np.repeat(np.arange(m * m), np.diff(A.indptr))) != 128
        A = csr_ref.from_coo(np.repeat(np.arange(m * m), np.diff(A.indptr))[keep], A.indices[keep], A.data[keep], A.shape)
    op = op9(A, fmt=fmt)
    x = np.random.default_rng(1).standard_normal(m * m)
    assert np.array_equal(op * x, A.matvec(x))
    info = march_info(op)
    gen = expected_geometry(128, m, two_d=True)[0]
    assert (info["fmt"], info["L"], info["P"], info["planes"], info["gen"]) == (fmt, 128, m, m, gen), info


@pytest.mark.parametrize("n,L,P,drop", [(700 * 9, 100, 700, 0.0), (750 * 8, 100, 750, 0.3), (909 * 7, 101, 909, 0.5),
                                        (1155 * 6, 165, 1155, 0.2), (2000 * 6, 250, 2000, 0.3), (3825 * 4, 255, 3825, 0.2),
                                        (1530 * 5, 510, 1530, 0.1)])
def test_general_band_matrix_bit_exact(n, L, P, drop):
    """No grid geometry: +-1 entries across line ends (a lane past the end of its line holds the NEXT line's first rows: the
    natural index), any subset of the offsets, P not a multiple of L, odd strides, zeros / negative zero / denormals among the
    values, an infinity in x next to entries a row does not have."""
    rng = np.random.default_rng(n)
    A = banded7(n, L, P, rng, drop=drop)
    for fmt in (9, 10):
        B = A
        if fmt == 10:
            B = csr_ref.RefCsr(A.indptr, A.indices, rng.standard_normal(A.nnz), A.shape)
        op = op9(B, fmt=fmt)
        x = rng.standard_normal(n)
        y = op * x
        assert fmt_of(op) == fmt, march_info(op)
        assert np.array_equal(y, B.matvec(x)) and np.array_equal(np.signbit(y), np.signbit(B.matvec(x)))
        j = int(rng.integers(P, n - P))
        x[j] = np.inf
        with np.errstate(invalid="ignore"):
            ref = B.matvec(x)
        got = op * x
        assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(np.isinf(got), np.isinf(ref))
        fin = np.isfinite(ref)
        assert np.array_equal(got[fin], ref[fin])


@pytest.mark.parametrize("n,L,P,drop", [(700 * 9, 100, 700, 0.3), (909 * 7, 101, 909, 0.0), (2000 * 6, 250, 2000, 0.4), (3825 * 4, 255, 3825, 0.2)])
def test_general_symmetric_band_format11(n, L, P, drop):
    rng = np.random.default_rng(n + 1)
    A = sym_banded(n, L, P, rng, drop=drop)
    op = op9(A, fmt=11)
    x = rng.standard_normal(n)
    assert np.array_equal(op * x, A.matvec(x)) and fmt_of(op) == 11
    assert march_info(op)["gen"] == expected_geometry(L, P)[0]
    j = int(rng.integers(P, n - P))
    x[j] = np.inf
    with np.errstate(invalid="ignore"):
        ref = A.matvec(x)
    got = op * x
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(np.isinf(got), np.isinf(ref))
    fin = np.isfinite(ref)
    assert np.array_equal(got[fin], ref[fin])


def test_sparse_bricks_degrade():
    """Less than half of a brick's lanes with rows (L = 132 in planes of 9 lines: 39 %; L = 37: 29 %) -> the windowed formats
    keep the matrix, as does a matrix with offsets outside the class."""
    from pykrylov_amd import CsrOperator, _lib
    rng = np.random.default_rng(1)
    for A in (csr_ref.poisson3d(132, 9, 8), csr_ref.poisson3d(37, 40, 6), csr_ref.stencil27(100, 8, 4)):
        for want in (9, 10):
            op = CsrOperator(A.indptr, A.indices, A.data, A.shape)
            _lib.check(_lib.init().mk_csr_set_format(op.handle, want))
            x = rng.standard_normal(A.shape[1])
            assert np.array_equal(op * x, A.matvec(x))
            assert fmt_of(op) not in (9, 10, 11) and march_info(op)["L"] == 0


@pytest.mark.parametrize("dims,fmt", [((100, 9, 7), 9), ((250, 7, 13), 9), ((101, 9, 11), 11), ((200, 8, 20), 11), ((250, 8, 13), 9),
                                      ((255, 15, 8), 11), ((500, 8, 7), 11)])
def test_cg_on_a_general_grid_bit_exact_fused_and_unfused(dims, fmt, monkeypatch):
    """CG with <p, Ap> fused into the march of a general geometry: history, iterate and matvec count equal the oracle run in
    the march's summation order bit for bit, and the fused passes (x / p update inside the next product kernel) equal the
    three-kernel passes bit for bit -- default stopping, matvec_max cutting the run short, an initial guess."""
    from pykrylov_amd import CG
    A = csr_ref.poisson3d(*dims) if fmt == 9 else csr_ref.poisson3d_varcoef(*dims, seed=5)
    n = A.shape[0]
    rng = np.random.default_rng(4)
    rhs = A.matvec(np.ones(n)) + 0.1 * rng.standard_normal(n)
    guess = rng.standard_normal(n)
    cases = [dict(matvec_max=150), dict(matvec_max=7), dict(guess=guess, matvec_max=23)]
    out = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("MK_CG_FUSE", fuse)
        op = op9(A, symmetric=True, fmt=fmt)
        assert _solver_is_fused(op, rhs) == (fuse == "1")
        res = []
        for kw in cases:
            s = CG(op)
            s.solve(rhs, **kw)
            res.append((s.nMatvec, np.array(s.residHistory), s.x.copy(), float(s.residNorm)))
        assert fmt_of(op) == fmt
        out[fuse] = res
        geo = gpu_order.launch_geometry(op)
    assert geo[1][0] == "pencil" and geo[1][6] == expected_geometry(dims[0], dims[0] * dims[1])[0]
    for a, b in zip(out["1"], out["0"]):
        assert a[0] == b[0] and a[3] == b[3] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    ref = krylov_ref.cg(A, rhs, matvec_max=150, red=krylov_ref.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES["cg"], geometry=geo)))
    a = out["1"][0]
    assert a[0] == ref["nMatvec"] and np.array_equal(a[1], ref["residHistory"]) and np.array_equal(a[2], ref["x"])


def test_other_loops_on_a_general_geometry_keep_the_bits():
    """Only plain products and CG have general-geometry kernels; a march format forced by hand on another loop's matrix runs
    that loop's products as the CSR gather kernel on the same arrays: same row sums, bit for bit."""
    from pykrylov_amd import Minres, BiCGSTAB
    A = csr_ref.poisson3d(100, 9, 7)
    n = A.shape[0]
    rhs = A.matvec(np.ones(n))
    op = op9(A, symmetric=True)
    m = Minres(op)
    m.solve(rhs, show=False, check=False, shift=1.5, etol=0.0, rtol=1e-10, itnlim=60)
    geo = gpu_order.launch_geometry(op)
    b = BiCGSTAB(op)
    b.solve(rhs, matvec_max=40)
    op0 = op9(A, symmetric=True, fmt=0)
    m0 = Minres(op0)
    m0.solve(rhs, show=False, check=False, shift=1.5, etol=0.0, rtol=1e-10, itnlim=60)
    assert m.itn == m0.itn and np.allclose(np.array(m.residHistory), np.array(m0.residHistory), rtol=1e-12, atol=0)
    assert np.linalg.norm(m.x - m0.x) <= 1e-12 * np.linalg.norm(m0.x)
    x = np.random.default_rng(0).standard_normal(n)
    assert np.array_equal(op * x, A.matvec(x)) and fmt_of(op) == 9 and geo[1][0] == "pencil"
    assert b.nMatvec <= 40 and np.isfinite(b.residNorm)


def test_composed_operator_on_a_general_geometry():
    from pykrylov_amd.linop import DiagonalOperator
    A = csr_ref.poisson3d(100, 9, 7)
    n = A.shape[0]
    op = op9(A)
    rng = np.random.default_rng(2)
    d = rng.standard_normal(n)
    x = rng.standard_normal(n)
    comp = 2.5 * op + DiagonalOperator(d)
    assert np.array_equal(comp * x, 2.5 * A.matvec(x) + d * x) and fmt_of(op) == 9


def test_general_kernels_on_an_aligned_geometry_change_no_bit(monkeypatch):
    """MK_PEN_GEN=1 runs the general-geometry kernels on a matrix whose bricks are whole and aligned (A/B runs): product and
    CG run equal the aligned kernels' bit for bit (child processes: the switch is read once)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, ctypes
import numpy as np
sys.path.insert(0, %r)
from oracle import csr_ref
from pykrylov_amd import CG, CsrOperator, _lib
out = []
for fmt, gen in ((9, csr_ref.poisson3d), (11, csr_ref.poisson3d_varcoef)):
    A = gen(256, 8, 26)
    op = CsrOperator(A.indptr, A.indices, A.data, A.shape, symmetric=True)
    _lib.check(_lib.init().mk_csr_set_format(op.handle, fmt))
    x = np.random.default_rng(0).standard_normal(A.shape[0])
    y = op * x
    assert np.array_equal(y, A.matvec(x))
    info = (ctypes.c_int64 * 12)()
    _lib.check(_lib.init().mk_csr_march_info(op.handle, info, 12))
    s = CG(op); s.solve(A.matvec(np.ones(A.shape[0])), matvec_max=60)
    out.append((int(info[9]), s.nMatvec, np.array(s.residHistory).tobytes().hex(), s.x.tobytes().hex()[:4096]))
print(repr(out))
''' % root
    res = {}
    for gen in ("0", "1"):
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, MK_PEN_GEN=gen))
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
        res[gen] = eval(p.stdout.strip().splitlines()[-1])
    assert [r[0] for r in res["0"]] == [0, 0] and [r[0] for r in res["1"]] == [2, 2]
    assert [r[1:] for r in res["0"]] == [r[1:] for r in res["1"]]



def test_automatic_choice_follows_the_measured_rule():
    """No format forced (csrc/mk_format.hip pencil_plan; measurements: profiles/r06_march_sizes.txt): the march is taken from 2^21
    rows on where the bricks are at least 90 % full -- aligned or not --, from 2^24 rows on where they are 50 .. 90 % full (L = 100:
    78 %; CG on 200^3 is 17 % faster on the windowed format), and for 5-point matrices from 2^23 rows on; smaller matrices keep
    the windowed formats.  Products stay bit-identical to the CSR gather path whatever was chosen."""
    from pykrylov_amd import _lib, gallery
    lib = _lib.init()
    cases = [((384, 300, 20), 9, 0),          # 2.3 M rows, whole aligned bricks
             ((250, 248, 40), 9, 2),          # 2.5 M rows, bricks 97.7 % full
             ((100, 100, 1700), 9, 2),        # 17 M rows, bricks 78 % full: past 2^24 rows
             ((100, 100, 300), 4, None),      # 3 M rows, bricks 78 % full: windowed
             ((64, 64, 64), 4, None)]         # 262 k rows: windowed
    for dims, want_fmt, want_gen in cases:
        op = gallery.poisson3d(*dims)
        n = op.shape[0]
        x = _lib.DeviceArray.from_numpy(np.random.default_rng(1).standard_normal(n))
        y = _lib.DeviceArray(n)
        op.spmv_device(x.ptr, y.ptr)
        info = march_info(op)
        assert fmt_of(op) == want_fmt, (dims, fmt_of(op), info)
        if want_gen is not None:
            assert info["gen"] == want_gen and (info["L"], info["P"], info["planes"]) == (dims[0], dims[0] * dims[1], dims[2]), info
        got = y.to_numpy()
        _lib.check(lib.mk_csr_set_format(op.handle, 0))
        op.spmv_device(x.ptr, y.ptr)
        assert np.array_equal(got, y.to_numpy()), dims
        for b in (x, y):
            b.free()
        op.free()
    for m, want in ((2000, 4), (3000, 9)):    # 5-point: 4 M rows windowed, 9 M rows marched line by line
        op = gallery.poisson2d(m)
        x = _lib.DeviceArray.from_numpy(np.ones(m * m))
        y = _lib.DeviceArray(m * m)
        op.spmv_device(x.ptr, y.ptr)
        assert fmt_of(op) == want, (m, fmt_of(op))
        for b in (x, y):
            b.free()
        op.free()
