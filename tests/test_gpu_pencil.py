"""Storage format 9 (csrc/mk_spmv_fmt9.h): z-marching bricks for 7-point-class matrices -- every column offset in
{0, +-1, +-L, +-P}, <= 256 distinct values; here on geometries of whole aligned bricks (L % 128 == 0, P % 4L == 0; any other
grid side: tests/test_gpu_march_general.py).  The product must be BIT-identical to the oracle's
scalar left-to-right CSR loop, the fused dots to the oracle run in the brick march's summation order (oracle/gpu_order.py
`pencil`), for every loop that launches a product kernel; matrices outside the class must degrade to the windowed formats."""
import ctypes

import numpy as np
import pytest

from oracle import csr_ref, gpu_order, krylov_ref

pytestmark = pytest.mark.gpu

ROOT = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))


def fmt_of(op):
    from pykrylov_amd import _lib
    fmt = ctypes.c_int32()
    _lib.check(_lib.init().mk_csr_format_info(op.handle, ctypes.byref(fmt), None, None, None, None))
    return fmt.value


def pencil_info(op):
    from pykrylov_amd import _lib
    sl, sp = ctypes.c_int64(), ctypes.c_int64()
    nz, zc, ch, npat = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    _lib.check(_lib.init().mk_csr_pencil_info(op.handle, ctypes.byref(sl), ctypes.byref(sp), ctypes.byref(nz), ctypes.byref(zc),
                                              ctypes.byref(ch), ctypes.byref(npat)))
    return dict(L=sl.value, P=sp.value, planes=nz.value, zc=zc.value, chunks=ch.value, patterns=npat.value)


def op9(A, symmetric=False, fmt=9):
    from pykrylov_amd import CsrOperator, _lib
    op = CsrOperator(A.indptr, A.indices, A.data, A.shape, symmetric=symmetric)
    _lib.check(_lib.init().mk_csr_set_format(op.handle, fmt))
    return op


def banded7(n, L, P, rng, values=(-1.0, 6.0, 0.5, -0.0, 2.0 ** -1060), combos=40, drop=0.0):
    """Any band matrix of the class: offsets {0, +-1, +-L, +-P} wherever the column exists (no grid geometry: the +-1
    entries cross line ends); a row follows one of `combos` random (presence mask, value per offset) combinations -- the
    diagonal always present, the other offsets dropped with probability `drop` -- with values from a small set (zero,
    negative zero and a denormal included)."""
    offs = np.array([-P, -L, -1, 0, 1, L, P])
    mask = rng.random((combos, 7)) >= drop
    mask[:, 3] = True
    vidx = rng.integers(0, len(values), size=(combos, 7))
    which = rng.integers(0, combos, size=n)
    r = np.repeat(np.arange(n), 7)
    k = np.tile(np.arange(7), n)
    c = r + offs[k]
    keep = mask[which[r], k] & (c >= 0) & (c < n)
    vals = np.asarray(values)[vidx[which[r], k]]
    return csr_ref.from_coo(r[keep], c[keep], vals[keep], (n, n))


GRIDS = [(128, 4, 2), (128, 8, 9), (256, 4, 13), (128, 12, 31), (384, 8, 6), (256, 16, 24)]


@pytest.mark.parametrize("dims", GRIDS)
def test_poisson3d_product_bit_exact(dims):
    A = csr_ref.poisson3d(*dims)
    op = op9(A)
    rng = np.random.default_rng(3)
    x = rng.standard_normal(A.shape[1])
    y = op * x
    assert fmt_of(op) == 9
    info = pencil_info(op)
    assert (info["L"], info["P"], info["planes"]) == (dims[0], dims[0] * dims[1], dims[2]) and info["patterns"] <= 27
    assert info["zc"] % 6 == 0 or info["zc"] % 2 == 0
    assert np.array_equal(y, A.matvec(x))
    x[::7] = 0.0
    x[5::11] *= 1e300
    assert np.array_equal(op * x, A.matvec(x))


@pytest.mark.parametrize("n,L,P,drop", [(128 * 4 * 7, 128, 512, 0.0), (256 * 8 * 5, 256, 2048, 0.3), (128 * 8 * 20, 128, 1024, 0.6)])
def test_generic_band_matrix_bit_exact(n, L, P, drop):
    """No grid geometry: +-1 entries across line ends, rows with any subset of the seven offsets, zeros / negative zero /
    denormals among the values, and x with an infinity next to entries a row does not have (a candidate the row has no
    entry for must not reach the sum)."""
    rng = np.random.default_rng(n)
    A = banded7(n, L, P, rng, drop=drop)
    op = op9(A)
    x = rng.standard_normal(n)
    y = op * x
    assert fmt_of(op) == 9, pencil_info(op)
    assert np.array_equal(y, A.matvec(x))
    assert np.array_equal(np.signbit(y), np.signbit(A.matvec(x)))
    if drop > 0:
        # an infinity in x reaches exactly the rows that have an entry in its column
        j = int(rng.integers(P, n - P))
        x[j] = np.inf
        with np.errstate(invalid="ignore"):
            ref = A.matvec(x)
        got = op * x
        assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(np.isinf(got), np.isinf(ref))
        fin = np.isfinite(ref)
        assert np.array_equal(got[fin], ref[fin])


def test_matrices_outside_the_class_degrade():
    from pykrylov_amd import CsrOperator, _lib
    rng = np.random.default_rng(1)
    for A in (csr_ref.poisson3d(132, 9, 8),                   # less than half of the bricks' lanes would have rows
              csr_ref.poisson2d(100),                         # one far stride, shorter than a brick line
              csr_ref.poisson3d_varcoef(128, 8, 8),           # > 256 distinct values (format 10's class: asked for 9 only)
              csr_ref.stencil27(128, 8, 4)):                  # 27 offsets
        for want in (9, 10):
            op = CsrOperator(A.indptr, A.indices, A.data, A.shape)
            _lib.check(_lib.init().mk_csr_set_format(op.handle, want))
            x = rng.standard_normal(A.shape[1])
            assert np.array_equal(op * x, A.matvec(x))
            if want == 10 and A.nnz > 6 * A.shape[0] and A.nnz <= 7 * A.shape[0] and A.shape[0] == 128 * 8 * 8:
                assert fmt_of(op) == 10                        # (the variable-coefficient matrix IS of format 10's class)
            else:
                assert fmt_of(op) not in (9, 10) and pencil_info(op)["L"] == 0


@pytest.mark.parametrize("dims", [(128, 8, 9), (256, 8, 26)])
def test_cg_bit_exact_in_brick_order(dims):
    """CG with its <p, Ap> fused into the brick march: history, iterate and matvec count equal the oracle run with the
    device's summation order (pencil order for the fused dot, stream order for the others), bit for bit."""
    from pykrylov_amd import CG
    A = csr_ref.poisson3d(*dims)
    op = op9(A, symmetric=True)
    rhs = A.matvec(np.ones(A.shape[0]))
    s = CG(op)
    s.solve(rhs)
    assert fmt_of(op) == 9
    geo = gpu_order.launch_geometry(op)
    assert geo[1][0] == "pencil"
    dots = gpu_order.GpuDots(A.shape[0], gpu_order.SPMV_SITES["cg"], geometry=geo)
    ref = krylov_ref.cg(A, rhs, red=krylov_ref.Reductions(dots))
    assert s.nMatvec == ref["nMatvec"]
    assert np.array_equal(np.array(s.residHistory), ref["residHistory"])
    assert np.array_equal(s.x, ref["x"])


def test_minres_and_bicgstab_on_format_9():
    """Epilogues with an on-the-fly input scaling (MINRES' v = y / beta through `xin`) and with several fused dots
    (BiCGSTAB) run on the brick march: bit-exact against the oracle in the device's order."""
    from pykrylov_amd import Minres, BiCGSTAB
    A = csr_ref.poisson3d(128, 8, 12)
    n = A.shape[0]
    rhs = A.matvec(np.ones(n))
    op = op9(A, symmetric=True)
    geo = gpu_order.launch_geometry(op)
    m = Minres(op)
    m.solve(rhs, show=False, check=False, shift=1.5, etol=0.0, rtol=1e-10, itnlim=60)
    ref = krylov_ref.minres(A, rhs, shift=1.5, etol=0.0, rtol=1e-10, itnlim=60, check=False,
                            red=krylov_ref.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES["minres"], geometry=geo)))
    assert fmt_of(op) == 9
    assert m.itn == ref["itn"] and np.array_equal(np.array(m.residHistory), np.array(ref["residHistory"]))
    assert np.array_equal(m.x, ref["x"])
    b = BiCGSTAB(op)
    b.solve(rhs, matvec_max=40)
    refb = krylov_ref.bicgstab(A, rhs, matvec_max=40,
                               red=krylov_ref.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES["bicgstab"], geometry=geo)))
    assert b.nMatvec == refb["nMatvec"] and b.residNorm == refb["residNorm"] and np.array_equal(b.x, refb["x"])


def test_composed_operator_on_format_9():
    """Row programs (alpha A + D x) ride on the brick march through mk_rowprog."""
    from pykrylov_amd.linop import DiagonalOperator
    A = csr_ref.poisson3d(128, 4, 7)
    n = A.shape[0]
    op = op9(A)
    rng = np.random.default_rng(2)
    d = rng.standard_normal(n)
    x = rng.standard_normal(n)
    comp = 2.5 * op + DiagonalOperator(d)
    y = comp * x
    assert fmt_of(op) == 9
    assert np.array_equal(y, 2.5 * A.matvec(x) + d * x)


def _solver_is_fused(op, rhs):
    from pykrylov_amd import _lib
    from pykrylov_amd.generic import DeviceRun
    with DeviceRun(op, _lib.MK_CG, rhs, None, abstol=1e-8, reltol=1e-6, matvec_max=10, check_curvature=1) as run:
        run.setup()
        f = ctypes.c_int32()
        _lib.check(run.lib.mk_solver_fused(run.handle, ctypes.byref(f)))
    return bool(f.value)


@pytest.mark.parametrize("dims", [(128, 8, 9), (128, 4, 14), (256, 8, 26)])
def test_fused_cg_passes_change_no_bit(dims, monkeypatch):
    """CG on a format-9 matrix runs FUSED passes (the x / p update of a pass inside the next pass's product kernel,
    csrc/mk_cg.hip).  Same operations on the same values: history, iterate, matvec count, residual norm and the search
    direction equal the three-kernel pass bit for bit -- with default stopping, with matvec_max cutting the run short
    (the pending update of the last pass must still land), with an initial guess, with a diagonal preconditioner and
    with store_iterates (the iterate as a caller sees it BETWEEN passes)."""
    from pykrylov_amd import CG
    from pykrylov_amd.linop import DiagonalOperator
    A = csr_ref.poisson3d(*dims)
    n = A.shape[0]
    rng = np.random.default_rng(4)
    rhs = A.matvec(np.ones(n)) + 0.1 * rng.standard_normal(n)
    guess = rng.standard_normal(n)
    prec = DiagonalOperator(1.0 / (6.0 + rng.random(n)))
    cases = [dict(), dict(matvec_max=7), dict(guess=guess, matvec_max=23), dict(store_iterates=True, matvec_max=12),
             dict(store_resids=True, matvec_max=9)]
    out = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("MK_CG_FUSE", fuse)
        op = op9(A, symmetric=True)
        assert _solver_is_fused(op, rhs) == (fuse == "1")
        res = []
        for kw in cases:
            s = CG(op)
            s.solve(rhs, **kw)
            res.append((s.nMatvec, np.array(s.residHistory), s.x.copy(), float(s.residNorm), bool(s.converged),
                        [np.array(v) for v in getattr(s, "iterates", [])], [np.array(v) for v in getattr(s, "resids", [])]))
        s = CG(op, precon=prec)
        s.solve(rhs, matvec_max=40)
        res.append((s.nMatvec, np.array(s.residHistory), s.x.copy(), float(s.residNorm), bool(s.converged), [], []))
        assert fmt_of(op) == 9
        out[fuse] = res
    for a, b in zip(out["1"], out["0"]):
        assert a[0] == b[0] and a[3] == b[3] and a[4] == b[4]
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        assert len(a[5]) == len(b[5]) and all(np.array_equal(u, v) for u, v in zip(a[5], b[5]))
        assert len(a[6]) == len(b[6]) and all(np.array_equal(u, v) for u, v in zip(a[6], b[6]))
    assert len(out["1"][3][5]) >= 12                                    # (the store_iterates case really kept iterates)


# ---------------------------------------------------------------------------------------------------------------------
# storage format 10: the brick march with streamed values (matrices of the class without a value dictionary)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dims", [(128, 4, 2), (128, 8, 9), (256, 4, 13), (128, 12, 31), (256, 16, 24)])
def test_format10_variable_coefficients_product_bit_exact(dims):
    A = csr_ref.poisson3d_varcoef(*dims, seed=7)
    op = op9(A, fmt=10)
    rng = np.random.default_rng(5)
    x = rng.standard_normal(A.shape[1])
    y = op * x
    assert fmt_of(op) == 10
    info = pencil_info(op)
    assert (info["L"], info["P"], info["planes"]) == (dims[0], dims[0] * dims[1], dims[2])
    assert np.array_equal(y, A.matvec(x))
    x[::5] = 0.0
    x[3::7] *= -1e200
    assert np.array_equal(op * x, A.matvec(x))


def test_format10_generic_band_with_many_values():
    """Any subset of the seven offsets per row, every value different (no dictionary), an infinity in x next to entries a
    row does not have."""
    n, L, P = 128 * 8 * 12, 128, 1024
    rng = np.random.default_rng(11)
    A = banded7(n, L, P, rng, combos=60, drop=0.4)
    A.data[:] = rng.standard_normal(A.nnz)                   # (all different: format 9 cannot take it)
    op = op9(A, fmt=10)
    x = rng.standard_normal(n)
    assert np.array_equal(op * x, A.matvec(x)) and fmt_of(op) == 10
    j = int(rng.integers(P, n - P))
    x[j] = np.inf
    with np.errstate(invalid="ignore"):
        ref = A.matvec(x)
    got = op * x
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(np.isinf(got), np.isinf(ref))
    fin = np.isfinite(ref)
    assert np.array_equal(got[fin], ref[fin])


@pytest.mark.parametrize("dims", [(128, 8, 9), (256, 8, 26)])
def test_format10_cg_fused_bit_exact_and_equal_to_the_three_kernel_pass(dims, monkeypatch):
    from pykrylov_amd import CG
    A = csr_ref.poisson3d_varcoef(*dims, seed=7)
    n = A.shape[0]
    rhs = A.matvec(np.ones(n))
    runs = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("MK_CG_FUSE", fuse)
        op = op9(A, symmetric=True, fmt=10)
        assert _solver_is_fused(op, rhs) == (fuse == "1")
        s = CG(op)
        s.solve(rhs, matvec_max=150)
        assert fmt_of(op) == 10
        runs[fuse] = (s.nMatvec, np.array(s.residHistory), s.x.copy())
        geo = gpu_order.launch_geometry(op)
    assert runs["1"][0] == runs["0"][0] and np.array_equal(runs["1"][1], runs["0"][1]) and np.array_equal(runs["1"][2], runs["0"][2])
    ref = krylov_ref.cg(A, rhs, matvec_max=150, red=krylov_ref.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES["cg"], geometry=geo)))
    assert runs["1"][0] == ref["nMatvec"] and np.array_equal(runs["1"][1], ref["residHistory"]) and np.array_equal(runs["1"][2], ref["x"])


# ---------------------------------------------------------------------------------------------------------------------
# storage format 11: format 10 for matrices that are symmetric bit for bit (the diagonal and the upper values only)
# ---------------------------------------------------------------------------------------------------------------------
def sym_banded(n, L, P, rng, drop=0.3):
    """A symmetric band matrix of the class with all values different: upper entries (r, r + off), off in {1, L, P}, kept with
    probability 1 - drop and mirrored; a diagonal everywhere (a few zeros, a negative zero, a denormal among them)."""
    rows, cols, vals = [np.arange(n)], [np.arange(n)], [rng.standard_normal(n)]
    vals[0][::97] = 0.0
    vals[0][5::101] = -0.0
    vals[0][7::103] = 2.0 ** -1060
    for off in (1, L, P):
        r = np.arange(n - off)
        keep = rng.random(n - off) >= drop
        v = rng.standard_normal(n - off)
        rows += [r[keep], r[keep] + off]
        cols += [r[keep] + off, r[keep]]
        vals += [v[keep], v[keep]]
    return csr_ref.from_coo(np.concatenate(rows), np.concatenate(cols), np.concatenate(vals), (n, n))


@pytest.mark.parametrize("dims", [(128, 4, 2), (128, 8, 9), (256, 4, 13), (128, 12, 31), (256, 16, 24)])
def test_format11_symmetric_product_bit_exact(dims):
    A = csr_ref.poisson3d_varcoef(*dims, seed=7)
    op = op9(A, fmt=11)
    rng = np.random.default_rng(5)
    x = rng.standard_normal(A.shape[1])
    y = op * x
    assert fmt_of(op) == 11
    assert np.array_equal(y, A.matvec(x))
    x[::5] = 0.0
    x[3::7] *= -1e200
    assert np.array_equal(op * x, A.matvec(x))


@pytest.mark.parametrize("drop", [0.0, 0.3, 0.7])
def test_format11_generic_symmetric_band(drop):
    """No grid geometry (the +-1 entries cross line ends), any symmetric subset of the offsets, every value different; an
    infinity in x next to entries a row does not have; the same matrix made unsymmetric in ONE value stays format 10."""
    n, L, P = 128 * 8 * 11, 128, 1024
    rng = np.random.default_rng(17)
    A = sym_banded(n, L, P, rng, drop=drop)
    op = op9(A, fmt=11)
    x = rng.standard_normal(n)
    assert np.array_equal(op * x, A.matvec(x)) and fmt_of(op) == 11
    j = int(rng.integers(P, n - P))
    x[j] = np.inf
    with np.errstate(invalid="ignore"):
        ref = A.matvec(x)
    got = op * x
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(np.isinf(got), np.isinf(ref))
    fin = np.isfinite(ref)
    assert np.array_equal(got[fin], ref[fin])
    B = csr_ref.RefCsr(A.indptr, A.indices, A.data.copy(), A.shape)
    k = int(np.nonzero(B.indices[: B.indptr[n // 2]] > np.repeat(np.arange(n // 2), np.diff(B.indptr[: n // 2 + 1])))[0][-1])
    B.data[k] = np.nextafter(B.data[k], np.inf)              # one upper value one ulp off its mirror
    opb = op9(B, fmt=11)
    x[j] = 1.0
    assert np.array_equal(opb * x, B.matvec(x)) and fmt_of(opb) == 10


@pytest.mark.parametrize("dims", [(128, 8, 9), (256, 8, 26)])
def test_format11_cg_equals_format10_bit_for_bit(dims, monkeypatch):
    """Same rows, same bits, same workgroup shares: CG on format 11 (fused passes and three-kernel passes) is the run on
    format 10 to the last bit of the history and of the iterate."""
    from pykrylov_amd import CG
    A = csr_ref.poisson3d_varcoef(*dims, seed=7)
    n = A.shape[0]
    rhs = A.matvec(np.ones(n))
    runs = {}
    for fmt in (11, 10):
        for fuse in ("1", "0"):
            monkeypatch.setenv("MK_CG_FUSE", fuse)
            op = op9(A, symmetric=True, fmt=fmt)
            assert _solver_is_fused(op, rhs) == (fuse == "1")
            s = CG(op)
            s.solve(rhs, matvec_max=150)
            assert fmt_of(op) == fmt
            runs[fmt, fuse] = (s.nMatvec, np.array(s.residHistory), s.x.copy())
    base = runs[10, "0"]
    for key, r in runs.items():
        assert r[0] == base[0] and np.array_equal(r[1], base[1]) and np.array_equal(r[2], base[2]), key


def test_automatic_march_follows_the_loop(monkeypatch):
    """The brick march is chosen per matrix but pays only for loops whose product epilogue loads nothing (CG, plain
    products): an automatically formatted matrix of the class is in format 9 for a plain product and for CG, goes back to
    the windowed pattern format when a MINRES solver is created on it, and returns to format 9 for the next CG -- with the
    same bits as the forced formats every time.  (MK_PENCIL_MIN_ROWS lowered so that a small grid qualifies.)"""
    from pykrylov_amd import CG, Minres, CsrOperator
    monkeypatch.setenv("MK_PENCIL_MIN_ROWS", "1024")
    # (the threshold is read once per process: run in a child so that the variable is seen)
    import subprocess
    import sys
    code = r'''
import sys, ctypes
import numpy as np
sys.path.insert(0, %r)
from oracle import csr_ref
from pykrylov_amd import CG, Minres, CsrOperator, _lib
def fmt_of(op):
    f = ctypes.c_int32()
    _lib.check(_lib.init().mk_csr_format_info(op.handle, ctypes.byref(f), None, None, None, None))
    return f.value
A = csr_ref.poisson3d(128, 8, 12)
n = A.shape[0]
rhs = A.matvec(np.ones(n))
op = CsrOperator(A.indptr, A.indices, A.data, A.shape, symmetric=True)
x = np.random.default_rng(0).standard_normal(n)
assert np.array_equal(op * x, A.matvec(x)) and fmt_of(op) == 9
c = CG(op); c.solve(rhs); assert fmt_of(op) == 9
m = Minres(op); m.solve(rhs, show=False, check=False, etol=0.0, rtol=1e-10); assert fmt_of(op) == 4, fmt_of(op)
c2 = CG(op); c2.solve(rhs); assert fmt_of(op) == 9
assert c.nMatvec == c2.nMatvec and np.array_equal(np.array(c.residHistory), np.array(c2.residHistory)) and np.array_equal(c.x, c2.x)
forced = CsrOperator(A.indptr, A.indices, A.data, A.shape, symmetric=True)
_lib.check(_lib.init().mk_csr_set_format(forced.handle, 4))
m4 = Minres(forced); m4.solve(rhs, show=False, check=False, etol=0.0, rtol=1e-10)
assert m.itn == m4.itn and np.array_equal(np.array(m.residHistory), np.array(m4.residHistory)) and np.array_equal(m.x, m4.x)
print("OK")
''' % ROOT
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(__import__("os").environ, MK_PENCIL_MIN_ROWS="1024"))
    assert p.returncode == 0 and p.stdout.strip().endswith("OK"), p.stdout[-2000:] + p.stderr[-3000:]


def test_fused_cg_negative_curvature_leaves_nothing_pending(monkeypatch):
    """An indefinite matrix of the class: CG meets p'Ap <= 0 after a few passes (cg.py:119-124), the pass is abandoned BEFORE
    its x update -- the fused loop must not apply a pending update then.  Fused, three-kernel and oracle agree bit for bit."""
    from pykrylov_amd import CG
    A = csr_ref.poisson3d(128, 8, 10)
    n = A.shape[0]
    diag = A.indices == np.repeat(np.arange(n), np.diff(A.indptr))
    data = A.data.copy()
    data[diag] = 2.5                                         # (inside the spectrum of the off-diagonal part: indefinite)
    B = csr_ref.RefCsr(A.indptr, A.indices, data, A.shape)
    rhs = B.matvec(np.cos(np.arange(n)))
    got = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("MK_CG_FUSE", fuse)
        op = op9(B, symmetric=True)
        s = CG(op)
        s.solve(rhs, matvec_max=60)
        assert fmt_of(op) == 9
        got[fuse] = (s.nMatvec, np.array(s.residHistory), s.x.copy(), s.definite, getattr(s, "infiniteDescent", None))
        geo = gpu_order.launch_geometry(op)
    assert got["1"][3] is False or got["1"][3] == 0          # negative curvature was met
    assert got["1"][0] == got["0"][0] and np.array_equal(got["1"][1], got["0"][1]) and np.array_equal(got["1"][2], got["0"][2])
    ref = krylov_ref.cg(B, rhs, matvec_max=60, red=krylov_ref.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES["cg"], geometry=geo)))
    assert got["1"][0] == ref["nMatvec"] and np.array_equal(got["1"][1], ref["residHistory"]) and np.array_equal(got["1"][2], ref["x"])


@pytest.mark.parametrize("fmt", [9, 10])
@pytest.mark.parametrize("solver", ["bicgstab", "cgs", "tfqmr", "minres", "symmlq"])
def test_every_square_loop_on_the_march_bit_exact(solver, fmt):
    """The other five square loops on the brick march (formats 9 and 10): their product epilogues take the vectors they
    read at their own row as operands loaded at the top of a step (`row_pf` / `row_x_pf`, csrc/mk_device.h) -- same
    operations, same order: history / iterate / counters equal the oracle run in the device's summation order, bit for bit."""
    import pykrylov_amd as pk
    A = csr_ref.poisson3d(128, 8, 14) if fmt == 9 else csr_ref.poisson3d_varcoef(128, 8, 14, seed=7)
    n = A.shape[0]
    rhs = A.matvec(np.ones(n)) + 0.05 * np.sin(np.arange(n))
    op = op9(A, symmetric=True, fmt=fmt)
    x0 = np.random.default_rng(8).standard_normal(n)
    assert np.array_equal(op * x0, A.matvec(x0)) and fmt_of(op) == fmt
    geo = gpu_order.launch_geometry(op)
    red = krylov_ref.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES[solver], geometry=geo))
    if solver in ("bicgstab", "cgs", "tfqmr"):
        s = {"bicgstab": pk.BiCGSTAB, "cgs": pk.CGS, "tfqmr": pk.TFQMR}[solver](op)
        s.solve(rhs, matvec_max=36)
        ref = getattr(krylov_ref, solver)(A, rhs, matvec_max=36, red=red)
        assert s.nMatvec == ref["nMatvec"] and s.residNorm == ref["residNorm"]
        assert np.array_equal(s.x, ref["x"])
    elif solver == "minres":
        s = pk.Minres(op)
        s.solve(rhs, show=False, check=False, shift=0.7, etol=0.0, rtol=1e-10, itnlim=40)
        ref = krylov_ref.minres(A, rhs, shift=0.7, etol=0.0, rtol=1e-10, itnlim=40, check=False, red=red)
        assert s.itn == ref["itn"] and np.array_equal(np.array(s.residHistory), np.array(ref["residHistory"]))
        assert np.array_equal(s.x, ref["x"])
    else:
        s = pk.Symmlq(op)
        s.solve(rhs, matvec_max=40, shift=0.7)
        ref = krylov_ref.symmlq(A, rhs, matvec_max=40, shift=0.7, red=red)
        assert s.nMatvec == ref["nMatvec"] and s.residNorm == ref["residNorm"]
        assert np.array_equal(s.x, ref["x"])
    assert fmt_of(op) == fmt


def test_format11_is_cgs_other_loops_get_format10():
    """Format 11 has kernels for plain products and CG only (compile-time budget, csrc/mk_device.h MkSymMarch): a matrix asked
    into format 11 serves any other loop as format 10 -- same bits as a matrix asked into format 10 -- and returns to 11 for CG."""
    import pykrylov_amd as pk
    A = csr_ref.poisson3d_varcoef(128, 8, 10, seed=7)
    n = A.shape[0]
    rhs = A.matvec(np.ones(n)) + 0.05 * np.sin(np.arange(n))
    op = op9(A, symmetric=True, fmt=11)
    x0 = np.random.default_rng(8).standard_normal(n)
    assert np.array_equal(op * x0, A.matvec(x0)) and fmt_of(op) == 11
    b = pk.BiCGSTAB(op)
    b.solve(rhs, matvec_max=30)
    assert fmt_of(op) == 10
    op10 = op9(A, symmetric=True, fmt=10)
    b10 = pk.BiCGSTAB(op10)
    b10.solve(rhs, matvec_max=30)
    assert b.nMatvec == b10.nMatvec and b.residNorm == b10.residNorm and np.array_equal(b.x, b10.x)
    c = pk.CG(op)
    c.solve(rhs, matvec_max=30)
    assert fmt_of(op) == 11
    c10 = pk.CG(op10)
    c10.solve(rhs, matvec_max=30)
    assert c.nMatvec == c10.nMatvec and np.array_equal(np.array(c.residHistory), np.array(c10.residHistory)) and np.array_equal(c.x, c10.x)


def test_least_squares_loop_on_a_matrix_forced_into_a_march_format():
    """The least-squares loops have no march kernels (their operators are rectangular; MkNoMarch): on a square matrix that a
    caller forced into format 9 their products take the CSR gather kernel on the same arrays -- same row sums, so the run
    follows the one on format 0 to rounding (the dots group differently)."""
    from pykrylov_amd import lls
    A = csr_ref.poisson3d(128, 4, 6)
    n = A.shape[0]
    b = A.matvec(np.ones(n)) + 0.1 * np.cos(np.arange(n))
    runs = []
    for fmt in (9, 0):
        op = op9(A, symmetric=True, fmt=fmt)
        s = lls.LSQRFramework(op)
        s.solve(b, itnlim=40, show=False)
        runs.append((s.itn, s.x.copy(), float(s.r1norm)))
        x0 = np.random.default_rng(1).standard_normal(n)
        assert np.array_equal(op * x0, A.matvec(x0)) and fmt_of(op) == fmt
    assert runs[0][0] == runs[1][0] == 40
    assert np.linalg.norm(runs[0][1] - runs[1][1]) <= 1e-10 * np.linalg.norm(runs[1][1])
    assert abs(runs[0][2] - runs[1][2]) <= 1e-10 * runs[1][2]
