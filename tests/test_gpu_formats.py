"""The storage formats of the SpMV (csrc/mk_format.hip, mk_device.h): plain CSR gathers (0), windowed tiles (1),
windowed tiles + value dictionary (2), LDS-resident tiles with column phases (3), row patterns (4).  Whatever format a tile ends up in, the product must be BIT-identical to the
scalar left-to-right CSR loop of the oracle -- including matrices that mix eligible and ineligible tiles, so that
both code paths run inside one launch and hand over to each other."""
import ctypes

import numpy as np
import pytest

from oracle import csr_ref

pytestmark = pytest.mark.gpu


def fmt_info(op):
    from pykrylov_amd import _lib
    fmt, chunks, nd = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    tiles, mb = ctypes.c_int64(), ctypes.c_int64()
    _lib.check(_lib.init().mk_csr_format_info(op.handle, ctypes.byref(fmt), ctypes.byref(tiles), ctypes.byref(chunks),
                                              ctypes.byref(nd), ctypes.byref(mb)))
    return dict(fmt=fmt.value, tiles=tiles.value, chunks=chunks.value, ndict=nd.value, bytes=mb.value)


def op_with_format(A, fmt):
    from pykrylov_amd import CsrOperator, _lib
    op = CsrOperator(A.indptr, A.indices, A.data, A.shape)
    _lib.check(_lib.init().mk_csr_set_format(op.handle, fmt))
    return op


def banded(n, offsets, rng, few_values=False, ncols=None):
    ncols = ncols or n
    rows, cols = [], []
    for o in offsets:
        r = np.arange(max(0, -o), min(n, ncols - o))
        rows.append(r)
        cols.append(r + o)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    vals = rng.choice([-1.0, 4.0, 0.5, -0.0], size=len(rows)) if few_values else rng.standard_normal(len(rows))
    return rows, cols, vals


def matrices():
    rng = np.random.default_rng(5)
    out = {}
    n = 9000
    r, c, v = banded(n, (-300, -1, 0, 1, 300), rng, few_values=True)
    out["banded_dict"] = (csr_ref.from_coo(r, c, v, (n, n)), 2)
    r, c, v = banded(n, (-300, -1, 0, 1, 300), rng)
    out["banded_manyvalues"] = (csr_ref.from_coo(r, c, v, (n, n)), 1)          # > 256 distinct values: no dictionary
    # a few dense rows: their tiles exceed 2048 nonzeros -> gather path between windowed tiles
    dr = np.repeat([700, 701, 5000], 3000)
    dc = np.concatenate([rng.choice(n, 3000, replace=False) for _ in range(3)])
    r2, c2, v2 = np.concatenate([r, dr]), np.concatenate([c, dc]), np.concatenate([v, rng.standard_normal(9000)])
    out["banded_plus_dense_rows"] = (csr_ref.from_coo(r2, c2, v2, (n, n)), 1)
    # a block of rows with scattered columns in the middle (cover fails there), banded elsewhere
    sr = np.repeat(np.arange(2048, 3072), 6)
    sc = rng.integers(0, n, size=len(sr))
    r3, c3 = np.concatenate([r, sr]), np.concatenate([c, sc])
    v3 = rng.choice([1.0, -2.0, 3.0], size=len(r3))
    out["banded_plus_scattered_block"] = (csr_ref.from_coo(r3, c3, v3, (n, n)), 2)
    # rectangular, odd number of columns, last column referenced (its pair would leave x: that tile must gather)
    m, k = 5000, 4097
    r4, c4, v4 = banded(m, (0, 1, 2, 40), rng, few_values=True, ncols=k)
    r4 = np.concatenate([r4, np.arange(3000, 3010)])
    c4 = np.concatenate([c4, np.full(10, k - 1)])
    v4 = np.concatenate([v4, np.ones(10)])
    out["rect_odd_cols"] = (csr_ref.from_coo(r4, c4, v4, (m, k)), 2)
    # empty rows and empty tiles
    r5, c5, v5 = banded(n, (-2, 0, 7), rng, few_values=True)
    keep = (r5 < 1000) | (r5 >= 2000)
    keep &= (r5 % 5 != 0)
    out["empty_rows_and_tiles"] = (csr_ref.from_coo(r5[keep], c5[keep], v5[keep], (n, n)), 2)
    # scattered everywhere: the builder gives up, plain CSR
    out["random"] = (csr_ref.random_diagdom(6000, seed=9), 0)
    # exactly 256 / 257 distinct values
    for nv, want in ((256, 2), (257, 1)):
        r6, c6, _ = banded(4096, (-1, 0, 1), rng)
        v6 = np.arange(len(r6)) % nv + 1.0
        out["distinct_%d" % nv] = (csr_ref.from_coo(r6, c6, v6, (4096, 4096)), want)
    return out


MATS = matrices()


@pytest.mark.parametrize("name", sorted(MATS))
@pytest.mark.parametrize("fmt", [0, 1, 2])
def test_spmv_bit_exact_in_every_format(name, fmt):
    A, best = MATS[name]
    op = op_with_format(A, fmt)
    info = fmt_info(op)
    assert info["fmt"] == min(fmt, best), (name, fmt, info)           # requests degrade 2 -> 1 -> 0
    if info["fmt"]:
        assert 0 < info["tiles"] <= (A.shape[0] + 255) // 256 and 0 < info["chunks"] <= 16
        assert info["bytes"] < 12 * A.nnz + 4 * (A.shape[0] + 1) + 80 * ((A.shape[0] + 255) // 256) + 1
    if info["fmt"] == 2:
        assert info["ndict"] == len(np.unique(A.data.view(np.int64)))
    rng = np.random.default_rng(3)
    for x in (np.ones(A.shape[1]), rng.standard_normal(A.shape[1]), 1e200 * rng.standard_normal(A.shape[1])):
        assert np.array_equal(op * x, A.matvec(x))
    op.free()


@pytest.mark.parametrize("name", sorted(MATS))
def test_resident_tile_format_bit_exact(name):
    """Format 3 (tile resident in LDS, gathers ordered by column block) is meant for long x vectors; forced here on
    small matrices, with several phase counts, it must still give the scalar loop's bits.  Tiles longer than the
    LDS budget (the dense rows) keep the matrix on the chunked gather path."""
    A, _ = MATS[name]
    op = op_with_format(A, 3)
    info = fmt_info(op)
    ip = np.asarray(A.indptr, dtype=np.int64)
    starts = np.arange(0, A.shape[0], 256)
    ends = np.minimum(starts + 256, A.shape[0])
    longest = int(np.max(ip[ends] - (ip[starts] & ~3)))          # tile streams as the kernel copies them
    assert info["fmt"] == (3 if longest <= 2560 else 0), (name, info, longest)
    assert (name in ("banded_plus_dense_rows", "banded_plus_scattered_block")) == (longest > 2560)
    rng = np.random.default_rng(3)
    for x in (np.ones(A.shape[1]), rng.standard_normal(A.shape[1]), 1e200 * rng.standard_normal(A.shape[1])):
        assert np.array_equal(op * x, A.matvec(x))
    assert np.array_equal(op_with_format(A.transpose(), 3) * np.ones(A.shape[0]), A.rmatvec(np.ones(A.shape[0])))
    op.free()


@pytest.mark.parametrize("phases", ["1", "3", "7", "64"])
def test_resident_tile_phase_counts(phases, monkeypatch):
    """The phase count only reorders WHEN a row's entries are gathered, never the order they are added in."""
    import subprocess, sys, os
    code = ("import numpy as np, ctypes\n"
            "from oracle import csr_ref\n"
            "from pykrylov_amd import CsrOperator, _lib\n"
            "A = csr_ref.random_diagdom(20011, seed=2)\n"
            "op = CsrOperator(A.indptr, A.indices, A.data, A.shape)\n"
            "_lib.check(_lib.init().mk_csr_set_format(op.handle, 3))\n"
            "x = np.random.default_rng(1).standard_normal(20011)\n"
            "assert np.array_equal(op * x, A.matvec(x))\n"
            "f, k = ctypes.c_int32(), ctypes.c_int32()\n"
            "_lib.check(_lib.load().mk_csr_format_info(op.handle, ctypes.byref(f), None, ctypes.byref(k), None, None))\n"
            "print('FMT', f.value, k.value)\n")
    env = dict(os.environ, MK_RT_PHASES=phases)              # (read once per process)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=root, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "FMT 3 %s" % phases in p.stdout, p.stdout


# what the pattern builder (format 4) makes of the test matrices when asked: 4 where every row of every windowed tile
# has <= 32 entries and the rows fall into <= 256 / 128 / 64 patterns, else the request degrades to the dictionary format
# (banded_dict: random value codes, 4^5 combinations per offset pattern -- too many)
PATTERNED = {"banded_dict": 2, "banded_manyvalues": 1, "banded_plus_dense_rows": 1, "banded_plus_scattered_block": None,
             "rect_odd_cols": None, "empty_rows_and_tiles": None, "random": 0, "distinct_256": None, "distinct_257": 1}


@pytest.mark.parametrize("name", sorted(MATS))
def test_row_pattern_format_bit_exact(name):
    """Format 4: one pattern byte per row instead of one packed word per nonzero.  Whatever the builder decides
    (patterns, or falling back to 2 / 1 / 0), the product keeps the scalar loop's bits -- also through the transpose."""
    A, best = MATS[name]
    op = op_with_format(A, 4)
    info = fmt_info(op)
    assert info["fmt"] in ((4, 2) if best == 2 else (best,)), (name, info)
    if PATTERNED[name] is not None:
        assert info["fmt"] == PATTERNED[name], (name, info)
    rng = np.random.default_rng(3)
    for x in (np.ones(A.shape[1]), rng.standard_normal(A.shape[1]), 1e200 * rng.standard_normal(A.shape[1])):
        assert np.array_equal(op * x, A.matvec(x))
    u = rng.standard_normal(A.shape[0])
    assert np.array_equal(op.T * u, A.rmatvec(u))
    op.free()


def test_row_patterns_on_stencils():
    """The matrices the format is made for: 5- and 7-point stencils (all BASELINE configs), shifted, composed,
    with ragged last tiles; and a matrix with a few longer rows (16-word patterns)."""
    from pykrylov_amd import CsrOperator, IdentityOperator, gallery
    rng = np.random.default_rng(0)
    for A, sure in ((csr_ref.poisson2d(256), True), (csr_ref.poisson3d(32), True), (csr_ref.poisson1d(70001), True),
                    (csr_ref.poisson2d(150), False), (csr_ref.poisson3d(30), False), (csr_ref.poisson3d(17, 23, 9), False)):
        op = CsrOperator(A.indptr, A.indices, A.data, A.shape)
        info = fmt_info(op)
        # (grid lines that do not divide the 256-row tiles give every tile its own window layout: such matrices may
        #  exceed the 128 patterns the table holds and then stay in the dictionary format)
        assert info["fmt"] in ((4,) if sure else (2, 4)), info
        if info["fmt"] == 4:
            assert info["bytes"] < 4 * A.nnz, info                       # (less than the dictionary format streams)
        x = rng.standard_normal(A.shape[1])
        assert np.array_equal(op * x, A.matvec(x))
        sh = op - 1.5 * IdentityOperator(A.shape[0])
        assert np.array_equal(sh * x, A.matvec(x) - 1.5 * x)
        op.free()
    # rows longer than 8 entries use 16-word patterns: a tridiagonal matrix whose every 16th row carries 12 entries
    n = 20000
    r, c, v = banded(n, (-1, 0, 1), rng, few_values=True)
    v[:] = np.where(r == c, 4.0, -1.0)
    long_rows = np.arange(32, n - 32, 16)
    offs = np.array([-9, -7, -5, -4, -3, 3, 4, 5, 7], dtype=np.int64)
    r2 = np.repeat(long_rows, len(offs))
    c2 = (long_rows[:, None] + offs[None, :]).ravel()
    L = csr_ref.from_coo(np.concatenate([r, r2]), np.concatenate([c, c2]),
                         np.concatenate([v, np.full(len(r2), 0.5)]), (n, n))
    op = CsrOperator(L.indptr, L.indices, L.data, L.shape)
    info = fmt_info(op)
    assert info["fmt"] == 4 and int(np.max(np.diff(L.indptr))) == 12, info
    x = rng.standard_normal(n)
    assert np.array_equal(op * x, L.matvec(x)) and np.array_equal(op.T * x, L.rmatvec(x))
    op.free()


def test_mixed_tiles_really_mix():
    A, _ = MATS["banded_plus_dense_rows"]
    op = op_with_format(A, 1)
    info = fmt_info(op)
    assert 0 < info["tiles"] < (A.shape[0] + 255) // 256            # some tiles windowed, some on the gather path
    op.free()


@pytest.mark.parametrize("fmt", [0, 1, 2, 4])
def test_solver_results_do_not_depend_on_the_format(fmt):
    """CG on a 2-D Poisson matrix: identical bits (history, iterate) in all windowed formats."""
    from pykrylov_amd import CG
    A = csr_ref.poisson2d(150)
    rhs = A.matvec(np.ones(A.shape[0]))
    op0 = op_with_format(A, 0)
    s0 = CG(op0)
    s0.solve(rhs)
    op = op_with_format(A, fmt)
    assert fmt_info(op)["fmt"] == fmt
    s = CG(op)
    s.solve(rhs)
    assert s.nMatvec == s0.nMatvec and np.array_equal(s.residHistory, s0.residHistory) and np.array_equal(s.x, s0.x)
    op.free()
    op0.free()


def test_transpose_and_composed_operators_use_the_format():
    """A.T builds its own windows; a composed operator (alpha*A + D) shares its base matrix's."""
    from pykrylov_amd import CsrOperator, IdentityOperator
    A, _ = MATS["banded_dict"]
    op = CsrOperator(A.indptr, A.indices, A.data, A.shape)
    x = np.random.default_rng(1).standard_normal(A.shape[0])
    assert np.array_equal(op.T * x, A.rmatvec(x))
    assert fmt_info(op.T)["fmt"] in (2, 4)                          # (4 when the rows follow few enough patterns)
    shifted = op - 1.5 * IdentityOperator(A.shape[0])
    assert np.array_equal(shifted * x, A.matvec(x) - 1.5 * x)
    op.free()


# ------------------------------------------------------------------------------------------------ column blocks
def blocked(op, kb=2048):
    """Turn column blocks on for this operator (they are off by default)."""
    from pykrylov_amd import _lib
    _lib.check(_lib.init().mk_csr_set_colblocks(op.handle, kb))
    return op


def colblocks(op):
    from pykrylov_amd import _lib
    k = ctypes.c_int32()
    _lib.check(_lib.init().mk_csr_colblocks(op.handle, ctypes.byref(k)))
    return k.value


@pytest.fixture(scope="module")
def big_random():
    """Scattered columns and an x vector of 5.6 MB: does not fit an XCD's L2, so the product is column-blocked."""
    return csr_ref.random_diagdom(700001, seed=4)


def test_column_blocked_product_is_bit_exact(big_random):
    from pykrylov_amd import CsrOperator, IdentityOperator
    A = big_random
    n = A.shape[0]
    op = CsrOperator(A.indptr, A.indices, A.data, A.shape)
    info = fmt_info(op)                                            # x = 5.6 MB: resident tiles, 6 column phases of <= 1 MiB
    assert info["fmt"] == 3 and info["chunks"] == 6 and colblocks(op) == 0
    rng = np.random.default_rng(8)
    for x in (np.ones(n), rng.standard_normal(n)):
        assert np.array_equal(op * x, A.matvec(x))
    blocked(op)
    assert fmt_info(op)["fmt"] == 0 and colblocks(op) == 3
    rng = np.random.default_rng(8)
    for x in (np.ones(n), rng.standard_normal(n), 1e180 * rng.standard_normal(n)):
        assert np.array_equal(op * x, A.matvec(x))
    # the row program of a composed operator runs once, on the finished sums
    x = rng.standard_normal(n)
    shifted = 2.0 * op - 0.5 * IdentityOperator(n)
    assert np.array_equal(shifted * x, 2.0 * A.matvec(x) - 0.5 * x)
    # transposed operator: its own blocks
    assert np.array_equal(blocked(op.T) * x, A.rmatvec(x))
    assert colblocks(op.T) == 3
    op.free()


@pytest.mark.parametrize("how", ["resident", "blocked"])
@pytest.mark.parametrize("solver", ["bicgstab", "cgs", "tfqmr"])
def test_column_blocked_solvers_bit_exact(big_random, solver, how):
    """Fused epilogues and gates with the column phases inside one launch (format 3, the default for this matrix) and
    across the block launches: same bits as the oracle in the device's dot order."""
    import pykrylov_amd
    from pykrylov_amd import CsrOperator
    from oracle import gpu_order, krylov_ref as kr
    A = big_random
    n = A.shape[0]
    op = CsrOperator(A.indptr, A.indices, A.data, A.shape)
    if how == "blocked":
        blocked(op)
    assert fmt_info(op)["fmt"] == (3 if how == "resident" else 0)
    rhs = A.matvec(np.ones(n))
    cls = {"bicgstab": pykrylov_amd.BiCGSTAB, "cgs": pykrylov_amd.CGS, "tfqmr": pykrylov_amd.TFQMR}[solver]
    s = cls(op, reltol=1e-9)
    s.solve(rhs, guess=np.linspace(0.5, 1.5, n))
    geo = gpu_order.launch_geometry(op)
    ref = getattr(kr, solver)(A, rhs, reltol=1e-9, guess=np.linspace(0.5, 1.5, n),
                              red=kr.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES[solver], geo)))
    assert s.nMatvec == ref["nMatvec"] and s.converged == ref["converged"]
    assert s.residNorm == ref["residNorm"] and np.array_equal(s.x, ref["x"])
    op.free()


def test_column_blocked_minres_scaled_gather(big_random, monkeypatch):
    """MINRES multiplies the gathered entries by 1/beta on the fly (epilogue `xin`): every block launch must do it."""
    from pykrylov_amd import CsrOperator, Minres
    from oracle import gpu_order, krylov_ref as kr
    monkeypatch.setattr(kr, "_sq", lambda a: a * a)
    R = big_random
    n = R.shape[0]
    rows = np.repeat(np.arange(n), np.diff(R.indptr))
    S = csr_ref.from_coo(np.concatenate([rows, R.indices]), np.concatenate([R.indices, rows]),
                         np.concatenate([R.data, R.data]), (n, n))                    # R + R^T: symmetric, scattered
    op = CsrOperator(S.indptr, S.indices, S.data, S.shape, symmetric=True)
    rhs = S.matvec(np.ones(n))
    for how in ("resident", "blocked"):
        if how == "blocked":
            blocked(op)
            assert colblocks(op) == 3
        else:
            assert fmt_info(op)["fmt"] == 3
        s = Minres(op)
        s.solve(rhs, show=False, check=False, etol=0.0, rtol=1e-12, itnlim=25)
        geo = gpu_order.launch_geometry(op)
        ref = kr.minres(S, rhs, check=False, etol=0.0, rtol=1e-12, itnlim=25,
                        red=kr.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES["minres"], geo)))
        assert (s.istop, s.itn) == (ref["istop"], ref["itn"]), how
        assert np.array_equal(np.array(s.residHistory), ref["residHistory"]) and np.array_equal(s.x, ref["x"]), how
    op.free()


def test_transpose_with_a_dense_column():
    """a dense column used to put an O(len^2) insertion sort on one lane; such matrices are transposed by a
    stable counting sort on the host.  Result: rows of A^T sorted by original row, bit-identical to the oracle."""
    from pykrylov_amd import CsrOperator
    rng = np.random.default_rng(2)
    m, k = 30000, 400
    rows = np.concatenate([np.arange(m), rng.integers(0, m, 60000)])
    cols = np.concatenate([np.full(m, 7), rng.integers(0, k, 60000)])
    A = csr_ref.from_coo(rows, cols, rng.standard_normal(len(rows)), (m, k))
    op = CsrOperator(A.indptr, A.indices, A.data, A.shape)
    At = A.transpose()
    tp, ti, td = op.T.to_csr_arrays()
    assert np.array_equal(tp, At.indptr) and np.array_equal(ti, At.indices) and np.array_equal(td, At.data)
    u = rng.standard_normal(m)
    assert np.array_equal(op.T * u, A.rmatvec(u))
    op.free()


def test_resident_tile_format_under_every_kind_of_launch():
    """Format 3 forced on small matrices, reached through every launch path: composed operators (row program),
    sums / products of two device matrices (two launches, wrapped epilogues), the fused epilogues and gates of a
    solver, the transposed matrix of a least-squares solve."""
    import pykrylov_amd
    from pykrylov_amd import IdentityOperator, lls
    from oracle import gpu_order, krylov_ref as kr, lls_ref
    A = csr_ref.random_diagdom(5003, seed=11)
    B = csr_ref.random_diagdom(5003, seed=12)
    n = A.shape[0]
    opA, opB = op_with_format(A, 3), op_with_format(B, 3)
    assert fmt_info(opA)["fmt"] == 3 and fmt_info(opB)["fmt"] == 3
    x = np.random.default_rng(5).standard_normal(n)
    assert np.array_equal((2.0 * opA - 0.5 * IdentityOperator(n)) * x, 2.0 * A.matvec(x) - 0.5 * x)
    assert np.array_equal((opA + opB) * x, A.matvec(x) + B.matvec(x))
    assert np.array_equal((opA - opB) * x, A.matvec(x) - B.matvec(x))
    assert np.array_equal((opA * opB) * x, A.matvec(B.matvec(x)))
    rhs = A.matvec(np.ones(n))
    for solver in ("bicgstab", "cgs", "tfqmr"):
        cls = {"bicgstab": pykrylov_amd.BiCGSTAB, "cgs": pykrylov_amd.CGS, "tfqmr": pykrylov_amd.TFQMR}[solver]
        s = cls(opA, reltol=1e-10)
        s.solve(rhs)
        ref = getattr(kr, solver)(A, rhs, reltol=1e-10, red=kr.Reductions(
            gpu_order.GpuDots(n, gpu_order.SPMV_SITES[solver], gpu_order.launch_geometry(opA))))
        assert s.nMatvec == ref["nMatvec"] and np.array_equal(s.x, ref["x"]), solver
    # a solver on the sum of the two (second launch carries the fused epilogue)
    S = opA + opB
    s = pykrylov_amd.BiCGSTAB(S, reltol=1e-10)
    s.solve(rhs)
    assert s.converged and np.linalg.norm(A.matvec(s.x) + B.matvec(s.x) - rhs) <= 1e-8 * np.linalg.norm(rhs)
    # least squares: A and its transpose both in format 3
    R = csr_ref.from_coo(*[np.concatenate(c) for c in zip(
        (np.arange(3000), np.arange(3000), np.ones(3000)),
        (np.random.default_rng(1).integers(0, 4000, 9000), np.random.default_rng(2).integers(0, 3000, 9000),
         np.random.default_rng(3).standard_normal(9000)))], (4000, 3000))
    opR = op_with_format(R, 3)
    from pykrylov_amd import _lib
    _lib.check(_lib.init().mk_csr_set_format(opR.T.handle, 3))
    assert fmt_info(opR)["fmt"] == 3 and fmt_info(opR.T)["fmt"] == 3
    b = R.matvec(np.ones(3000))
    s = lls.LSQRFramework(opR)
    s.solve(b, etol=0.0)
    ref = lls_ref.lsqr(R.matvec, R.transpose().matvec, R.shape, b.copy(), etol=0.0)
    assert abs(s.itn - ref["itn"]) <= 1 and np.linalg.norm(s.x - ref["x"]) <= 1e-8 * np.linalg.norm(ref["x"])
    for o in (opA, opB, opR):
        o.free()


def test_pattern_format_rows_without_a_diagonal_entry(monkeypatch):
    """MINRES' epilogue takes s * y[i] from the product (row_x hook): in the pattern format that is the diagonal entry's
    LDS slot -- rows WITHOUT a diagonal entry (here: the adjacency matrix of a path graph plus a few diagonal entries)
    must fall back to loading it.  Bit equality with the oracle in the device's summation order."""
    from pykrylov_amd import CsrOperator, Minres
    from oracle import gpu_order, krylov_ref as kr
    monkeypatch.setattr(kr, "_sq", lambda a: a * a)
    n = 6001
    i = np.arange(n - 1)
    dg = np.arange(0, n, 7)
    A = csr_ref.from_coo(np.concatenate([i, i + 1, dg]), np.concatenate([i + 1, i, dg]),
                         np.concatenate([np.ones(n - 1), np.ones(n - 1), np.full(len(dg), 3.0)]), (n, n))
    op = CsrOperator(A.indptr, A.indices, A.data, A.shape, symmetric=True)
    assert fmt_info(op)["fmt"] == 4
    rhs = A.matvec(np.linspace(1.0, 2.0, n))
    s = Minres(op)
    s.solve(rhs, show=False, check=False, etol=0.0, rtol=1e-12, itnlim=60)
    ref = kr.minres(A, rhs, check=False, etol=0.0, rtol=1e-12, itnlim=60,
                    red=kr.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES["minres"], gpu_order.launch_geometry(op))))
    assert (s.istop, s.itn) == (ref["istop"], ref["itn"]) and s.itn == 60
    assert np.array_equal(np.array(s.residHistory), ref["residHistory"]) and np.array_equal(s.x, ref["x"])
    op.free()


def test_pattern_format_fuzz_over_grid_shapes():
    """Stencil matrices on grids whose lines do not divide the 256-row tiles in every possible way (tile boundaries
    inside lines and planes, ragged last tiles, tiny grids): whatever format the builder settles on, products and
    transposed products keep the scalar loop's bits."""
    from pykrylov_amd import CsrOperator
    rng = np.random.default_rng(12)
    shapes = [(int(a), int(b), int(c)) for a, b, c in zip(rng.integers(3, 70, 14), rng.integers(1, 40, 14), rng.integers(1, 30, 14))]
    shapes += [(256, 3, 2), (255, 2, 2), (257, 2, 1), (512, 2, 2), (1, 1, 300), (2, 129, 5)]
    seen = set()
    for mx, my, mz in shapes:
        A = csr_ref.poisson3d(mx, my, mz)
        op = CsrOperator(A.indptr, A.indices, A.data, A.shape)
        seen.add(fmt_info(op)["fmt"])
        x = rng.standard_normal(A.shape[1])
        assert np.array_equal(op * x, A.matvec(x)), (mx, my, mz, fmt_info(op))
        assert np.array_equal(op.T * x, A.rmatvec(x)), (mx, my, mz)
        op.free()
    assert 4 in seen
