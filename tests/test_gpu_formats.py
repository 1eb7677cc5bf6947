"""The storage formats of the SpMV (csrc/mk_format.hip, mk_device.h): plain CSR gathers (0), windowed tiles (1) and
windowed tiles + value dictionary (2).  Whatever format a tile ends up in, the product must be BIT-identical to the
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


def test_mixed_tiles_really_mix():
    A, _ = MATS["banded_plus_dense_rows"]
    op = op_with_format(A, 1)
    info = fmt_info(op)
    assert 0 < info["tiles"] < (A.shape[0] + 255) // 256            # some tiles windowed, some on the gather path
    op.free()


@pytest.mark.parametrize("fmt", [0, 1, 2])
def test_solver_results_do_not_depend_on_the_format(fmt):
    """CG on a 2-D Poisson matrix: identical bits (history, iterate) in all three formats."""
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
    assert fmt_info(op.T)["fmt"] == 2
    shifted = op - 1.5 * IdentityOperator(A.shape[0])
    assert np.array_equal(shifted * x, A.matvec(x) - 1.5 * x)
    op.free()
