"""Tile orders of the SpMV launches (mk_csr_set_tile_order): the order changes which workgroup visits which tile, never
a row sum -- products are bit-identical in every order, and the per-workgroup partial sums of a fused dot follow the
order, which the oracle mirrors (oracle/gpu_order.py) so that whole solves stay bit-identical too."""
import ctypes

import numpy as np
import pytest

from oracle import csr_ref, gpu_order, krylov_ref as kr
from test_gpu_formats import fmt_info, op_with_format

pytestmark = pytest.mark.gpu


def set_order(op, order, stripe=0, plane=0, nt=-1):
    from pykrylov_amd import _lib
    _lib.check(_lib.init().mk_csr_set_tile_order(op.handle, order, stripe, plane, nt))


def get_order(op):
    from pykrylov_amd import _lib
    o, s, p, nt = (ctypes.c_int32() for _ in range(4))
    _lib.check(_lib.init().mk_csr_tile_order(op.handle, ctypes.byref(o), ctypes.byref(s), ctypes.byref(p), ctypes.byref(nt)))
    return o.value, s.value, p.value, nt.value


# 64 x 64 x 12 grid: 192 tiles, 16 per plane -> order 4 with strips of 2 tiles (16 % (8 * 2) == 0)
ORDERS = [(0, 0, 0), (1, 0, 0), (2, 0, 0), (3, 4, 0), (3, 1, 0), (4, 2, 16), (4, 1, 16)]


@pytest.mark.parametrize("fmt", [0, 1, 5])
@pytest.mark.parametrize("order", ORDERS)
def test_products_and_cg_in_every_tile_order(fmt, order):
    from pykrylov_amd import CG
    A = csr_ref.poisson3d_varcoef(64, 64, 12)
    n = A.shape[0]
    op = op_with_format(A, fmt)
    set_order(op, *order, nt=1 if fmt == 5 else -1)
    assert fmt_info(op)["fmt"] == fmt
    got = get_order(op)
    assert got[0] == order[0] and (order[0] < 3 or got[1] == order[1]) and (order[0] < 4 or got[2] == order[2])
    rng = np.random.default_rng(2)
    x = rng.standard_normal(n)
    assert np.array_equal(op * x, A.matvec(x))
    rhs = A.matvec(np.ones(n))
    s = CG(op)
    s.solve(rhs, matvec_max=60)
    geo = gpu_order.launch_geometry(op)
    assert geo[0] % 8 == 0
    ref = kr.cg(A, rhs, matvec_max=60, red=kr.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES["cg"], geo)))
    assert s.nMatvec == ref["nMatvec"]
    assert np.array_equal(np.array(s.residHistory), ref["residHistory"]) and np.array_equal(s.x, ref["x"])
    op.free()


def test_order4_falls_back_when_planes_do_not_divide():
    A = csr_ref.poisson3d_varcoef(64, 64, 12)
    op = op_with_format(A, 5)
    set_order(op, 4, 2, 17)                                  # 192 tiles are not a multiple of 17
    assert get_order(op)[0] == 2
    set_order(op, 4, 4, 16)                                  # 16 tiles per plane are not 8 strips of 4
    assert get_order(op)[0] == 2
    x = np.ones(A.shape[1])
    assert np.array_equal(op * x, A.matvec(x))
    set_order(op, -1)
    op.free()


def test_ragged_tile_counts_in_striped_orders():
    """Stripes that do not divide the tile count, a last partial tile, fewer tiles than XCDs."""
    from pykrylov_amd import CsrOperator
    rng = np.random.default_rng(5)
    for shape, stripe in (((33, 31, 7), 4), ((50, 50, 3), 8), ((40, 5, 3), 2), ((9, 9, 9), 1), ((100, 77, 2), 16)):
        A = csr_ref.poisson3d_varcoef(*shape)
        op = CsrOperator(A.indptr, A.indices, A.data, A.shape)
        set_order(op, 3, stripe)
        x = rng.standard_normal(A.shape[1])
        assert np.array_equal(op * x, A.matvec(x)), (shape, stripe)
        op.free()
