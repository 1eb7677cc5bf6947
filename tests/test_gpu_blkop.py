"""Block operators as device composites (mk_csr_create_block; reference linop/blkop.py:8-152, :154-257): a
BlockLinearOperator / BlockDiagonalLinearOperator whose blocks all live on the device is ONE device operator for the
solvers -- one launch per block, block products added to the block row one at a time as the reference does -- with no
host callback anywhere, and with the bits of the host composition."""
import numpy as np
import pytest

from oracle import csr_ref, gpu_order, krylov_ref as kr

pytestmark = pytest.mark.gpu


class OracleOp(object):
    def __init__(self, op):
        self.op, self.shape = op, op.shape

    def matvec(self, x):
        return self.op * x
    __call__ = matvec


def host_grid_product(grid, x):
    "The reference's evaluation order on plain oracle matrices: y_i = ((0 + B_i0 x_0) + B_i1 x_1) + ..."
    widths = [b[1] for b in grid[0]]
    c = np.concatenate([[0], np.cumsum(widths)])
    out = []
    for row in grid:
        acc = np.zeros(row[0][0])
        for j, (h, w, f) in enumerate(row):
            acc = acc + f(x[c[j]:c[j + 1]])
        out.append(acc)
    return np.concatenate(out)


def saddle(rng, na=401, nb=100):
    """[[A, B^T], [B, D]] with an odd first block width: the second block column starts at an odd offset of x."""
    from pykrylov_amd import CsrOperator, DiagonalOperator
    P = csr_ref.from_coo(*[np.concatenate(t) for t in zip(
        (np.arange(na), np.arange(na), 4.0 + rng.random(na)),
        (np.arange(na - 1), np.arange(1, na), -np.ones(na - 1)),
        (np.arange(1, na), np.arange(na - 1), -np.ones(na - 1)))], (na, na))
    rows = np.repeat(np.arange(nb), 3)
    Bm = csr_ref.from_coo(rows, rng.integers(0, na, 3 * nb), rng.standard_normal(3 * nb), (nb, na))
    d = -(1.0 + rng.random(nb))
    A = CsrOperator(P.indptr, P.indices, P.data, P.shape, symmetric=True)
    B = CsrOperator(Bm.indptr, Bm.indices, Bm.data, Bm.shape)
    D = DiagonalOperator(d)
    return P, Bm, d, A, B, D


def test_saddle_point_minres_runs_without_host_callbacks(monkeypatch):
    from pykrylov_amd import Minres
    from pykrylov_amd.blkop import BlockLinearOperator
    from pykrylov_amd.linop import HostOperatorShell, _BlockCsrOperator
    monkeypatch.setattr(kr, "_sq", lambda a: a * a)
    rng = np.random.default_rng(4)
    P, Bm, d, A, B, D = saddle(rng)
    K = BlockLinearOperator([[A, B.T], [D]], symmetric=True)
    n = K.shape[0]
    s = Minres(K)
    dev = s._device_operator()
    assert isinstance(dev, _BlockCsrOperator) and not isinstance(dev, HostOperatorShell) and dev.shape == K.shape
    x = rng.standard_normal(n)
    want = host_grid_product([[(401, 401, P.matvec), (401, 100, Bm.rmatvec)],
                              [(100, 401, Bm.matvec), (100, 100, lambda v: d * v)]], x)
    assert np.array_equal(K * x, want)                       # host composition (blkop.py:86-96)
    assert np.array_equal(dev * x, want)                     # the device composite: same bits
    rhs = K * np.ones(n)
    counts0 = (K.nMatvec, A.nMatvec, B.nMatvec, B.T.nMatvec, D.nMatvec)
    s.solve(rhs, show=False, check=False, etol=0.0, rtol=1e-10)
    g = (n + 255) // 256
    ref = kr.minres(OracleOp(K), rhs, check=False, etol=0.0, rtol=1e-10,
                    red=kr.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES["minres"], (g, 0))))
    assert (s.istop, s.itn) == (ref["istop"], ref["itn"]) and s.itn > 20
    assert np.array_equal(np.array(s.residHistory), ref["residHistory"]) and np.array_equal(s.x, ref["x"])
    # products are counted on the block operator and on every block, once per product of the solve
    k = s.nMatvec
    assert K.nMatvec - counts0[0] >= k and A.nMatvec - counts0[1] >= k and B.nMatvec - counts0[2] >= k
    assert D.nMatvec - counts0[4] >= k                       # ... on the DiagonalOperator block too, not on its device form
    # destroying a block while the composite is alive is deferred: the composite keeps working
    A.free()
    assert np.array_equal(dev * x, want)
    dev.free()
    B.free()


def test_block_diagonal_cg_and_general_grid():
    from pykrylov_amd import CG, CsrOperator, IdentityOperator
    from pykrylov_amd.blkop import BlockDiagonalLinearOperator, BlockLinearOperator
    from pykrylov_amd.linop import _BlockCsrOperator
    rng = np.random.default_rng(9)
    mats = [csr_ref.poisson2d(17), csr_ref.poisson1d(333), csr_ref.poisson3d(6, 5, 4)]
    ops = [CsrOperator(M.indptr, M.indices, M.data, M.shape, symmetric=True) for M in mats]
    Kd = BlockDiagonalLinearOperator(ops + [IdentityOperator(7)])
    n = Kd.shape[0]
    s = CG(Kd)
    dev = s._device_operator()
    assert isinstance(dev, _BlockCsrOperator)
    x = rng.standard_normal(n)
    off = np.concatenate([[0], np.cumsum([M.shape[0] for M in mats] + [7])])
    want = np.concatenate([(np.zeros(M.shape[0]) + M.matvec(x[off[i]:off[i + 1]])) for i, M in enumerate(mats)] + [x[off[3]:]])
    assert np.array_equal(Kd * x, want) and np.array_equal(dev * x, want)
    rhs = Kd * np.ones(n)
    s.solve(rhs)
    ref = kr.cg(OracleOp(Kd), rhs, red=kr.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES["cg"], ((n + 255) // 256, 0))))
    assert s.converged and s.nMatvec == ref["nMatvec"]
    assert np.array_equal(np.array(s.residHistory), ref["residHistory"]) and np.array_equal(s.x, ref["x"])
    # a rectangular 2 x 3 grid, products only
    R = [[csr_ref.from_coo(rng.integers(0, h, 5 * h), rng.integers(0, w, 5 * h), rng.standard_normal(5 * h), (h, w))
          for w in (300, 301, 77)] for h in (513, 200)]
    G = BlockLinearOperator([[CsrOperator(M.indptr, M.indices, M.data, M.shape) for M in row] for row in R])
    gd = G._device_view()
    assert isinstance(gd, _BlockCsrOperator) and gd.shape == (713, 678)
    z = rng.standard_normal(678)
    want = host_grid_product([[(M.shape[0], M.shape[1], M.matvec) for M in row] for row in R], z)
    assert np.array_equal(G * z, want) and np.array_equal(gd * z, want)
    for o in ops:
        o.free()


def test_blocks_without_a_device_form_keep_the_host_composition():
    from pykrylov_amd import CG, CsrOperator, LinearOperator
    from pykrylov_amd.blkop import BlockDiagonalLinearOperator
    from pykrylov_amd.linop import HostOperatorShell
    M = csr_ref.poisson2d(12)
    a = CsrOperator(M.indptr, M.indices, M.data, M.shape, symmetric=True)
    f = LinearOperator(50, 50, lambda v: 3.0 * v, symmetric=True)
    K = BlockDiagonalLinearOperator([a, f])
    assert K._device_view() is None
    s = CG(K)
    assert isinstance(s._device_operator(), HostOperatorShell)
    rhs = K * np.ones(K.shape[0])
    s.solve(rhs)
    assert s.converged and np.linalg.norm(s.x - 1.0) < 1e-5
    a.free()
