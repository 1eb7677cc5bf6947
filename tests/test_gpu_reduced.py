"""Reduced operators of device matrices (mk_csr_create_reduced; reference linop/linop.py:560-623): scatter, product and
gather on the device, the reference's evaluation `z = 0; z[cols] = x; y = (A z)[rows]`, accepted by the device solvers
with no host callback."""
import numpy as np
import pytest

from oracle import csr_ref, gpu_order, krylov_ref as kr

pytestmark = pytest.mark.gpu


def test_reduced_products_transposes_and_fallbacks():
    from pykrylov_amd import CsrOperator, LinearOperator, ReducedLinearOperator, SymmetricallyReducedLinearOperator
    from pykrylov_amd.linop import _ReducedCsrOperator
    rng = np.random.default_rng(1)
    A = csr_ref.random_diagdom(3001, seed=5)
    op = CsrOperator(A.indptr, A.indices, A.data, A.shape)
    rows = rng.choice(3001, 700, replace=True)               # rows may repeat
    cols = rng.choice(3001, 1200, replace=False)
    R = ReducedLinearOperator(op, rows, cols)
    assert isinstance(R, _ReducedCsrOperator) and R.shape == (700, 1200)
    x = rng.standard_normal(1200)
    z = np.zeros(3001)
    z[cols] = x
    assert np.array_equal(R * x, A.matvec(z)[rows])
    assert op.nMatvec == 1 and R.nMatvec == 1                 # counted on both, as the reference's closure does
    u = rng.standard_normal(700)
    zt = np.zeros(3001)
    zt[rows] = u                                              # (repeated rows: NumPy keeps the last -- host path)
    Rt = R.T
    assert not isinstance(Rt, _ReducedCsrOperator)            # repeated indices on the scatter side: host closure
    assert np.array_equal(Rt * u, A.rmatvec(zt)[cols])
    rows_u = rng.choice(3001, 700, replace=False)
    R2 = ReducedLinearOperator(op, rows_u, cols)
    zt = np.zeros(3001)
    zt[rows_u] = u
    assert isinstance(R2.T, _ReducedCsrOperator) and np.array_equal(R2.T * u, A.rmatvec(zt)[cols])
    # a host operator keeps the host closures
    H = ReducedLinearOperator(LinearOperator(3001, 3001, lambda v: 2.0 * v, symmetric=True), rows_u, cols)
    assert not isinstance(H, CsrOperator)
    # symmetric restriction of a symmetric matrix
    P = csr_ref.poisson2d(40)
    p = CsrOperator(P.indptr, P.indices, P.data, P.shape, symmetric=True)
    idx = np.sort(rng.choice(1600, 1000, replace=False))
    S = SymmetricallyReducedLinearOperator(p, idx)
    assert isinstance(S, _ReducedCsrOperator) and S.symmetric and S.T is S
    y = rng.standard_normal(1000)
    zz = np.zeros(1600)
    zz[idx] = y
    assert np.array_equal(S * y, P.matvec(zz)[idx])
    # destroying the base first is deferred
    op.free()
    assert np.array_equal(R * x, A.matvec(z)[rows])
    p.free()


def test_cg_on_a_symmetrically_reduced_device_matrix_bit_exact():
    """The principal submatrix of an SPD matrix is SPD: CG on it runs entirely on the device (no HostOperatorShell) and
    agrees bit for bit with the oracle on the explicit submatrix with its dots in the device's order."""
    from pykrylov_amd import CG, CsrOperator, SymmetricallyReducedLinearOperator
    from pykrylov_amd.linop import HostOperatorShell
    rng = np.random.default_rng(3)
    P = csr_ref.poisson3d_varcoef(14, 13, 12)
    n = P.shape[0]
    p = CsrOperator(P.indptr, P.indices, P.data, P.shape, symmetric=True)
    idx = np.sort(rng.choice(n, 1500, replace=False))
    S = SymmetricallyReducedLinearOperator(p, idx)

    class Sub(object):
        shape = (1500, 1500)

        def matvec(self, x):
            z = np.zeros(n)
            z[idx] = x
            return P.matvec(z)[idx]
        __call__ = matvec
    rhs = Sub().matvec(np.ones(1500))
    s = CG(S)
    assert not isinstance(s._device_operator(), HostOperatorShell)
    s.solve(rhs)
    ref = kr.cg(Sub(), rhs, red=kr.Reductions(gpu_order.GpuDots(1500, gpu_order.SPMV_SITES["cg"], ((1500 + 255) // 256, 0))))
    assert s.converged and s.nMatvec == ref["nMatvec"]
    assert np.array_equal(np.array(s.residHistory), ref["residHistory"]) and np.array_equal(s.x, ref["x"])
    assert p.nMatvec >= s.nMatvec
    p.free()
