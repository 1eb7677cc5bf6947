"""Block operators (pykrylov_amd/blkop.py) against the protocol of the reference's own tests
(pykrylov/linop/tests/test_blkop.py:38-151): shapes, membership, symmetric fill-in, products against the explicit dense
matrix, transposes, item assignment.  Host-only composition: runs without a GPU."""
import numpy as np
import pytest

from pykrylov_amd import linop as lo
from pykrylov_amd import blkop as bo
from pykrylov_amd.linop import ShapeError


@pytest.fixture
def ops():
    A = lo.IdentityOperator(2)
    B = lo.linop_from_ndarray(np.arange(1, 7).reshape([2, 3]))
    C = lo.DiagonalOperator(np.arange(3))
    D = lo.linop_from_ndarray(np.arange(6, 0, -1).reshape([3, 2]))
    return A, B, C, D


def test_block_operator_construction(ops):
    A, B, C, D = ops
    M = bo.BlockLinearOperator([[A, B], [D, C]])
    assert M.shape == (5, 5) and M.blocks == [[A, B], [D, C]]
    for blk, at in ((A, (0, 0)), (B, (0, 1)), (C, (1, 1)), (D, (1, 0))):
        assert blk in M and M[at] is blk
    assert [row for row in M] == [[A, B], [D, C]]
    M = bo.BlockLinearOperator([[A, B], [C]], symmetric=True)        # upper triangle only
    assert M.shape == (5, 5) and B.T in M and M[1, 0] is B.T and M.T is M
    assert bo.BlockLinearOperator([[A, B], [A, B]]).shape == (4, 5)
    with pytest.raises(TypeError):
        bo.BlockLinearOperator([A, C, D, C])                          # one block row must be [[...]]
    with pytest.raises(ShapeError):
        bo.BlockLinearOperator([[A, C], [D, C]])
    with pytest.raises(ShapeError):
        bo.BlockLinearOperator([[A, B], [B, C]])
    with pytest.raises(ValueError):
        bo.BlockLinearOperator([[A, B], [B]], symmetric=True)         # diagonal block not symmetric
    with pytest.raises(ShapeError):
        bo.BlockLinearOperator([[A, B]], symmetric=True)


def test_block_operator_products(ops):
    A, B, C, D = ops
    rng = np.random.default_rng(0)
    M = bo.BlockLinearOperator([[A, B], [D, C]])
    dense = np.array([[1, 0, 1, 2, 3], [0, 1, 4, 5, 6], [6, 5, 0, 0, 0], [4, 3, 0, 1, 0], [2, 1, 0, 0, 2]], dtype=float)
    x = rng.random(5)
    assert np.allclose(M * x, dense @ x) and np.allclose(M.T * x, dense.T @ x) and np.allclose(M.H * x, dense.T @ x)
    assert np.array_equal(M.to_array(), dense)
    # the accumulation order of a block row is the reference's: (0 + A x0) + B x1
    assert np.array_equal((M * x)[:2], (0.0 + A * x[:2]) + B * x[2:])
    S = bo.BlockLinearOperator([[A, B], [C]], symmetric=True)
    sd = np.array([[1, 0, 1, 2, 3], [0, 1, 4, 5, 6], [1, 4, 0, 0, 0], [2, 5, 0, 1, 0], [3, 6, 0, 0, 2]], dtype=float)
    assert np.allclose(S * x, sd @ x)
    R = bo.BlockLinearOperator([[A, B], [A, B]])
    rd = np.vstack([dense[:2], dense[:2]])
    assert np.allclose(R * x, rd @ x)
    y = rng.random(4)
    assert np.allclose(R.T * y, rd.T @ y)
    with pytest.raises(ValueError):                                   # (LinearOperator's own check, linop.py:290-298)
        M * np.ones(4)
    # sub-blocks keep their orientation (the reference indexes an np.matrix of blocks)
    assert M[0:2, 1].shape == (5, 3) and M[0, 0:2].shape == (2, 5) and M[1].shape == (3, 5)
    # item assignment refreshes the transposed grid
    M[0, 0] = 2 * A
    dense[:2, :2] *= 2
    assert np.allclose(M * x, dense @ x) and np.allclose(M.T * x, dense.T @ x)
    S[0, 1] = 2 * B                                                   # the mirrored block follows
    sd[:2, 2:] *= 2
    sd[2:, :2] *= 2
    assert np.allclose(S * x, sd @ x)


def test_block_diagonal_operator(ops):
    A, B, C, D = ops
    rng = np.random.default_rng(1)
    M = bo.BlockDiagonalLinearOperator([A, C])
    assert M.shape == (5, 5) and M.symmetric is True and M.blocks == [A, C] and M.T is M
    x = rng.random(5)
    assert np.allclose(M * x, np.concatenate([x[:2], np.arange(3) * x[2:]]))
    N = bo.BlockDiagonalLinearOperator([A, B])
    assert N.shape == (4, 5) and N.symmetric is False
    dense = np.zeros((4, 5))
    dense[:2, :2] = np.eye(2)
    dense[2:, 2:] = np.arange(1, 7).reshape(2, 3)
    y = rng.random(4)
    assert np.allclose(N * x, dense @ x) and np.allclose(N.T * y, dense.T @ y) and np.allclose(N.H * y, dense.T @ y)
    assert N[1] is B and N[0:2].shape == (4, 5)
    N[1] = D.T
    dense[2:, 2:] = np.arange(6, 0, -1).reshape(3, 2).T
    assert np.allclose(N * x, dense @ x) and np.allclose(N.T * y, dense.T @ y)
    with pytest.raises(ValueError):
        N[0] = 3.0
    P = bo.BlockDiagonalPreconditioner([A, C])
    assert np.array_equal(P.solve(x), P * x)
    Q = bo.BlockPreconditioner([[A, B], [D, C]])
    assert np.array_equal(Q.solve(x), Q * x)
