"""Preconditioners that are device operators (mk_solver_set_precon_csr): ``precon * r`` evaluated as a product on the
device at the reference's preconditioner sites (generic/generic.py:76; cg.py:91-92,137-138; bicgstab.py:96-99,120-123;
cgs.py:79-80,88-94; tfqmr.py:109-110,142-143; minres.py:249-251; symmlq.py:188-190,308-310).  The product of a device
matrix has the bits of the scalar CSR loop, so a solve with the preconditioner ON the device must agree bit for bit
with the same solve whose preconditioner is the same matrix applied on the HOST through the callback path."""
import numpy as np
import pytest

from oracle import csr_ref

pytestmark = pytest.mark.gpu


class HostMatrixPrecon(object):
    """The same preconditioner as a host object: ``precon * r`` = the oracle's CSR product.  Counts its calls."""

    def __init__(self, M):
        self.M, self.calls = M, 0

    def __mul__(self, r):
        self.calls += 1
        return self.M.matvec(r)


def problem(n_side=20, nonsym=False, seed=3):
    from pykrylov_amd import CsrOperator, tools
    rng = np.random.default_rng(seed)
    A = csr_ref.poisson3d_varcoef(n_side, n_side, 5, seed=seed)
    if nonsym:
        rows = np.repeat(np.arange(A.shape[0]), np.diff(A.indptr))
        skew = np.where(A.indices > rows, 1.25, 1.0)                         # upper triangle scaled: nonsymmetric
        A = csr_ref.RefCsr(A.indptr, A.indices, A.data * skew, A.shape)
    op = CsrOperator(A.indptr, A.indices, A.data, A.shape, symmetric=not nonsym)
    Mop = tools.block_jacobi(op, 4)
    Mref = csr_ref.RefCsr(*Mop.to_csr_arrays(), shape=Mop.shape)
    rhs = A.matvec(1.0 + rng.random(A.shape[0]))
    return A, op, Mop, Mref, rhs


def test_block_jacobi_is_the_inverse_of_the_diagonal_blocks():
    A, op, Mop, Mref, _ = problem(8)
    D, Mi = A.to_dense(), Mref.to_dense()
    n = A.shape[0]
    for b0 in range(0, n, 4):
        blk = D[b0:b0 + 4, b0:b0 + 4]
        assert np.allclose(Mi[b0:b0 + 4, b0:b0 + 4] @ blk, np.eye(len(blk)), atol=1e-12)
    assert Mref.nnz <= 4 * n and np.array_equal(Mi, Mi * (np.abs(np.subtract.outer(np.arange(n) // 4, np.arange(n) // 4)) == 0))
    op.free()
    Mop.free()


@pytest.mark.parametrize("solver", ["cg", "bicgstab", "cgs", "tfqmr", "minres", "symmlq"])
def test_device_preconditioner_matches_the_host_callback_path_bit_for_bit(solver):
    import pykrylov_amd
    from pykrylov_amd.generic import DevicePrecon
    nonsym = solver in ("bicgstab", "cgs", "tfqmr")
    A, op, Mop, Mref, rhs = problem(nonsym=nonsym)
    cls = dict(cg=pykrylov_amd.CG, bicgstab=pykrylov_amd.BiCGSTAB, cgs=pykrylov_amd.CGS, tfqmr=pykrylov_amd.TFQMR,
               minres=pykrylov_amd.Minres, symmlq=pykrylov_amd.Symmlq)[solver]
    host = HostMatrixPrecon(Mref)
    runs = []
    for precon in (Mop, host):
        if solver == "minres":
            s = cls(op)
            assert isinstance(s._device_precon(precon), DevicePrecon) == (precon is Mop)
            s.solve(rhs, precon=precon, show=False, check=False, etol=0.0, rtol=1e-10)
            runs.append((s.itn, s.istop, np.array(s.residHistory), s.x))
        elif solver == "symmlq":
            s = cls(op, precon=precon)
            s.solve(rhs, rtol=1e-10)
            runs.append((s.nMatvec, 0, np.array([s.residNorm]), s.x))
        else:
            s = cls(op, precon=precon, reltol=1e-10)
            assert isinstance(s._device_precon(precon), DevicePrecon) == (precon is Mop)
            s.solve(rhs, matvec_max=400)
            runs.append((s.nMatvec, int(s.converged), np.array(getattr(s, "residHistory", [s.residNorm])), s.x))
    (k0, c0, h0, x0), (k1, c1, h1, x1) = runs
    assert host.calls > 3                                                    # the host path really went through callbacks
    assert k0 == k1 and c0 == c1 and np.array_equal(h0, h1) and np.array_equal(x0, x1), solver
    if solver != "cg":                                                       # (the reference's preconditioned CG stalls,
        assert np.linalg.norm(A.matvec(x0) - rhs) <= 0.1 * np.linalg.norm(rhs)    # DESIGN.md section 7)
    op.free()
    Mop.free()


def test_block_preconditioner_of_device_matrices_and_wrong_shapes():
    """A BlockDiagonalLinearOperator of device matrices as preconditioner (device view), and shape errors."""
    import pykrylov_amd
    from pykrylov_amd import CsrOperator, tools
    from pykrylov_amd.blkop import BlockDiagonalLinearOperator
    from pykrylov_amd.generic import DevicePrecon
    A, op, Mop, Mref, rhs = problem(nonsym=True)
    n = A.shape[0]
    half = n // 2
    # two independent block-Jacobi preconditioners of the two halves of the diagonal, glued by a block operator
    ip, ix, dv = Mop.to_csr_arrays()
    r = np.repeat(np.arange(n), np.diff(ip))
    top, bot = r < half, r >= half
    M1 = CsrOperator.from_coo(r[top], ix[top], dv[top], (half, half))
    M2 = CsrOperator.from_coo(r[bot] - half, ix[bot] - half, dv[bot], (n - half, n - half))
    K = BlockDiagonalLinearOperator([M1, M2])
    s = pykrylov_amd.BiCGSTAB(op, precon=K, reltol=1e-10)
    assert isinstance(s._device_precon(K), DevicePrecon)
    s.solve(rhs)
    s2 = pykrylov_amd.BiCGSTAB(op, precon=Mop, reltol=1e-10)
    s2.solve(rhs)
    assert s.converged and s.nMatvec == s2.nMatvec and np.array_equal(s.x, s2.x)   # same matrix, same bits
    with pytest.raises(ValueError):
        pykrylov_amd.BiCGSTAB(op, precon=M1).solve(rhs)
    for o in (op, Mop, M1, M2):
        o.free()
