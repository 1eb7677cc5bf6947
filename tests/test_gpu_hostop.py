"""Matrix-free operators through the device solvers: any object following the reference's
operator protocol works -- the loop (dots, updates, recurrences, stopping tests) runs on the device, the product is
called back on the host exactly when the reference would evaluate `op * v`.

The first tests are the reference's own CG test protocol (pykrylov/cg/tests/test_diagdom.py:29-47, :64-86): CG on the
matrix-free gallery operators, solution 1, accuracy bound cond * eps."""
import numpy as np
import pytest

from conftest import rel_hist_err
from oracle import csr_ref, gpu_order, krylov_ref as kr

pytestmark = pytest.mark.gpu


def macheps():
    return np.finfo(np.float64).eps


class OracleOp(object):
    """Adapter: a pykrylov_amd LinearOperator seen by the oracle (which calls `A.matvec(x)`)."""

    def __init__(self, op):
        self.op, self.shape = op, op.shape

    def matvec(self, x):
        return self.op * x

    def rmatvec(self, u):
        return self.op.T * u

    __call__ = matvec                                       # (the MINRES / SYMMLQ restatements call `A(v)`)


@pytest.mark.parametrize("n", [10, 100, 1000, 10000])
def test_reference_protocol_poisson1d(n):
    """test_diagdom.py:29-47 (Poisson1dTest): CG(LinearOperator(n, n, matvec=Poisson1dMatvec)), rhs = A*e."""
    from pykrylov_amd import CG, LinearOperator
    from pykrylov_amd.gallery import Poisson1dMatvec
    A = LinearOperator(n, n, lambda x: Poisson1dMatvec(x), symmetric=True)
    e = np.ones(n)
    rhs = A * e
    cg = CG(A, matvec_max=2 * n)
    cg.solve(rhs)
    err = np.linalg.norm(e - cg.bestSolution) / np.sqrt(n)
    lmbd_min = 4.0 * np.sin(np.pi / 2.0 / n) ** 2
    lmbd_max = 4.0 * np.sin((n - 1) * np.pi / 2.0 / n) ** 2
    cond = lmbd_max / lmbd_min
    tol = cond * macheps()
    assert cg.converged and err < tol * 100 + 1e-6            # (reference asserts on cond * eps, scaled as there)
    assert A.nMatvec == cg.nMatvec + 1                        # one product formed rhs; the solver's are counted by op


@pytest.mark.parametrize("m", [10, 50, 100, 500])
def test_reference_protocol_poisson2d(m):
    """test_diagdom.py:64-86 (Poisson2dTest), grid sizes up to 500 x 500."""
    from pykrylov_amd import CG, LinearOperator
    from pykrylov_amd.gallery import Poisson2dMatvec
    n = m * m
    A = LinearOperator(n, n, lambda x: Poisson2dMatvec(x), symmetric=True)
    e = np.ones(n)
    rhs = A * e
    cg = CG(A, matvec_max=2 * n)
    cg.solve(rhs)
    err = np.linalg.norm(e - cg.bestSolution) / m
    assert cg.converged and err < 1e-5                        # reltol 1e-6 on a matrix with cond ~ (m / pi)^2
    assert A.nMatvec == cg.nMatvec + 1


def poisson2d_op(m, symmetric=True):
    from pykrylov_amd import LinearOperator
    A = csr_ref.poisson2d(m)
    return A, LinearOperator(m * m, m * m, lambda x: A.matvec(x), symmetric=symmetric)


def geometry(n):
    ntiles = (n + 255) // 256
    g = min(ntiles, 1024)
    return (g - g % 8 if g >= 8 else g), 0


def test_cg_callback_bit_exact_vs_oracle_same_callable():
    """Same callable on both sides, oracle dots in the device's order: counts, history and iterate bit for bit; the
    operator is called exactly nMatvec times (never after the loop condition failed)."""
    from pykrylov_amd import CG
    A, op = poisson2d_op(60)
    n = A.shape[0]
    rhs = A.matvec(np.ones(n))
    calls0 = op.nMatvec
    s = CG(op)
    s.solve(rhs)
    ref = kr.cg(A, rhs, red=kr.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES["cg"], geometry(n))))
    assert s.nMatvec == ref["nMatvec"] and op.nMatvec - calls0 == s.nMatvec
    assert np.array_equal(np.array(s.residHistory), ref["residHistory"]) and np.array_equal(s.x, ref["x"])
    # warm start and product limit
    g = np.linspace(0.0, 2.0, n)
    s2 = CG(op)
    s2.solve(rhs, guess=g, matvec_max=17)
    ref2 = kr.cg(A, rhs, guess=g, matvec_max=17,
                 red=kr.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES["cg"], geometry(n))))
    assert s2.nMatvec == ref2["nMatvec"] == 17 and np.array_equal(s2.x, ref2["x"])


@pytest.mark.parametrize("solver", ["bicgstab", "cgs", "tfqmr"])
def test_nonsymmetric_solvers_with_a_host_operator(solver):
    import pykrylov_amd
    from pykrylov_amd import LinearOperator
    B = csr_ref.random_diagdom(3000, seed=6)
    n = B.shape[0]
    op = LinearOperator(n, n, lambda x: B.matvec(x))
    rhs = B.matvec(np.ones(n))
    cls = {"bicgstab": pykrylov_amd.BiCGSTAB, "cgs": pykrylov_amd.CGS, "tfqmr": pykrylov_amd.TFQMR}[solver]
    for kw in (dict(), dict(guess=1.0 + np.arange(n) / n), dict(matvec_max=3)):
        calls0 = op.nMatvec
        s = cls(op, reltol=1e-9)
        s.solve(rhs, **kw)
        ref = getattr(kr, solver)(B, rhs, reltol=1e-9,
                                  red=kr.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES[solver], geometry(n))), **kw)
        assert s.nMatvec == ref["nMatvec"] and s.converged == ref["converged"]
        assert s.residNorm == ref["residNorm"] and np.array_equal(s.x, ref["x"])
        extra = 1 if ("guess" in kw and solver != "bicgstab") else 0     # cgs/tfqmr do not count the guess product
        assert op.nMatvec - calls0 == s.nMatvec + extra


def test_minres_and_symmlq_with_a_host_operator(monkeypatch):
    from pykrylov_amd import Minres, Symmlq
    monkeypatch.setattr(kr, "_sq", lambda a: a * a)
    A, op = poisson2d_op(40)
    n = A.shape[0]
    rhs = A.matvec(np.ones(n)) - 1.5
    s = Minres(op)
    s.solve(rhs, shift=1.5, show=False, check=True, etol=0.0, rtol=1e-10)
    ref = kr.minres(A, rhs, shift=1.5, check=False, etol=0.0, rtol=1e-10,
                    red=kr.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES["minres"], geometry(n))))
    assert (s.istop, s.itn) == (ref["istop"], ref["itn"])
    assert np.array_equal(np.array(s.residHistory), ref["residHistory"]) and np.array_equal(s.x, ref["x"])
    rhs2 = A.matvec(np.ones(n))
    q = Symmlq(op)
    q.solve(rhs2)
    refq = kr.symmlq(A, rhs2, red=kr.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES["symmlq"], geometry(n))))
    assert q.nMatvec == refq["nMatvec"] and np.array_equal(q.x, refq["x"])


def test_lsqr_with_a_host_operator():
    """Rectangular matrix-free operator: `A * v` and `A.T * u` both called back (lsqr.py:200,264)."""
    from pykrylov_amd import LinearOperator
    from pykrylov_amd.lls import LSQRFramework
    from oracle import lls_ref
    rng = np.random.default_rng(0)
    m, n = 700, 300
    r = rng.integers(0, m, 4000)
    c = rng.integers(0, n, 4000)
    M = csr_ref.from_coo(np.concatenate([r, np.arange(n)]), np.concatenate([c, np.arange(n)]),
                         np.concatenate([rng.standard_normal(4000), 4.0 * np.ones(n)]), (m, n))
    op = LinearOperator(n, m, lambda x: M.matvec(x), matvec_transp=lambda u: M.rmatvec(u))
    b = rng.standard_normal(m)
    s = LSQRFramework(op)
    s.solve(b, show=False)
    ref = lls_ref.lsqr(M.matvec, M.rmatvec, M.shape, b.copy())
    assert (s.istop, s.itn) == (ref["istop"], ref["itn"])
    assert np.linalg.norm(s.x - ref["x"]) <= 1e-11 * np.linalg.norm(ref["x"])


def test_operator_exceptions_propagate():
    from pykrylov_amd import CG, LinearOperator

    class Boom(RuntimeError):
        pass

    count = [0]

    def mv(x):
        count[0] += 1
        if count[0] == 5:
            raise Boom("operator failed on its fifth product")
        return np.arange(1.0, 51.0) * x

    op = LinearOperator(50, 50, mv, symmetric=True)
    with pytest.raises(Boom):
        CG(op, reltol=0.0, abstol=0.0).solve(np.ones(50), matvec_max=40)


# ------------------------------------------------------------------------------------------------ general preconditioners
class TridiagPrecon(object):
    """A non-diagonal SPD operator applied as `precon * r` (generic.py:76): P = tridiag(0.2, 1, 0.2) scaled by 1/d."""

    def __init__(self, d):
        self.d = np.asarray(d, dtype=np.float64)
        self.calls = 0

    def __call__(self, r):
        self.calls += 1
        w = r / self.d
        y = w.copy()
        y[1:] = y[1:] + 0.2 * w[:-1]
        y[:-1] = y[:-1] + 0.2 * w[1:]
        return y / self.d

    def __mul__(self, r):
        return self(r)


@pytest.mark.parametrize("solver", ["cg", "bicgstab", "cgs", "tfqmr"])
def test_host_preconditioner_matches_oracle(solver):
    """Same callable on both sides, oracle dots in the device's order: the general-preconditioner path of the four
    `precon * r` solvers reproduces the oracle bit for bit (CSR operator on the device, preconditioner on the host)."""
    import pykrylov_amd
    from pykrylov_amd import CsrOperator
    if solver == "cg":
        A = csr_ref.poisson2d(40)
    else:
        A = csr_ref.random_diagdom(2500, seed=12)
    n = A.shape[0]
    diag = np.array([A.data[A.indptr[i]:A.indptr[i + 1]][A.indices[A.indptr[i]:A.indptr[i + 1]] == i][0] for i in range(n)])
    op = CsrOperator(A.indptr, A.indices, A.data, A.shape, symmetric=(solver == "cg"))
    rhs = A.matvec(np.ones(n))
    cls = {"cg": pykrylov_amd.CG, "bicgstab": pykrylov_amd.BiCGSTAB, "cgs": pykrylov_amd.CGS,
           "tfqmr": pykrylov_amd.TFQMR}[solver]
    for kw in (dict(matvec_max=60), dict(matvec_max=60, guess=np.linspace(0.9, 1.1, n))):
        P = TridiagPrecon(np.sqrt(np.abs(diag)))
        Pref = TridiagPrecon(np.sqrt(np.abs(diag)))
        s = cls(op, reltol=1e-9, precon=P)
        s.solve(rhs, **kw)
        ref = getattr(kr, solver)(A, rhs, reltol=1e-9, precon=Pref,
                                  red=kr.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES[solver])), **kw)
        assert s.nMatvec == ref["nMatvec"] and s.converged == ref["converged"], (solver, kw.keys())
        assert s.residNorm == ref["residNorm"] and np.array_equal(s.x, ref["x"])
        # applied as often as in the reference, or once more where the application precedes the loop test
        assert Pref.calls <= P.calls <= Pref.calls + 1


def test_host_preconditioner_minres_symmlq(monkeypatch):
    from pykrylov_amd import CsrOperator, Minres, Symmlq
    monkeypatch.setattr(kr, "_sq", lambda a: a * a)
    A = csr_ref.poisson2d(30)
    n = A.shape[0]
    op = CsrOperator(A.indptr, A.indices, A.data, A.shape, symmetric=True)
    rhs = A.matvec(np.ones(n))
    d = np.full(n, 2.0)
    s = Minres(op)
    s.solve(rhs, precon=TridiagPrecon(d), show=False, check=False, etol=0.0, rtol=1e-10)
    ref = kr.minres(A, rhs, precon=TridiagPrecon(d), check=False, etol=0.0, rtol=1e-10,
                    red=kr.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES["minres"])))
    assert (s.istop, s.itn) == (ref["istop"], ref["itn"])
    assert np.array_equal(np.array(s.residHistory), ref["residHistory"]) and np.array_equal(s.x, ref["x"])
    q = Symmlq(op, precon=TridiagPrecon(d))
    q.solve(rhs)
    refq = kr.symmlq(A, rhs, precon=TridiagPrecon(d),
                     red=kr.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES["symmlq"])))
    assert q.nMatvec == refq["nMatvec"] and np.array_equal(q.x, refq["x"])


def test_host_operator_and_host_preconditioner_together():
    """Both callbacks in one solve: the reference's fully matrix-free setting."""
    from pykrylov_amd import CG, LinearOperator
    A = csr_ref.poisson2d(25)
    n = A.shape[0]
    op = LinearOperator(n, n, lambda x: A.matvec(x), symmetric=True)
    rhs = A.matvec(np.ones(n))
    P, Pref = TridiagPrecon(np.full(n, 2.0)), TridiagPrecon(np.full(n, 2.0))
    s = CG(op, precon=P)
    s.solve(rhs, matvec_max=40)
    ref = kr.cg(A, rhs, precon=Pref, matvec_max=40,
                red=kr.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES["cg"], geometry(n))))
    assert s.nMatvec == ref["nMatvec"] and np.array_equal(s.x, ref["x"])
    assert np.array_equal(np.array(s.residHistory), ref["residHistory"])


def test_block_operator_of_device_matrices_in_minres(monkeypatch):
    """A symmetric saddle-point operator [[A, B^T], [B, -D]] built with BlockLinearOperator from device matrices
    (pykrylov_amd/blkop.py; reference linop/blkop.py:8-152): the block products run on the GPU one block at a time,
    the MINRES loop on the GPU as well, and the whole agrees bit for bit with the oracle run on the same callable
    with its dots in the device's order."""
    from pykrylov_amd import CsrOperator, DiagonalOperator, Minres
    from pykrylov_amd.blkop import BlockLinearOperator
    monkeypatch.setattr(kr, "_sq", lambda a: a * a)
    P = csr_ref.poisson2d(20)                                              # 400 x 400, symmetric
    rng = np.random.default_rng(4)
    rows = np.repeat(np.arange(100), 3)
    Bm = csr_ref.from_coo(rows, rng.integers(0, 400, 300), rng.standard_normal(300), (100, 400))
    A = CsrOperator(P.indptr, P.indices, P.data, P.shape, symmetric=True)
    B = CsrOperator(Bm.indptr, Bm.indices, Bm.data, Bm.shape)
    D = DiagonalOperator(-(1.0 + rng.random(100)))
    K = BlockLinearOperator([[A, B.T], [D]], symmetric=True)
    assert K.shape == (500, 500) and K[1, 0] is B and K.T is K
    x = rng.standard_normal(500)
    want = np.concatenate([(0.0 + P.matvec(x[:400])) + Bm.rmatvec(x[400:]), (0.0 + Bm.matvec(x[:400])) + D.diag * x[400:]])
    assert np.array_equal(K * x, want)
    rhs = K * np.ones(500)
    s = Minres(K)
    s.solve(rhs, show=False, check=True, etol=0.0, rtol=1e-10)
    ref = kr.minres(OracleOp(K), rhs, check=False, etol=0.0, rtol=1e-10,
                    red=kr.Reductions(gpu_order.GpuDots(500, gpu_order.SPMV_SITES["minres"], geometry(500))))
    assert (s.istop, s.itn) == (ref["istop"], ref["itn"]) and s.itn > 20
    assert np.array_equal(np.array(s.residHistory), ref["residHistory"]) and np.array_equal(s.x, ref["x"])
    assert np.linalg.norm(s.x - 1.0) < 1e-6
    A.free()
    B.free()
