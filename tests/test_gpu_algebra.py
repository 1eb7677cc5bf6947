"""Operator algebra that stays on the device (reference linop.py:307-330, :375-426).

`alpha * A`, `-A`, `A / alpha`, `A +/- D`, `D +/- A` (D an IdentityOperator, a DiagonalOperator or a real scalar
multiple of one) applied to a CsrOperator give a CsrOperator again, whose products equal the reference's composite
closures BIT FOR BIT (fixture composed_ops.npz, produced by the real reference) and which the device solvers accept.
"""
import numpy as np
import pytest

from conftest import rel_hist_err
from oracle import csr_ref, gpu_order, krylov_ref as kr

pytestmark = pytest.mark.gpu


def golden_csr(d, prefix):
    return csr_ref.RefCsr(d[prefix + "indptr"], d[prefix + "indices"], d[prefix + "data"], d[prefix + "shape"])


def op_from(A, **kw):
    from pykrylov_amd import CsrOperator
    return CsrOperator(A.indptr, A.indices, A.data, A.shape, **kw)


def forms(n, dv):
    from pykrylov_amd import IdentityOperator, DiagonalOperator
    I = IdentityOperator(n)
    D = DiagonalOperator(dv)
    return {"Am15I": lambda o: o - 1.5 * I, "ApD": lambda o: o + D, "DmA": lambda o: D - o,
            "2p5A": lambda o: 2.5 * o, "negA": lambda o: -o, "Adiv3": lambda o: o / 3.0,
            "nested": lambda o: 2.0 * (o - 1.5 * I) + 0.25 * D}


@pytest.mark.parametrize("form", ["Am15I", "ApD", "DmA", "2p5A", "negA", "Adiv3", "nested"])
def test_composed_products_match_reference_bits(golden, form):
    from pykrylov_amd import CsrOperator
    d = golden("composed_ops.npz")
    A = golden_csr(d, "A_")
    n = A.shape[0]
    base = op_from(A, symmetric=True)
    op = forms(n, d["dv"])[form](base)
    assert isinstance(op, CsrOperator) and op.shape == (n, n) and op.symmetric and op.T is op
    y = op * d["x"]
    assert np.array_equal(y, d["y_" + form])
    assert op.nMatvec == 1 and base.nMatvec == 1              # counted on the composite and on A, as in the reference
    assert isinstance(y, np.ndarray) and y is not (op * d["x"])


def test_host_compositions_still_solve(golden):
    from pykrylov_amd import CsrOperator, CG, LinearOperator
    d = golden("composed_ops.npz")
    A = golden_csr(d, "A_")
    a = op_from(A, symmetric=True)
    op = (2 + 1j) * a                                          # complex scalar: a host closure, as in the reference
    assert isinstance(op, LinearOperator) and not isinstance(op, CsrOperator)
    deep = a
    for _ in range(4):
        deep = 2.0 * deep
    assert isinstance(deep, CsrOperator)
    deeper = 2.0 * deep                                        # more than MK_ROWPROG_MAX steps: host closure ...
    assert not isinstance(deeper, CsrOperator)
    n = A.shape[0]
    s = CG(deeper)                                             # ... which the device loop calls back at each product
    s.solve(deeper * np.ones(n))
    assert s.converged and np.allclose(s.x, 1.0, rtol=0, atol=1e-3)


def pair_matrices():
    rng = np.random.default_rng(21)
    n = 1500
    def rnd(seed, m=n, k=n, nz=6000):
        r = np.random.default_rng(seed)
        return csr_ref.from_coo(np.concatenate([r.integers(0, m, nz), np.arange(min(m, k))]),
                                np.concatenate([r.integers(0, k, nz), np.arange(min(m, k))]),
                                np.concatenate([r.standard_normal(nz), 5.0 + r.random(min(m, k))]), (m, k))
    return rnd(1), rnd(2), rng


def test_sum_difference_product_are_device_operators_with_reference_bits():
    """(A + B) * x = (A*x) + (B*x), (A - B) * x, (A * B) * x = A * (B * x): the reference's closures
    (linop.py:332-354, :375-426), evaluated on the device with the same bits; products are counted on the composite
    and on both operands."""
    from pykrylov_amd import CsrOperator
    A, B, rng = pair_matrices()
    a, b = op_from(A), op_from(B)
    x = rng.standard_normal(A.shape[1])
    for op, want in ((a + b, A.matvec(x) + B.matvec(x)), (a - b, A.matvec(x) - B.matvec(x)),
                     (a * b, A.matvec(B.matvec(x)))):
        assert isinstance(op, CsrOperator) and op.shape == A.shape
        assert np.array_equal(op * x, want)
    assert a.nMatvec == 3 and b.nMatvec == 3
    # transposes
    u = rng.standard_normal(A.shape[0])
    assert np.array_equal((a + b).T * u, A.rmatvec(u) + B.rmatvec(u))
    assert np.array_equal((a * b).T * u, B.rmatvec(A.rmatvec(u)))
    # scaled operands keep their row programs; a pair of pairs falls back to a host closure that still works
    assert np.array_equal((2.0 * a - 0.5 * b) * x, 2.0 * A.matvec(x) - 0.5 * B.matvec(x))
    nested = (a + b) + a
    assert not isinstance(nested, CsrOperator)
    assert np.array_equal(nested * x, (A.matvec(x) + B.matvec(x)) + A.matvec(x))
    # rectangular product
    C = csr_ref.from_coo(rng.integers(0, 1500, 4000), rng.integers(0, 700, 4000), rng.standard_normal(4000), (1500, 700))
    c = op_from(C)
    z = rng.standard_normal(700)
    ac = a * c
    assert isinstance(ac, CsrOperator) and ac.shape == (1500, 700)
    assert np.array_equal(ac * z, A.matvec(C.matvec(z)))
    with pytest.raises(Exception):
        c * a                                                                # shapes do not chain


def test_solvers_on_device_sums_and_products(monkeypatch):
    """The fused epilogues and gates of the solver kernels around a two-launch product: bit-exact against the oracle
    run on the same composite (dots in the device's order)."""
    from pykrylov_amd import CG, BiCGSTAB, Minres, CsrOperator
    monkeypatch.setattr(kr, "_sq", lambda a: a * a)
    P = csr_ref.poisson2d(35)
    n = P.shape[0]
    dgl = csr_ref.from_coo(np.arange(n), np.arange(n), 0.5 + np.arange(n) / n, (n, n))
    p, dgo = op_from(P, symmetric=True), op_from(dgl, symmetric=True)
    # CG on P + D (SPD)
    op = p + dgo
    assert isinstance(op, CsrOperator) and op.symmetric
    rhs = P.matvec(np.ones(n)) + dgl.matvec(np.ones(n))
    s = CG(op)
    s.solve(rhs)

    class Sum(object):
        shape = P.shape

        def matvec(self, x):
            return P.matvec(x) + dgl.matvec(x)
        __call__ = matvec
    ref = kr.cg(Sum(), rhs, red=kr.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES["cg"], gpu_order.launch_geometry(op))))
    assert s.nMatvec == ref["nMatvec"] and np.array_equal(s.x, ref["x"])
    assert np.array_equal(np.array(s.residHistory), ref["residHistory"])
    assert p.nMatvec == s.nMatvec and dgo.nMatvec == s.nMatvec                # counted on the operands too
    # MINRES on P * P (symmetric positive definite, declared unsymmetric like the reference's product): the scaled
    # gather (epilogue xin) belongs to the INNER product only
    pp = p * p

    class Prod(object):
        shape = P.shape

        def matvec(self, x):
            return P.matvec(P.matvec(x))
        __call__ = matvec
    rhs2 = Prod().matvec(np.ones(n))
    m = Minres(pp)
    m.solve(rhs2, show=False, check=False, etol=0.0, rtol=1e-10, itnlim=60)
    refm = kr.minres(Prod(), rhs2, check=False, etol=0.0, rtol=1e-10, itnlim=60,
                     red=kr.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES["minres"], gpu_order.launch_geometry(pp))))
    assert (m.istop, m.itn) == (refm["istop"], refm["itn"])
    assert np.array_equal(np.array(m.residHistory), refm["residHistory"]) and np.array_equal(m.x, refm["x"])
    # BiCGSTAB on A - B (nonsymmetric), gates in both products
    A, B, _ = pair_matrices()
    a, b = op_from(A), op_from(B)
    amb = a - 0.25 * b

    class Diff(object):
        shape = A.shape

        def matvec(self, x):
            return A.matvec(x) - 0.25 * B.matvec(x)
        __call__ = matvec
    rhs3 = Diff().matvec(np.ones(A.shape[0]))
    q = BiCGSTAB(amb, reltol=1e-9)
    q.solve(rhs3, matvec_max=80)
    refq = kr.bicgstab(Diff(), rhs3, reltol=1e-9, matvec_max=80,
                       red=kr.Reductions(gpu_order.GpuDots(A.shape[0], gpu_order.SPMV_SITES["bicgstab"],
                                                           gpu_order.launch_geometry(amb))))
    assert q.nMatvec == refq["nMatvec"] and q.residNorm == refq["residNorm"] and np.array_equal(q.x, refq["x"])


def test_lsqr_on_a_device_product():
    from pykrylov_amd.lls import LSQRFramework
    from oracle import lls_ref
    rng = np.random.default_rng(31)
    A = csr_ref.from_coo(np.concatenate([rng.integers(0, 90, 500), np.arange(60)]),
                         np.concatenate([rng.integers(0, 60, 500), np.arange(60)]),
                         np.concatenate([rng.standard_normal(500), 3.0 * np.ones(60)]), (90, 60))
    B = csr_ref.from_coo(np.concatenate([rng.integers(0, 60, 300), np.arange(40)]),
                         np.concatenate([rng.integers(0, 40, 300), np.arange(40)]),
                         np.concatenate([rng.standard_normal(300), 2.0 * np.ones(40)]), (60, 40))
    op = op_from(A) * op_from(B)
    b = rng.standard_normal(90)
    s = LSQRFramework(op)
    s.solve(b, show=False)
    ref = lls_ref.lsqr(lambda v: A.matvec(B.matvec(v)), lambda u: B.rmatvec(A.rmatvec(u)), (90, 40), b.copy())
    assert abs(s.itn - ref["itn"]) <= 1 and s.istop == ref["istop"]
    assert np.linalg.norm(s.x - ref["x"]) <= 1e-9 * np.linalg.norm(ref["x"])


def test_rectangular_scaling_and_transpose():
    from pykrylov_amd import CsrOperator
    rng = np.random.default_rng(3)
    A = csr_ref.from_coo(rng.integers(0, 40, 300), rng.integers(0, 25, 300), rng.standard_normal(300), (40, 25))
    op = 0.75 * op_from(A)
    assert isinstance(op, CsrOperator) and op.shape == (40, 25)
    x, u = rng.standard_normal(25), rng.standard_normal(40)
    assert np.array_equal(op * x, 0.75 * A.matvec(x))
    assert np.array_equal(op.T * u, 0.75 * A.transpose().matvec(u))     # linop.py:318-319
    with pytest.raises(Exception):
        op_from(A) + op_from(A).T                                        # shapes differ


def test_cg_on_A_plus_D(golden):
    from pykrylov_amd import CG, DiagonalOperator
    d = golden("composed_ops.npz")
    A = golden_csr(d, "A_")
    n = A.shape[0]
    dv = d["dv"]
    op = op_from(A, symmetric=True) + DiagonalOperator(dv)
    s = CG(op)
    s.solve(d["cg_ApD_rhs"])
    assert s.nMatvec == int(d["cg_ApD_nMatvec"]) and s.converged
    assert rel_hist_err(s.residHistory, d["cg_ApD_residHistory"]) <= 1e-12
    assert np.linalg.norm(s.x - d["cg_ApD_x"]) <= 1e-12 * np.linalg.norm(d["cg_ApD_x"])
    ref = kr.cg(csr_ref.Composed(A, lambda y, x: y + dv * x), d["cg_ApD_rhs"],
                red=kr.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES["cg"])))
    assert ref["nMatvec"] == s.nMatvec
    assert np.array_equal(ref["residHistory"], np.array(s.residHistory)) and np.array_equal(ref["x"], s.x)


@pytest.mark.parametrize("form", ["Am15I", "halfA"])
def test_minres_on_composed(golden, form, monkeypatch):
    from pykrylov_amd import Minres, IdentityOperator
    d = golden("composed_ops.npz")
    A = golden_csr(d, "A_")
    n = A.shape[0]
    base = op_from(A, symmetric=True)
    op = base - 1.5 * IdentityOperator(n) if form == "Am15I" else 0.5 * base
    fn = (lambda y, x: y - 1.5 * x) if form == "Am15I" else (lambda y, x: 0.5 * y)
    k = "minres_%s_" % form
    s = Minres(op)
    s.solve(d[k + "rhs"], show=False, check=False, etol=0.0, rtol=1e-10)
    definite = form == "halfA"
    assert s.istop == int(d[k + "istop"]) and abs(s.itn - int(d[k + "itn"])) <= (0 if definite else 2)
    head = len(d[k + "residHistory"]) if definite else 30
    assert rel_hist_err(s.residHistory[:head], d[k + "residHistory"][:head]) <= 1e-11
    assert np.linalg.norm(s.x - d[k + "x"]) <= (1e-11 if definite else 1e-6) * np.linalg.norm(d[k + "x"])
    assert base.nMatvec == s.itn                                          # the solve is counted on A as well
    monkeypatch.setattr(kr, "_sq", lambda a: a * a)
    ref = kr.minres(csr_ref.Composed(A, fn), d[k + "rhs"], check=False, etol=0.0, rtol=1e-10,
                    red=kr.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES["minres"])))
    assert (s.istop, s.itn) == (ref["istop"], ref["itn"])
    assert np.array_equal(np.array(s.residHistory), ref["residHistory"]) and np.array_equal(s.x, ref["x"])
    if form == "Am15I":
        # the same problem through the solver's own `shift` keyword performs the same arithmetic (minres.py:239-240)
        s2 = Minres(op_from(A, symmetric=True))
        s2.solve(d[k + "rhs"], shift=1.5, show=False, check=False, etol=0.0, rtol=1e-10)
        assert np.array_equal(np.array(s2.residHistory), np.array(s.residHistory)) and np.array_equal(s2.x, s.x)


def test_lsqr_on_scaled_operator():
    from pykrylov_amd.lls import LSQRFramework
    from oracle import lls_ref
    rng = np.random.default_rng(11)
    A = csr_ref.from_coo(np.concatenate([rng.integers(0, 60, 400), np.arange(40)]),
                         np.concatenate([rng.integers(0, 40, 400), np.arange(40)]),
                         np.concatenate([rng.standard_normal(400), 3.0 + np.arange(40.0)]), (60, 40))
    b = A.matvec(np.ones(40)) + 0.1 * rng.standard_normal(60)
    s = LSQRFramework(0.5 * op_from(A))
    s.solve(b, show=False)
    # scaling by a power of two is exact, so the composed operator must behave bit for bit like the matrix with
    # halved entries (both run on the device: same summation order)
    s2 = LSQRFramework(op_from(csr_ref.RefCsr(A.indptr, A.indices, 0.5 * A.data, A.shape)))
    s2.solve(b, show=False)
    assert (s.itn, s.istop) == (s2.itn, s2.istop) and np.array_equal(s.x, s2.x)
    At = A.transpose()
    ref = lls_ref.lsqr(lambda v: 0.5 * A.matvec(v), lambda u: 0.5 * At.matvec(u), (60, 40), b)
    assert abs(s.itn - ref["itn"]) <= 1 and s.istop == ref["istop"]      # stops on a tolerance: +-1 across orders
    assert np.linalg.norm(s.x - ref["x"]) <= 1e-8 * np.linalg.norm(ref["x"])
