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


def test_unsupported_compositions_stay_host_operators(golden):
    from pykrylov_amd import CsrOperator, CG, LinearOperator
    d = golden("composed_ops.npz")
    A = golden_csr(d, "A_")
    a, b = op_from(A, symmetric=True), op_from(A, symmetric=True)
    for op in (a + b, a * b, (2 + 1j) * a):
        assert isinstance(op, LinearOperator) and not isinstance(op, CsrOperator)
    y = (a + b) * d["x"]
    assert np.array_equal(y, A.matvec(d["x"]) + A.matvec(d["x"]))
    # a host composition is an operator like any other: the device loop calls it back at each product site (its
    # matvec in turn runs the two device products through the plumbing path), reference semantics (linop.py:332-354)
    n = A.shape[0]
    s = CG(a + b)
    s.solve((a + b) * np.ones(n))
    assert s.converged and np.allclose(s.x, 1.0, rtol=0, atol=1e-3)
    deep = a
    for _ in range(4):
        deep = 2.0 * deep
    assert isinstance(deep, CsrOperator)
    assert not isinstance(2.0 * deep, CsrOperator)             # more than MK_ROWPROG_MAX steps: host closure


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
