"""GPU parity of the diagonal (Jacobi-type) preconditioner hook (reference call sites
cg.py:91-92,137-138, bicgstab.py:96-99,120-123, cgs.py:79-82,88-91, tfqmr.py:77-80,109-112,142-145,
minres.py:162-163,249, symmlq.py:134,228).

Two kinds of checks per solver: (i) against the fixture the real reference produced
(tests/golden/precon_jacobi.npz) within the 1e-12 tolerance, over the stretch of the run where the
recurrence is not chaotic in the summation order; (ii) BIT equality with the oracle when the oracle's
dots are evaluated in the device's summation order (oracle/gpu_order.py).
"""
import numpy as np
import pytest

from conftest import rel_hist_err
from oracle import csr_ref, gpu_order, krylov_ref as kr

pytestmark = pytest.mark.gpu
TOL = 1e-12


def golden_csr(d, prefix):
    return csr_ref.RefCsr(d[prefix + "indptr"], d[prefix + "indices"], d[prefix + "data"], d[prefix + "shape"])


def op_from(A, **kw):
    from pykrylov_amd import CsrOperator
    return CsrOperator(A.indptr, A.indices, A.data, A.shape, **kw)


def relerr(x, xref):
    return float(np.linalg.norm(x - xref) / np.linalg.norm(xref))


def dots_for(solver, n):
    return kr.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES[solver]))


# ------------------------------------------------------------------ CG
@pytest.mark.parametrize("gtag", ["zero", "guess"])
def test_cg_diagonal_precon(golden, gtag):
    from pykrylov_amd import CG, DiagonalOperator
    d = golden("precon_jacobi.npz")
    A = golden_csr(d, "spd_A_")
    n = A.shape[0]
    dg = d["spd_d"]
    kw = {} if gtag == "zero" else {"guess": 1.0 + np.arange(n)}
    s = CG(op_from(A, symmetric=True), precon=DiagonalOperator(dg))
    s.solve(d["spd_rhs"], **kw)
    k = "cg_%s_" % gtag
    # the reference's preconditioned CG takes its direction from r instead of precon*r (cg.py:104,150-151) and
    # stalls; that behaviour is reproduced: same matvec count, no convergence, same head of the history
    assert s.nMatvec == int(d[k + "nMatvec"]) and not s.converged
    assert abs(s.residNorm0 - float(d[k + "residNorm0"])) <= 1e-14 * float(d[k + "residNorm0"])
    assert rel_hist_err(s.residHistory[:40], d[k + "residHistory"][:40]) <= 1e-11
    out = kr.cg(A, d["spd_rhs"], precon=lambda v: dg * v, red=dots_for("cg", n), **kw)
    assert out["nMatvec"] == s.nMatvec
    assert np.array_equal(out["residHistory"], np.array(s.residHistory))
    assert np.array_equal(out["x"], s.x)


def test_cg_precon_store_resids_are_preconditioned(golden):
    from pykrylov_amd import CG, DiagonalOperator
    d = golden("precon_jacobi.npz")
    A = golden_csr(d, "spd_A_")
    dg = d["spd_d"]
    s = CG(op_from(A, symmetric=True), precon=DiagonalOperator(dg))
    s.solve(d["spd_rhs"], matvec_max=5, store_resids=True)
    assert len(s.resids) == 6                                           # cg.py:96-97,142-143 store y = precon*r
    assert np.array_equal(s.resids[0], dg * (-d["spd_rhs"]))


def test_operator_precon_equal_to_the_diagonal_one_gives_the_same_bits(golden):
    """A LinearOperator preconditioner (no `.diag`) goes through the host-callback path (test_gpu_hostop.py); with
    the same action as the DiagonalOperator it must reproduce the in-kernel diagonal path bit for bit."""
    from pykrylov_amd import CG, DiagonalOperator, LinearOperator
    d = golden("precon_jacobi.npz")
    A = golden_csr(d, "spd_A_")
    n = A.shape[0]
    dg = d["spd_d"]
    s1 = CG(op_from(A, symmetric=True), precon=DiagonalOperator(dg))
    s1.solve(d["spd_rhs"], matvec_max=40)
    s2 = CG(op_from(A, symmetric=True), precon=LinearOperator(n, n, matvec=lambda v: dg * v, symmetric=True))
    s2.solve(d["spd_rhs"], matvec_max=40)
    assert s1.nMatvec == s2.nMatvec and np.array_equal(s1.x, s2.x)
    assert np.array_equal(s1.residHistory, s2.residHistory)
    with pytest.raises(TypeError):
        CG(op_from(A, symmetric=True), precon=object()).solve(d["spd_rhs"])


# ------------------------------------------------------------------ BiCGSTAB / CGS / TFQMR
def solver_class(name):
    import pykrylov_amd
    return {"bicgstab": pykrylov_amd.BiCGSTAB, "cgs": pykrylov_amd.CGS, "tfqmr": pykrylov_amd.TFQMR}[name]


NONSYM_READY = ["bicgstab", "cgs", "tfqmr"]


@pytest.mark.parametrize("solver", NONSYM_READY)
@pytest.mark.parametrize("gtag", ["zero", "guess"])
def test_nonsymmetric_diagonal_precon(golden, solver, gtag):
    from pykrylov_amd import DiagonalOperator
    d = golden("precon_jacobi.npz")
    A = golden_csr(d, "ns_A_")
    n = A.shape[0]
    dg = d["ns_d"]
    kw = dict(matvec_max=2 * n)
    if gtag == "guess":
        kw["guess"] = 1.0 + np.arange(n)
    s = solver_class(solver)(op_from(A), reltol=1e-8, precon=DiagonalOperator(dg))
    s.solve(d["ns_rhs"], **kw)
    k = "%s_%s_" % (solver, gtag)
    # (i) the reference's own run (np.dot order)
    assert s.nMatvec == int(d[k + "nMatvec"]) and s.converged == bool(d[k + "converged"])
    xref = d[k + "x"]
    scale = np.linalg.norm(kw["guess"] - xref) if gtag == "guess" else np.linalg.norm(xref)
    assert np.linalg.norm(s.x - xref) <= 1e-11 * scale
    assert abs(s.residNorm0 - float(d[k + "residNorm0"])) <= 1e-14 * float(d[k + "residNorm0"])
    # (ii) bit equality with the oracle in the device's summation order
    ref = getattr(kr, solver)(A, d["ns_rhs"], reltol=1e-8, precon=lambda v: dg * v, red=dots_for(solver, n), **kw)
    assert s.nMatvec == ref["nMatvec"]
    assert np.array_equal(s.x, ref["x"])
    assert s.residNorm == ref["residNorm"] and s.residNorm0 == ref["residNorm0"] and s.converged == ref["converged"]


# ------------------------------------------------------------------ MINRES / SYMMLQ
# shift = 1.5 makes the operator indefinite: after the Lanczos vectors lose orthogonality the recurrence is chaotic
# in the summation order (see test_gpu_minres.py), so against the reference's np.dot-order run only the head of
# the trajectory, the iteration count to a few, convergence and the solution are comparable; the bit-level check
# against the oracle in the device's order has no such limit.
@pytest.mark.parametrize("shift", [0.0, 1.5])
def test_minres_diagonal_precon(golden, shift, monkeypatch):
    from pykrylov_amd import Minres, DiagonalOperator
    d = golden("precon_jacobi.npz")
    A = golden_csr(d, "spd_A_")
    n = A.shape[0]
    dg = d["spd_d"]
    k = "minres_s%g_" % shift
    s = Minres(op_from(A, symmetric=True))
    s.solve(d[k + "rhs"], precon=DiagonalOperator(dg), shift=shift, show=False, check=False, etol=0.0, rtol=1e-10)
    definite = shift == 0.0
    assert s.istop == int(d[k + "istop"]) and s.converged
    assert abs(s.itn - int(d[k + "itn"])) <= (0 if definite else 2)
    href = d[k + "residHistory"]
    head = len(href) if definite else 30
    assert rel_hist_err(s.residHistory[:head], href[:head]) <= 1e-11
    assert relerr(s.x, d[k + "x"]) <= (1e-11 if definite else 1e-6)
    for name in ("Anorm", "ynorm", "residNorm0"):
        assert abs(getattr(s, name) - float(d[k + name])) <= (1e-10 if definite else 1e-2) * abs(float(d[k + name])), name
    monkeypatch.setattr(kr, "_sq", lambda a: a * a)          # the reference's pow(x, 2) is not always x*x
    ref = kr.minres(A, d[k + "rhs"], precon=lambda v: dg * v, shift=shift, check=False, etol=0.0, rtol=1e-10,
                    red=dots_for("minres", n))
    assert (s.istop, s.itn) == (ref["istop"], ref["itn"])
    assert np.array_equal(np.array(s.residHistory), ref["residHistory"])
    assert np.array_equal(s.x, ref["x"])
    for name in ("Anorm", "Acond", "Arnorm", "ynorm", "rnorm"):
        assert getattr(s, name) == ref[name], name


@pytest.mark.parametrize("shift", [0.0, 1.5])
def test_symmlq_diagonal_precon(golden, shift, monkeypatch):
    from pykrylov_amd import Symmlq, DiagonalOperator
    d = golden("precon_jacobi.npz")
    A = golden_csr(d, "spd_A_")
    n = A.shape[0]
    dg = d["spd_d"]
    k = "symmlq_s%g_" % shift
    rhs = d["minres_s%g_rhs" % shift]
    s = Symmlq(op_from(A, symmetric=True), precon=DiagonalOperator(dg))
    s.solve(rhs, **({} if shift == 0.0 else {"shift": shift}))
    definite = shift == 0.0
    assert abs(s.nMatvec - int(d[k + "nMatvec"])) <= (0 if definite else 2)
    assert relerr(s.x, d[k + "x"]) <= (1e-11 if definite else 1e-5)
    assert abs(s.anorm - float(d[k + "anorm"])) <= (1e-10 if definite else 1e-2) * float(d[k + "anorm"])
    assert abs(s.xNorm - float(d[k + "xNorm"])) <= (1e-10 if definite else 1e-5) * float(d[k + "xNorm"])
    monkeypatch.setattr(kr, "_sq", lambda a: a * a)
    ref = kr.symmlq(A, rhs, precon=lambda v: dg * v, shift=(shift or None), red=dots_for("symmlq", n))
    assert s.nMatvec == ref["nMatvec"]
    assert np.array_equal(s.x, ref["x"])
    for name in ("residNorm", "xNorm", "anorm", "acond"):
        assert getattr(s, name) == ref[name], name
