"""GPU parity of the device-resident BiCGSTAB / CGS / TFQMR with the oracle and the golden traces.

These solvers record no history in the reference; the observables are x, residNorm, residNorm0,
nMatvec and converged.  Parity levels:
  * bit-exact against the oracle run with the device's dot summation order (every other operation
    rounds identically);
  * against the reference's own run (np.dot order): same product count on the fixture problems, x to
    1e-12 relative where the problem is well conditioned.
"""
import numpy as np
import pytest

from oracle import csr_ref, gpu_order, krylov_ref as kr

pytestmark = pytest.mark.gpu


def op_from(A, **kw):
    from pykrylov_amd import CsrOperator
    return CsrOperator(A.indptr, A.indices, A.data, A.shape, **kw)


def golden_csr(d, prefix):
    return csr_ref.RefCsr(d[prefix + "indptr"], d[prefix + "indices"], d[prefix + "data"], d[prefix + "shape"])


def solver_class(name):
    import pykrylov_amd
    return {"bicgstab": pykrylov_amd.BiCGSTAB, "cgs": getattr(pykrylov_amd, "CGS", None),
            "tfqmr": getattr(pykrylov_amd, "TFQMR", None)}[name]


SOLVERS = ["bicgstab", "cgs", "tfqmr"]


@pytest.mark.parametrize("solver", SOLVERS)
@pytest.mark.parametrize("fix", ["nonsym_jpwh991.npz", "nonsym_rand10k.npz"])
@pytest.mark.parametrize("tol", [1e-5, 1e-8])
@pytest.mark.parametrize("gtag", ["guess", "zero"])
def test_bit_exact_with_emulated_dot_order(golden, solver, fix, tol, gtag):
    cls = solver_class(solver)
    if cls is None:
        pytest.skip("%s not built yet" % solver)
    d = golden(fix)
    A = golden_csr(d, "A_")
    n = A.shape[0]
    kw = dict(matvec_max=2 * n)
    if gtag == "guess":
        kw["guess"] = 1.0 + np.arange(n)
    if fix == "nonsym_jpwh991.npz" and gtag == "zero":
        kw["matvec_max"] = 60          # breakdown case (NaN from the first pass on): keep it short
    s = cls(op_from(A), reltol=tol)
    s.solve(d["rhs"], **kw)
    with np.errstate(all="ignore"):
        ref = getattr(kr, solver)(A, d["rhs"], reltol=tol,
                                  red=kr.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES[solver])), **kw)
    assert s.nMatvec == ref["nMatvec"]
    assert np.array_equal(s.x, ref["x"], equal_nan=True)
    assert np.array_equal(s.residNorm, ref["residNorm"], equal_nan=True)
    assert s.residNorm0 == ref["residNorm0"] and s.converged == ref["converged"]


@pytest.mark.parametrize("solver", SOLVERS)
@pytest.mark.parametrize("tol", [1e-5, 1e-8])
@pytest.mark.parametrize("gtag", ["guess", "zero"])
def test_rand10k_vs_reference_run(golden, solver, tol, gtag):
    """Diagonally dominant matrix: the np.dot-order reference run is reproduced to 1e-12."""
    cls = solver_class(solver)
    if cls is None:
        pytest.skip("%s not built yet" % solver)
    d = golden("nonsym_rand10k.npz")
    A = golden_csr(d, "A_")
    n = A.shape[0]
    kw = dict(matvec_max=2 * n)
    if gtag == "guess":
        kw["guess"] = 1.0 + np.arange(n)
    op = op_from(A)
    s = cls(op, reltol=tol)
    s.solve(d["rhs"], **kw)
    k = "%s_%g_%s_" % (solver, tol, gtag)
    assert s.nMatvec == int(d[k + "nMatvec"])
    assert op.nMatvec == s.nMatvec + (1 if (gtag == "guess" and solver != "bicgstab") else 0)
    # x = guess + (accumulated updates): rounding noise scales with the size of the correction, which for
    # guess = 1..n is 5.8e5 (the iterates lose ~6e-11 absolute), hence the scale below
    scale = np.linalg.norm(d[k + "x"]) if gtag == "zero" else np.linalg.norm(d[k + "x"] - kw["guess"])
    assert np.linalg.norm(s.x - d[k + "x"]) <= 1e-12 * scale
    assert abs(s.residNorm0 - float(d[k + "residNorm0"])) <= 1e-14 * float(d[k + "residNorm0"])
    # the final residual is ~1e-9 of residNorm0 and carries the recurrence's rounding noise, which is
    # proportional to residNorm0: 1e-12 relative to the initial residual
    assert abs(s.residNorm - float(d[k + "residNorm"])) <= 1e-12 * s.residNorm0
    assert s.converged == bool(d[k + "converged"]) and s.bestSolution is s.x


@pytest.mark.parametrize("solver", SOLVERS)
def test_jpwh991_doc_protocol(golden, solver):
    """examples/bmark.py protocol (guess = 1 + arange(n), matvec_max = 2n): product counts of the
    reference run are reproduced within the spread the summation order causes on this matrix."""
    cls = solver_class(solver)
    if cls is None:
        pytest.skip("%s not built yet" % solver)
    d = golden("nonsym_jpwh991.npz")
    A = golden_csr(d, "A_")
    n = A.shape[0]
    for tol in (1e-5, 1e-8):
        s = cls(op_from(A), reltol=tol)
        s.solve(d["rhs"], guess=1.0 + np.arange(n), matvec_max=2 * n)
        k = "%s_%g_guess_" % (solver, tol)
        assert abs(s.nMatvec - int(d[k + "nMatvec"])) <= 4 and s.converged
        assert np.linalg.norm(s.x - 1.0) / np.sqrt(n) < 1e-2


@pytest.mark.parametrize("solver", SOLVERS)
def test_edge_cases(golden, solver):
    cls = solver_class(solver)
    if cls is None:
        pytest.skip("%s not built yet" % solver)
    d = golden("nonsym_rand10k.npz")
    A = golden_csr(d, "A_")
    n = A.shape[0]
    op = op_from(A)
    s = cls(op)
    s.solve(np.zeros(n))                                      # zero rhs: nothing to do
    assert s.nMatvec == 0 and s.residNorm == 0.0 and np.array_equal(s.x, np.zeros(n))
    for mm in (0, 1, 2, 3, 5):                                # product limits hit at every exit point
        s = cls(op)
        s.solve(d["rhs"], matvec_max=mm)
        ref = getattr(kr, solver)(A, d["rhs"], matvec_max=mm,
                                  red=kr.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES[solver])))
        assert s.nMatvec == ref["nMatvec"], mm
        assert np.array_equal(s.x, ref["x"]) and s.residNorm == ref["residNorm"] and s.converged == ref["converged"]
    rhs = d["rhs"].copy()
    s = cls(op, abstol=1e-30, reltol=1e-14)                   # exact solution as guess
    s.solve(rhs, guess=np.ones(n))
    assert np.array_equal(rhs, d["rhs"])
    with pytest.raises(ValueError):
        s.solve(np.ones(n + 1))


def test_bicgstab_config3_n1e6(golden):
    """BASELINE config 3: random nonsymmetric n = 1e6, 5 nnz/row, reltol = 1e-10."""
    from pykrylov_amd import BiCGSTAB, gallery
    d = golden("large_summaries.npz")
    indptr, indices, data, shape = gallery.random_diagdom_csr(1000000, seed=1)
    import hashlib
    assert np.array_equal(np.frombuffer(hashlib.sha256(indices.tobytes()).digest(), dtype=np.uint8),
                          d["rand1m_indices_sha"])
    assert np.array_equal(np.frombuffer(hashlib.sha256(data.tobytes()).digest(), dtype=np.uint8), d["rand1m_data_sha"])
    from pykrylov_amd import CsrOperator
    op = CsrOperator(indptr, indices, data, shape)
    rhs = op * np.ones(shape[0])
    s = BiCGSTAB(op, reltol=1e-10)
    s.solve(rhs)
    assert s.nMatvec == int(d["rand1m_bicgstab_nMatvec"]) == 44 and s.converged
    assert np.max(np.abs(s.x[::997] - d["rand1m_bicgstab_x_sample"])) <= 1e-12
    assert abs(s.residNorm0 - float(d["rand1m_bicgstab_residNorm0"])) <= 1e-13 * s.residNorm0
