"""Order-independent anchors for the solver families beyond CG.

north_star's "1e-12 against the NumPy reference" is a statement about the distance to the loop evaluated with EXACT inner
products: the reference's own np.dot order is itself 1e-13 ... 5e-12 away from that, by problem size.  The anchor is the
oracle loop with every inner product formed in extended precision and rounded once (oracle/gpu_order.py ExactDots,
within 1e-18 of the exactly rounded dot).  Each test asserts device-vs-anchor <= 1e-12 and prints np.dot-vs-anchor beside
it (DESIGN.md section 4 quotes the printed figures).  CG's anchors: tests/test_gpu_full_size.py."""
import numpy as np
import pytest

from conftest import rel_hist_err
from oracle import csr_ref, gpu_order, krylov_ref as kr, lls_ref

pytestmark = pytest.mark.gpu


def exact():
    return kr.Reductions(gpu_order.ExactDots())


def relerr(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(np.asarray(b)))


@pytest.mark.slow
def test_minres_config4_head_against_exact_dots():
    """BASELINE configs[3]: MINRES on the shifted 2-D Laplacian, n = 4e6, shift 1.5 (indefinite).  The Lanczos recurrence
    with the shift inside the spectrum is chaotic in the long run (any two summation orders part ways after a few
    hundred passes), so the HEAD of the trajectory is what can be held to 1e-12: the first 120 passes."""
    from pykrylov_amd import Minres, gallery
    m, passes = 2000, 120
    A = csr_ref.poisson2d(m)
    n = m * m
    rhs = A.matvec(np.ones(n)) - 1.5
    anchor = kr.minres(A, rhs, shift=1.5, check=False, etol=0.0, rtol=0.0, itnlim=passes, red=exact())
    blas = kr.minres(A, rhs, shift=1.5, check=False, etol=0.0, rtol=0.0, itnlim=passes)
    op = gallery.poisson2d(m)
    s = Minres(op)
    s.solve(rhs, shift=1.5, show=False, check=False, etol=0.0, rtol=0.0, itnlim=passes)
    assert s.itn == anchor["itn"] == blas["itn"] == passes
    dev = rel_hist_err(s.residHistory, anchor["residHistory"])
    ref = rel_hist_err(blas["residHistory"], anchor["residHistory"])
    xdev, xref = relerr(s.x, anchor["x"]), relerr(blas["x"], anchor["x"])
    print("MINRES config 4 (n = 4e6, shift 1.5), first %d passes: history device vs anchor %.2e, np.dot vs anchor %.2e; "
          "x device %.2e, np.dot %.2e" % (passes, dev, ref, xdev, xref))
    assert dev <= 1e-12 and xdev <= 1e-12, (dev, xdev)
    op.free()


def test_minres_shift0_full_run_against_exact_dots():
    """The same operator family without the shift (SPD: regular convergence), m = 500 (n = 250 000), the whole run."""
    from pykrylov_amd import Minres, gallery
    m = 500
    A = csr_ref.poisson2d(m)
    n = m * m
    rhs = A.matvec(np.ones(n))
    anchor = kr.minres(A, rhs, check=False, etol=0.0, rtol=1e-10, red=exact())
    blas = kr.minres(A, rhs, check=False, etol=0.0, rtol=1e-10)
    op = gallery.poisson2d(m)
    s = Minres(op)
    s.solve(rhs, show=False, check=False, etol=0.0, rtol=1e-10)
    assert s.itn == anchor["itn"] == blas["itn"] and s.istop == anchor["istop"]
    dev = rel_hist_err(s.residHistory, anchor["residHistory"])
    ref = rel_hist_err(blas["residHistory"], anchor["residHistory"])
    xdev, xref = relerr(s.x, anchor["x"]), relerr(blas["x"], anchor["x"])
    print("MINRES 2-D Poisson m = 500, shift 0, all %d passes: history device vs anchor %.2e, np.dot vs anchor %.2e; "
          "x device %.2e, np.dot %.2e" % (s.itn, dev, ref, xdev, xref))
    assert dev <= 1e-12 and xdev <= 1e-12, (dev, xdev)
    op.free()


@pytest.mark.parametrize("solver", ["bicgstab", "cgs", "tfqmr"])
def test_nonsymmetric_config3_against_exact_dots(solver):
    """BASELINE configs[2]: the random diagonally dominant n = 1e6 matrix.  These loops keep no history (bicgstab.py:148-
    151); what a run leaves is the iterate, the product count and the last residual norm, after ~20 passes of a
    recurrence whose every scalar comes from inner products over 1e6 terms."""
    from pykrylov_amd import BiCGSTAB, CGS, TFQMR, gallery
    n = 1000000
    A = csr_ref.random_diagdom(n, seed=1)
    rhs = A.matvec(np.ones(n))
    fn = dict(bicgstab=kr.bicgstab, cgs=kr.cgs, tfqmr=kr.tfqmr)[solver]
    cls = dict(bicgstab=BiCGSTAB, cgs=CGS, tfqmr=TFQMR)[solver]
    anchor = fn(A, rhs, reltol=1e-10, red=exact())
    blas = fn(A, rhs, reltol=1e-10)
    op = gallery.random_diagdom(n, seed=1)
    s = cls(op, reltol=1e-10)
    s.solve(rhs)
    xdev, xref = relerr(s.x, anchor["x"]), relerr(blas["x"], anchor["x"])
    rdev = abs(float(s.residNorm) - anchor["residNorm"]) / anchor["residNorm0"]
    rref = abs(blas["residNorm"] - anchor["residNorm"]) / anchor["residNorm0"]
    print("%s config 3 (n = 1e6 random), %d / %d / %d products (device / anchor / np.dot): x device vs anchor %.2e, np.dot vs "
          "anchor %.2e; last residual norm / r0: device %.2e, np.dot %.2e" % (solver, s.nMatvec, anchor["nMatvec"],
                                                                               blas["nMatvec"], xdev, xref, rdev, rref))
    assert s.nMatvec == anchor["nMatvec"] and s.converged
    assert xdev <= 1e-12 and rdev <= 1e-12, (xdev, rdev)
    op.free()


@pytest.mark.parametrize("solver", ["lsqr", "lsmr", "craig", "craigmr"])
def test_least_squares_against_exact_dots(golden, solver):
    """The 2000 x 1500 fixture problem with etol = 0 (no direct-error stop) for 40 passes of the Golub-Kahan process -- before
    its vectors lose orthogonality, which is when ANY two summation orders start to drift: iterate and norm estimates."""
    from pykrylov_amd import CsrOperator, lls
    d = golden("lls_random.npz")
    A = csr_ref.RefCsr(d["l_A_indptr"], d["l_A_indices"], d["l_A_data"], d["l_A_shape"])
    At = A.transpose()
    b = d["l_b_cons"]
    passes = 40
    fn = dict(lsqr=lls_ref.lsqr, lsmr=lls_ref.lsmr, craig=lls_ref.craig, craigmr=lls_ref.craigmr)[solver]
    kw = dict(itnlim=passes, etol=0.0)
    if solver in ("lsqr", "lsmr", "craig"):
        kw.update(atol=0.0, btol=0.0)
    if solver in ("lsqr", "lsmr"):
        kw.update(conlim=0.0)
    anchor = fn(A.matvec, At.matvec, A.shape, b.copy(), red=exact(), **kw)
    blas = fn(A.matvec, At.matvec, A.shape, b.copy(), **kw)
    op = CsrOperator(A.indptr, A.indices, A.data, A.shape)
    cls = dict(lsqr=lls.LSQRFramework, lsmr=lls.LSMRFramework, craig=lls.CRAIGFramework, craigmr=lls.CRAIGMRFramework)[solver]
    s = cls(op)
    ret = s.solve(b, **kw)
    x = np.asarray(ret[0]) if solver == "lsmr" else np.asarray(s.x)
    itn = int(ret[2]) if solver == "lsmr" else int(s.itn)
    assert itn == anchor["itn"] == blas["itn"] == passes
    xdev, xref = relerr(x, anchor["x"]), relerr(blas["x"], anchor["x"])
    print("%s 2000 x 1500, etol = 0, %d passes: x device vs anchor %.2e, np.dot vs anchor %.2e" % (solver, passes, xdev, xref))
    assert xdev <= 1e-12, xdev
    op.free()


def test_symmlq_full_run_against_exact_dots():
    """SYMMLQ (the other Lanczos loop) on the 2-D Poisson matrix m = 300, the whole run: iterate, residual norm, product count."""
    from pykrylov_amd import Symmlq, gallery
    m = 300
    A = csr_ref.poisson2d(m)
    n = m * m
    rhs = A.matvec(np.ones(n))
    anchor = kr.symmlq(A, rhs, rtol=1e-10, red=exact())
    blas = kr.symmlq(A, rhs, rtol=1e-10)
    op = gallery.poisson2d(m)
    s = Symmlq(op)
    s.solve(rhs, rtol=1e-10)
    assert s.nMatvec == anchor["nMatvec"] == blas["nMatvec"] and s.istop == anchor["istop"]
    xdev, xref = relerr(s.x, anchor["x"]), relerr(blas["x"], anchor["x"])
    rdev = abs(float(s.residNorm) - anchor["residNorm"]) / np.linalg.norm(rhs)
    print("SYMMLQ 2-D Poisson m = 300, all %d products: x device vs anchor %.2e, np.dot vs anchor %.2e; residual norm / |b|: "
          "device %.2e" % (s.nMatvec, xdev, xref, rdev))
    assert xdev <= 1e-12 and rdev <= 1e-12, (xdev, rdev)
    op.free()
