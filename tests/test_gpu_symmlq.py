"""GPU parity of the device-resident SYMMLQ with the oracle and the golden traces."""
import numpy as np
import pytest

from oracle import csr_ref, gpu_order, krylov_ref as kr

pytestmark = pytest.mark.gpu


def op_from(A, **kw):
    from pykrylov_amd import CsrOperator
    return CsrOperator(A.indptr, A.indices, A.data, A.shape, **kw)


def golden_csr(d, prefix):
    return csr_ref.RefCsr(d[prefix + "indptr"], d[prefix + "indices"], d[prefix + "data"], d[prefix + "shape"])


@pytest.mark.parametrize("m", [30, 100])
@pytest.mark.parametrize("shift", [0.0, 1.5])
def test_symmlq_bit_exact_with_emulated_dot_order(golden, m, shift, monkeypatch):
    from pykrylov_amd import Symmlq
    monkeypatch.setattr(kr, "_sq", lambda a: a * a)
    d = golden("symmlq_poisson2d.npz")
    A = golden_csr(golden("minres_poisson2d.npz"), "m%d_A_" % m)
    n = m * m
    rhs = d["m%d_s%g_rhs" % (m, shift)]
    s = Symmlq(op_from(A, symmetric=True))
    s.solve(rhs, **({} if shift == 0.0 else {"shift": shift}))
    ref = kr.symmlq(A, rhs, shift=(shift or None),
                    red=kr.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES["symmlq"])))
    assert s.nMatvec == ref["nMatvec"] and s.istop == ref["istop"]
    assert np.array_equal(s.x, ref["x"])
    for name in ("residNorm", "xNorm", "anorm", "acond"):
        assert getattr(s, name) == ref[name], name
    assert s.solutionNorm == s.xNorm and s.bestSolution is s.x


@pytest.mark.parametrize("m", [30, 100])
@pytest.mark.parametrize("shift", [0.0, 1.5])
def test_symmlq_vs_golden(golden, m, shift):
    """Against the reference's own run (np.dot order).  Definite case: same product count, x to 1e-12.
    Indefinite case (shift inside the spectrum): the Lanczos recurrence is chaotic in the summation order
    after the onset of orthogonality loss, so only the count (to a few), x and the norms are compared."""
    from pykrylov_amd import Symmlq
    d = golden("symmlq_poisson2d.npz")
    A = golden_csr(golden("minres_poisson2d.npz"), "m%d_A_" % m)
    k = "m%d_s%g_" % (m, shift)
    op = op_from(A, symmetric=True)
    s = Symmlq(op)
    s.solve(d[k + "rhs"], **({} if shift == 0.0 else {"shift": shift}))
    xref = d[k + "x"]
    if shift == 0.0:
        assert s.nMatvec == int(d[k + "nMatvec"]) == op.nMatvec
        assert np.linalg.norm(s.x - xref) <= 1e-12 * np.linalg.norm(xref)
        assert abs(s.anorm - float(d[k + "anorm"])) <= 1e-12 * float(d[k + "anorm"])
        assert abs(s.acond - float(d[k + "acond"])) <= 1e-10 * float(d[k + "acond"])
        assert abs(s.residNorm - float(d[k + "residNorm"])) <= 1e-6 * float(d[k + "residNorm"])
    else:
        assert abs(s.nMatvec - int(d[k + "nMatvec"])) <= 6
        assert np.linalg.norm(s.x - xref) <= 1e-5 * np.linalg.norm(xref)
    assert abs(s.xNorm - float(d[k + "xNorm"])) <= 1e-6 * float(d[k + "xNorm"])


def test_symmlq_edge_cases(golden):
    from pykrylov_amd import Symmlq
    A = golden_csr(golden("minres_poisson2d.npz"), "m30_A_")
    n = 900
    op = op_from(A, symmetric=True)
    rhs = golden("symmlq_poisson2d.npz")["m30_s0_rhs"]
    s = Symmlq(op)
    s.solve(np.zeros(n))                                       # b = 0: x = 0, only the final-residual product
    assert s.nMatvec == 1 and np.array_equal(s.x, np.zeros(n)) and s.residNorm == 0.0
    for mm in (1, 2, 3, 10):                                   # product limit
        s.solve(rhs, matvec_max=mm)
        ref = kr.symmlq(A, rhs, matvec_max=mm,
                        red=kr.Reductions(gpu_order.GpuDots(n, gpu_order.SPMV_SITES["symmlq"])))
        assert s.nMatvec == ref["nMatvec"], mm
        assert np.allclose(s.x, ref["x"], rtol=1e-13, atol=1e-300) and abs(s.residNorm - ref["residNorm"]) <= 1e-13 * ref["residNorm"]
    s.solve(rhs, check=True)                                   # symmetric operator passes the check
    ref = kr.symmlq(A, rhs)
    assert s.nMatvec == ref["nMatvec"]
    s.solve(rhs, store_iterates=True, rtol=1e-4)
    assert len(s.iterates) >= 2 and s.iterates[0].shape == (n,)
    nonsym = golden_csr(golden("nonsym_jpwh991.npz"), "A_")
    with pytest.raises(NotImplementedError):
        Symmlq(op_from(nonsym)).solve(np.ones(991), check=True)
    sp = Symmlq(op, precon=op)                              # operator preconditioner: host callback path
    sp.solve(rhs, matvec_max=6)
    assert sp.nMatvec >= 6
