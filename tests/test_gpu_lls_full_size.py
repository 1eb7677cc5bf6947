"""The least-squares workload of bench.py at its own size: the seeded 4e6 x 1e6 matrix with 5 entries per row.  Its products run
on round 4's scattered-matrix paths -- `A v`: format 3, pair kernel, one launch per step (15 625 tiles = 3.8 steps per workgroup);
`A' u`: 1e6 rows of ~20 entries over 32 MB of u, four 8 MiB column blocks as resident tiles with their own column phases -- which the small fixtures never reach."""
import ctypes

import numpy as np
import pytest

from oracle import csr_ref, lls_ref

pytestmark = pytest.mark.gpu
M, N = 4000000, 1000000


@pytest.fixture(scope="module")
def tall():
    import bench
    from pykrylov_amd import CsrOperator
    indptr, indices, data = bench.random_tall_csr(M, N)
    A = csr_ref.RefCsr(indptr, indices, data, (M, N))
    op = CsrOperator(indptr, indices, data, (M, N))
    yield A, op
    op.free()


def test_both_products_bit_exact_and_on_the_round4_paths(tall):
    from pykrylov_amd import _lib
    lib = _lib.init()
    A, op = tall
    rng = np.random.default_rng(21)
    x, u = rng.standard_normal(N), rng.standard_normal(M)
    y = op * x
    assert np.array_equal(y, A.matvec(x))                                  # 4e6 rows, every one to the last bit
    At = op.T
    v = At * u
    assert np.array_equal(v, A.rmatvec(u))                                 # A' u through the column blocks: same bits
    fmt, nb = ctypes.c_int32(), ctypes.c_int32()
    _lib.check(lib.mk_csr_format_info(op.handle, ctypes.byref(fmt), None, None, None, None))
    assert fmt.value == 3                                                  # (pair kernel, stepped: 15 625 tiles on 2 048 workgroups)
    _lib.check(lib.mk_csr_colblocks(At.handle, ctypes.byref(nb)))
    assert nb.value == 4                                                   # 32 MB of u in 8 MiB blocks, chosen automatically
    # adjoint identity on the device products
    lhs, rhs = float(np.dot(y, u)), float(np.dot(x, v))
    assert abs(lhs - rhs) <= 1e-12 * np.linalg.norm(y) * np.linalg.norm(u)


@pytest.mark.parametrize("solver", ["lsqr", "craigmr"])
def test_first_passes_match_the_oracle(tall, solver):
    """25 passes of the loop at full size against the oracle (np.dot order): iterate to 1e-11, same pass count."""
    from pykrylov_amd import lls
    A, op = tall
    xs = np.random.default_rng(12).standard_normal(N)
    b = A.matvec(xs)
    At = A.transpose()
    if solver == "lsqr":
        ref = lls_ref.lsqr(A.matvec, At.matvec, A.shape, b.copy(), itnlim=25, etol=0.0, atol=0.0, btol=0.0, conlim=0.0)
        s = lls.LSQRFramework(op)
        s.solve(b, itnlim=25, etol=0.0, atol=0.0, btol=0.0, conlim=0.0)
    else:
        ref = lls_ref.craigmr(A.matvec, At.matvec, A.shape, b.copy(), itnlim=25, etol=0.0)
        s = lls.CRAIGMRFramework(op)
        s.solve(b, itnlim=25, etol=0.0)
    assert int(s.itn) == ref["itn"] == 25
    err = float(np.linalg.norm(np.asarray(s.x) - ref["x"]) / np.linalg.norm(ref["x"]))
    print("%s 4e6 x 1e6, 25 passes: x device vs oracle (np.dot order) %.2e" % (solver, err))
    assert err <= 1e-11
