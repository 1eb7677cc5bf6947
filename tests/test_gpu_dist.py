"""Single-GPU checks of the multi-GPU plumbing: RCCL bootstrap with one rank, the all-gather
exchange, device-side column localisation and the partitioned operator path of the solvers."""
import ctypes

import numpy as np
import pytest

from conftest import rel_hist_err
from oracle import csr_ref, krylov_ref as kr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world():
    from pykrylov_amd import dist, _lib
    w = dist.World(0, 1, None)
    w.init_device_comm()                 # ncclCommInitRank with nranks = 1
    yield w
    _lib.load().mk_comm_destroy()


@pytest.mark.parametrize("mode", ["halo", "allgather"])
def test_partitioned_cg_single_rank(world, mode):
    from pykrylov_amd import CG, dist
    m = 12
    op, ranges = dist.partition_poisson3d(world, m, m, m, mode=mode)
    n = m ** 3
    assert ranges == [(0, n)] and op.local_size == n
    assert op.shape == (n, n if mode == "halo" else 2 * n)
    A = csr_ref.poisson3d(m)
    rhs = A.matvec(np.ones(n))
    ref = kr.cg(A, rhs)
    s = CG(op)
    s.solve(rhs)
    assert s.nMatvec == ref["nMatvec"]
    assert rel_hist_err(s.residHistory, ref["residHistory"]) <= 1e-12
    assert np.linalg.norm(s.x - ref["x"]) <= 1e-12 * np.linalg.norm(ref["x"])
    g = 1.0 + np.arange(n) / n
    s2 = CG(op)
    s2.solve(rhs, guess=g, matvec_max=30)
    ref2 = kr.cg(A, rhs, guess=g, matvec_max=30)
    assert s2.nMatvec == ref2["nMatvec"] and rel_hist_err(s2.residHistory, ref2["residHistory"]) <= 1e-12


def test_localize_banded_on_device(world):
    """Rows [r0, r1) of a 3-D Poisson matrix: the device remap equals the NumPy plan."""
    from pykrylov_amd import _lib, dist
    from pykrylov_amd.linop import CsrOperator
    lib = _lib.init()
    nx, ny, nz = 6, 5, 8
    A = csr_ref.poisson3d(nx, ny, nz)
    n = nx * ny * nz
    for (c0, c1) in dist.row_ranges(n, 4, align=nx * ny):
        h = ctypes.c_void_p()
        _lib.check(lib.mk_csr_poisson3d(nx, ny, nz, c0, c1, ctypes.byref(h)))
        lo, hi = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(lib.mk_csr_localize(h, 0, c0, c1, 0, ctypes.byref(lo), ctypes.byref(hi)))
        op = CsrOperator.from_handle(h.value)
        indptr, indices, data = op.to_csr_arrays()
        sl = slice(A.indptr[c0], A.indptr[c1])
        need = dist.needed_columns(A.indices[sl], c0, c1)
        assert lo.value == (nx * ny if c0 > 0 else 0) and hi.value == (nx * ny if c1 < n else 0)
        assert np.array_equal(need, dist.banded_needs(c0, c1, lo.value, hi.value))
        assert np.array_equal(indices, dist.localize_columns(A.indices[sl], c0, c1, need))
        assert np.array_equal(indptr, A.indptr[c0:c1 + 1] - A.indptr[c0]) and np.array_equal(data, A.data[sl])
        assert op.shape == (c1 - c0, c1 - c0 + lo.value + hi.value)
