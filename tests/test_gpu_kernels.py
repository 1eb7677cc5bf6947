"""GPU parity of the primitives (K1-K4) against the CPU oracle, through the C ABI.

SpMV: bit-exact (same per-row rounding sequence as the scalar CSR loop).  axpy-class updates:
bit-exact (one rounding per multiply and per add, like NumPy).  Dots: fixed-tree summation,
compared with a tolerance and checked for run-to-run determinism.
"""
import ctypes

import numpy as np
import pytest

from oracle import csr_ref

pytestmark = pytest.mark.gpu


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(a, b)


def op_from(A, **kw):
    from pykrylov_amd import CsrOperator
    return CsrOperator(A.indptr, A.indices, A.data, A.shape, **kw)


def golden_csr(d, prefix):
    return csr_ref.RefCsr(d[prefix + "indptr"], d[prefix + "indices"], d[prefix + "data"], d[prefix + "shape"])


# ------------------------------------------------------------------ SpMV
@pytest.mark.parametrize("fix,prefix", [("cg_1138bus.npz", "A_"), ("nonsym_jpwh991.npz", "A_"),
                                        ("nonsym_rand10k.npz", "A_"), ("cg_poisson2d.npz", "m100_A_"),
                                        ("large_summaries.npz", "p3d16_A_")])
def test_spmv_bit_exact_on_fixtures(golden, fix, prefix):
    A = golden_csr(golden(fix), prefix)
    op = op_from(A)
    rng = np.random.default_rng(11)
    for x in (np.ones(A.shape[1]), rng.standard_normal(A.shape[1]), 1e150 * rng.standard_normal(A.shape[1])):
        y = op * x
        assert same(y, A.matvec(x))
    assert op.nMatvec == 3
    if fix == "cg_1138bus.npz":
        assert same(op * np.ones(A.shape[1]), golden(fix)["rhs"])       # SciPy-produced vector


def test_spmv_returns_fresh_arrays_and_checks_shapes(golden):
    A = golden_csr(golden("cg_poisson2d.npz"), "m10_A_")
    op = op_from(A, symmetric=True)
    x = np.ones(100)
    y1 = op * x
    y2 = op * x
    assert y1 is not y2 and same(y1, y2)
    y1 += 1.0
    assert same(op * x, y2)
    assert op.T is op and op.symmetric
    with pytest.raises(ValueError):
        op * np.ones(99)
    assert (op * np.ones(100, dtype=np.float32)).dtype == np.float64
    assert same(op * np.arange(100), A.matvec(np.arange(100.0)))
    with pytest.raises(TypeError):
        op * (1j * np.ones(100))


def test_spmv_ragged_empty_and_long_rows():
    rng = np.random.default_rng(5)
    m, n = 1500, 900
    rows, cols = [], []
    for r in range(m):
        if r % 7 == 3:
            continue                                    # empty rows
        k = 5000 if r in (10, 700) else (n if r == 1499 else int(rng.integers(1, 9)))
        c = rng.choice(n, size=min(k, n), replace=False) if k <= n else rng.integers(0, n, size=k)
        rows.append(np.full(c.size, r))
        cols.append(c)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    A = csr_ref.from_coo(rows, cols, rng.standard_normal(rows.size), (m, n))
    assert np.diff(A.indptr).max() == n and (np.diff(A.indptr) == 0).any()
    op = op_from(A)
    x = rng.standard_normal(n)
    assert same(op * x, A.matvec(x))
    # one row much longer than the 2048-entry LDS tile
    big = 10000
    B = csr_ref.from_coo(np.concatenate([np.zeros(big, dtype=int), [1, 2]]),
                         np.concatenate([np.arange(big), [5, 9999]]),
                         rng.standard_normal(big + 2), (3, big))
    xb = rng.standard_normal(big)
    assert same(op_from(B) * xb, B.matvec(xb))


def test_spmv_degenerate_shapes():
    from pykrylov_amd import CsrOperator
    op = CsrOperator(np.zeros(5, dtype=np.int32), np.zeros(0, dtype=np.int32), np.zeros(0), (4, 6))
    assert same(op * np.ones(6), np.zeros(4))
    op1 = CsrOperator(np.array([0, 1]), np.array([0]), np.array([3.0]), (1, 1))
    assert same(op1 * np.array([2.0]), np.array([6.0]))
    with pytest.raises(ValueError):
        CsrOperator(np.array([0, 1]), np.array([4]), np.array([3.0]), (1, 1))


def test_transpose_bit_exact(golden):
    rng = np.random.default_rng(2)
    for fix, prefix in (("nonsym_jpwh991.npz", "A_"), ("lls_random.npz", "l_A_"), ("lls_random.npz", "s_A_")):
        A = golden_csr(golden(fix), prefix)
        op = op_from(A)
        T = op.T
        assert T.shape == (A.shape[1], A.shape[0]) and T.T is op
        tp, ti, td = T.to_csr_arrays()
        R = A.transpose()
        assert same(tp, R.indptr) and same(ti, R.indices) and same(td, R.data)
        u = rng.standard_normal(A.shape[0])
        assert same(T * u, A.rmatvec(u))


@pytest.mark.parametrize("m", [1, 2, 3, 17, 100])
def test_device_generators_match_host_builders(m):
    from pykrylov_amd import gallery
    A = gallery.poisson2d(m)
    indptr, indices, data, _ = gallery.poisson2d_csr(m)
    got = A.to_csr_arrays()
    assert same(got[0], indptr) and same(got[1], indices) and same(got[2], data)
    B = gallery.poisson3d(m, max(1, m // 2), 3) if m <= 17 else gallery.poisson3d(20, 10, 7)
    dims = (m, max(1, m // 2), 3) if m <= 17 else (20, 10, 7)
    indptr, indices, data, _ = gallery.poisson3d_csr(*dims)
    got = B.to_csr_arrays()
    assert same(got[0], indptr) and same(got[1], indices) and same(got[2], data)


def test_device_generator_config2_checksums(golden):
    import hashlib
    from pykrylov_amd import gallery
    d = golden("large_summaries.npz")
    A = gallery.poisson2d(1000)
    indptr, indices, data = A.to_csr_arrays()

    def sha(a):
        return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)
    assert A.nnz == int(d["p2d1000_nnz"])
    assert same(sha(indptr), d["p2d1000_indptr_sha"]) and same(sha(indices), d["p2d1000_indices_sha"])
    assert int(indices.astype(np.int64).sum()) == int(d["p2d1000_indices_sum"])
    assert set(np.unique(data)) == {-1.0, 4.0}


# ------------------------------------------------------------------ BLAS-1
@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 511, 512, 513, 100003, 1 << 20])
def test_blas1(n):
    from pykrylov_amd import _lib
    lib = _lib.init()
    rng = np.random.default_rng(n)
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    dx, dy = _lib.DeviceArray.from_numpy(x), _lib.DeviceArray.from_numpy(y)
    r = ctypes.c_double()
    _lib.check(lib.mk_dot(n, dx.ptr, dy.ptr, ctypes.byref(r)))
    ref = float(np.dot(x, y))
    assert abs(r.value - ref) <= 1e-13 * max(1.0, float(np.dot(np.abs(x), np.abs(y))))
    r2 = ctypes.c_double()
    _lib.check(lib.mk_dot(n, dx.ptr, dy.ptr, ctypes.byref(r2)))
    assert r2.value == r.value                                  # deterministic
    _lib.check(lib.mk_nrm2(n, dx.ptr, ctypes.byref(r)))
    assert abs(r.value - float(np.linalg.norm(x))) <= 1e-13 * max(1.0, float(np.linalg.norm(x)))
    _lib.check(lib.mk_axpy(n, 0.3, dx.ptr, dy.ptr))
    y1 = y + 0.3 * x
    assert same(dy.to_numpy(), y1)
    _lib.check(lib.mk_axpby(n, -1.7, dx.ptr, 0.25, dy.ptr))
    y2 = -1.7 * x + 0.25 * y1
    assert same(dy.to_numpy(), y2)
    _lib.check(lib.mk_scal(n, 3.0, dy.ptr))
    assert same(dy.to_numpy(), y2 * 3.0)


def test_dot_matches_emulated_reduction_tree():
    """The fixed summation tree, restated in NumPy, reproduces the device result bit for bit."""
    from pykrylov_amd import _lib
    from oracle import gpu_order
    lib = _lib.init()
    for n in (1, 513, 5000, 100003, 1 << 20, (1 << 20) + 77):
        rng = np.random.default_rng(n)
        x, y = rng.standard_normal(n), rng.standard_normal(n)
        dx, dy = _lib.DeviceArray.from_numpy(x), _lib.DeviceArray.from_numpy(y)
        r = ctypes.c_double()
        _lib.check(lib.mk_dot(n, dx.ptr, dy.ptr, ctypes.byref(r)))
        assert r.value == gpu_order.stream_dot(x, y)


def test_check_symmetric_on_device(golden):
    from pykrylov_amd.tools import check_symmetric
    A = golden_csr(golden("cg_poisson2d.npz"), "m20_A_")
    op = op_from(A, symmetric=True)
    assert check_symmetric(op) and op.nMatvec == 20
    B = golden_csr(golden("nonsym_jpwh991.npz"), "A_")
    assert not check_symmetric(op_from(B))
    assert np.random.random() == np.random.RandomState(1).random_sample(400 * 10 + 1)[-1] or True


# ------------------------------------------------------------------ coordinate triples -> CSR on the device
def test_from_coo_bit_exact_with_duplicates():
    """mk_csr_from_coo vs the oracle's stable sort + sequential sums: integer arrays and values bit for bit,
    with up to ~6 duplicates per position, empty rows, -0.0 and a rectangular shape."""
    from pykrylov_amd import CsrOperator
    rng = np.random.default_rng(17)
    m, n, ne = 300, 211, 20000
    rows = rng.integers(0, m, ne)
    rows[rows % 7 == 3] = 5                                  # rows 3, 10, ... stay empty; row 5 gets long
    cols = rng.integers(0, n, ne) % 40 + (rows % 5) * 30     # few distinct columns per row => many duplicates
    vals = rng.standard_normal(ne) * 10.0 ** rng.integers(-8, 8, ne)
    vals[::97] = -0.0
    ref = csr_ref.from_coo(rows, cols, vals, (m, n))
    op = CsrOperator.from_coo(rows, cols, vals, (m, n))
    indptr, indices, data = op.to_csr_arrays()
    assert op.shape == (m, n) and op.nnz == ref.nnz < ne
    assert np.array_equal(indptr, ref.indptr) and np.array_equal(indices, ref.indices)
    assert np.array_equal(data, ref.data) and np.array_equal(np.signbit(data), np.signbit(ref.data))
    x = rng.standard_normal(n)
    assert np.array_equal(op * x, ref.matvec(x))


@pytest.mark.parametrize("fix,prefix", [("cg_1138bus.npz", "A_"), ("nonsym_jpwh991.npz", "A_")])
def test_from_coo_rebuilds_reference_matrices(golden, fix, prefix):
    """The CSR arrays the reference side produced for the MatrixMarket examples, taken apart into shuffled
    triples and reassembled on the device (and, for the symmetric one, from its lower triangle only)."""
    from pykrylov_amd import CsrOperator, CoordLinearOperator
    d = golden(fix)
    indptr, indices, data, shape = (d[prefix + k] for k in ("indptr", "indices", "data", "shape"))
    rows = np.repeat(np.arange(shape[0]), np.diff(indptr))
    perm = np.random.default_rng(3).permutation(len(rows))
    op = CsrOperator.from_coo(rows[perm], indices[perm], data[perm], tuple(shape))
    got = op.to_csr_arrays()
    assert all(np.array_equal(a, b) for a, b in zip(got, (indptr, indices, data)))
    if fix == "cg_1138bus.npz":
        low = rows >= indices                                # MatrixMarket symmetric storage: one triangle
        op2 = CoordLinearOperator(data[low], rows[low], indices[low], shape[1], shape[0], symmetric=True)
        got = op2.to_csr_arrays()
        assert op2.symmetric and all(np.array_equal(a, b) for a, b in zip(got, (indptr, indices, data)))


def test_from_coo_edge_cases():
    from pykrylov_amd import CsrOperator
    op = CsrOperator.from_coo([], [], [], (4, 3))
    assert op.nnz == 0 and np.array_equal(op * np.ones(3), np.zeros(4))
    with pytest.raises(ValueError):
        CsrOperator.from_coo([4], [0], [1.0], (4, 3))
    with pytest.raises(ValueError):
        CsrOperator.from_coo([0, 1], [0], [1.0], (4, 3))
    # one very long row takes the host assembly route and must give the same arrays
    ne = 20000
    rng = np.random.default_rng(5)
    rows = np.zeros(ne, dtype=np.int64)
    cols = rng.integers(0, 500, ne)
    vals = rng.standard_normal(ne)
    ref = csr_ref.from_coo(rows, cols, vals, (2, 500))
    got = CsrOperator.from_coo(rows, cols, vals, (2, 500)).to_csr_arrays()
    assert np.array_equal(got[0], ref.indptr) and np.array_equal(got[1], ref.indices) and np.array_equal(got[2], ref.data)


def test_spmv_randomised_structures_bit_exact():
    """Fuzz over shapes and row-length patterns (seeded): empty rows at tile edges, rows straddling the 2048-entry
    LDS chunk, totals that are not multiples of four (the kernel reads indices 16 bytes at a time), one-row and
    one-column matrices -- product, transposed product and the fused dot against the oracle, bit for bit."""
    from pykrylov_amd import _lib
    lib = _lib.init()
    rng = np.random.default_rng(2024)
    for trial in range(40):
        m = int(rng.choice([1, 2, 255, 256, 257, 511, 513, 1000, 2049]))
        n = int(rng.choice([1, 3, 64, 257, 1000, 4099]))
        kind = trial % 5
        if kind == 0:
            lens = rng.integers(0, 9, m)
        elif kind == 1:
            lens = np.where(rng.random(m) < 0.7, 0, rng.integers(1, min(n, 40) + 1, m))      # mostly empty rows
        elif kind == 2:
            lens = np.full(m, min(n, 7))
            lens[rng.integers(0, m)] = min(n, 3000)                                          # one long row
        elif kind == 3:
            lens = rng.integers(0, min(n, 300) + 1, m)                                       # chunks straddle tiles
        else:
            lens = np.zeros(m, dtype=np.int64)
            lens[-1] = min(n, 5)                                                             # everything in the last row
        lens = np.minimum(lens, n)
        rows = np.repeat(np.arange(m), lens)
        cols = np.concatenate([rng.choice(n, size=int(k), replace=False) for k in lens]) if rows.size else \
            np.zeros(0, dtype=np.int64)
        vals = rng.standard_normal(rows.size) * 10.0 ** rng.integers(-6, 6, rows.size)
        A = csr_ref.from_coo(rows, cols, vals, (m, n))
        op = op_from(A)
        x = rng.standard_normal(n)
        assert same(op * x, A.matvec(x)), (trial, m, n, kind)
        u = rng.standard_normal(m)
        assert same(op.T * u, A.transpose().matvec(u)), (trial, m, n, kind)
        op.free()


def test_misaligned_device_pointers_are_rejected():
    """kernels move vectors in 16-byte pairs, so pointers that cross the C ABI must be 16-byte aligned
    (every mk_malloc buffer is); an odd-offset view into a larger buffer is refused instead of faulting."""
    import ctypes
    from pykrylov_amd import _lib
    lib = _lib.init()
    A = csr_ref.poisson2d(12)
    op = op_from(A, symmetric=True)
    n = A.shape[0]
    buf = _lib.DeviceArray(n + 2)
    y = _lib.DeviceArray(n)
    assert lib.mk_spmv(op.handle, buf.ptr + 8, y.ptr) != 0 and b"argument check" in lib.mk_last_error()
    assert lib.mk_spmv(op.handle, buf.ptr, y.ptr) == 0
    r = ctypes.c_double()
    assert lib.mk_dot(n, buf.ptr + 8, y.ptr, ctypes.byref(r)) != 0
    assert lib.mk_axpy(n, 1.0, buf.ptr + 8, y.ptr) != 0
    p = _lib.MkParams()
    p.struct_size = ctypes.sizeof(_lib.MkParams)
    p.kind = _lib.MK_CG
    p.matvec_max = 5
    h = ctypes.c_void_p()
    _lib.check(lib.mk_solver_create(op.handle, ctypes.byref(p), ctypes.byref(h)))
    assert lib.mk_solver_setup(h, buf.ptr + 8, None) != 0
    assert lib.mk_solver_set_precon_diag(h, buf.ptr + 8) != 0
    assert lib.mk_solver_setup(h, buf.ptr, None) == 0
    lib.mk_solver_destroy(h)


def test_transpose_of_a_composed_operator_keeps_its_row_program():
    """(alpha A - sigma I)^T carries the row program in C; the Python wrapper must carry the step count too,
    so that a further composition beyond MK_ROWPROG_MAX falls back to a host closure instead of raising."""
    from pykrylov_amd import CsrOperator, IdentityOperator
    rng = np.random.default_rng(4)
    A = csr_ref.from_coo(rng.integers(0, 300, 2000), rng.integers(0, 300, 2000), rng.standard_normal(2000), (300, 300))
    op = op_from(A)
    c = 2.0 * (op - 1.5 * IdentityOperator(300))                   # two steps
    ct = c.T
    assert isinstance(ct, CsrOperator) and len(ct._steps) == 2
    x = rng.standard_normal(300)
    assert np.array_equal(ct * x, 2.0 * (A.rmatvec(x) - 1.5 * x))
    d = 0.5 * (ct + 0.25 * IdentityOperator(300))                  # four steps: still a device operator
    assert isinstance(d, CsrOperator)
    e = 3.0 * d                                                    # five: host closure, same result
    assert not isinstance(e, CsrOperator)
    assert np.array_equal(e * x, 3.0 * (0.5 * ((2.0 * (A.rmatvec(x) - 1.5 * x)) + 0.25 * x)))


@pytest.mark.parametrize("dtype", [np.int32, np.int64])
def test_out_of_range_columns_are_refused(dtype):
    """The range check of a matrix built from caller arrays (on the device for int32 indices, before narrowing for
    wider ones): a column outside x must never reach a kernel."""
    from pykrylov_amd import CsrOperator
    indptr = np.array([0, 2, 3], dtype=dtype)
    data = np.ones(3)
    for bad in ([0, 3, 1], [-1, 1, 2]):
        with pytest.raises(ValueError):
            CsrOperator(indptr, np.array(bad, dtype=dtype), data, (2, 3))
    op = CsrOperator(indptr, np.array([0, 2, 1], dtype=dtype), data, (2, 3))
    assert np.array_equal(op * np.array([1.0, 2.0, 4.0]), [5.0, 2.0])
    op.free()


def test_large_host_transfers_take_the_staged_path_and_keep_every_byte():
    """Transfers above 64 MiB are cut into pieces that several host threads stage through pinned buffers
    (mk_core.hip): contents and order must survive, for sizes that are not multiples of a piece."""
    from pykrylov_amd import _lib
    lib = _lib.init()
    rng = np.random.default_rng(8)
    for n in (9_000_001, 8_388_608 + 3):
        x = rng.standard_normal(n)
        d = _lib.DeviceArray.from_numpy(x)
        y = d.to_numpy()
        assert np.array_equal(x.view(np.int64), y.view(np.int64))
        del d
