"""CPU oracle: CSR container, left-to-right SpMV and the synthetic test matrices.

TEST INFRASTRUCTURE ONLY -- nothing under ``pykrylov_amd/`` imports this.

The reference ships no CSR product; the oracle operator is
``LinearOperator(n, n, matvec=lambda v: A_csr @ v)`` with SciPy's ``csr_matvec``
(per row, left to right, one rounding per multiply and per add).  ``matvec``
below is that algorithm restated twice: a C loop (``csr_ref.c``, used when
``libcsr_ref.so`` has been built) and a pure-NumPy version that walks the k-th
stored entry of every row at once, which performs exactly the same additions in
exactly the same order.  Both are pinned against SciPy-produced vectors in
``tests/golden`` by ``tests/test_oracle_golden.py``.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _clib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libcsr_ref.so")
        if os.path.exists(path):
            lib = ctypes.CDLL(path)
            lib.ref_csr_matvec.argtypes = [ctypes.c_int64] + [ctypes.c_void_p] * 5
            lib.ref_csr_matvec.restype = None
            lib.ref_csr_transpose.argtypes = [ctypes.c_int64, ctypes.c_int64] + [ctypes.c_void_p] * 6
            lib.ref_csr_transpose.restype = None
            lib.ref_poisson3d_indptr.argtypes = [ctypes.c_int64] * 5 + [ctypes.c_void_p]
            lib.ref_poisson3d_indptr.restype = ctypes.c_int64
            lib.ref_poisson3d_varcoef_fill.argtypes = ([ctypes.c_int64] * 3 + [ctypes.c_uint64] + [ctypes.c_int64] * 2
                                                       + [ctypes.c_void_p] * 3)
            lib.ref_poisson3d_varcoef_fill.restype = None
            lib.ref_poisson3d_const_fill.argtypes = [ctypes.c_int64] * 5 + [ctypes.c_void_p] * 3
            lib.ref_poisson3d_const_fill.restype = None
            _LIB = lib
        else:
            _LIB = False
    return _LIB


class RefCsr(object):
    """Canonical CSR (sorted columns, no duplicates, int32 indices, float64 data)."""

    def __init__(self, indptr, indices, data, shape):
        self.indptr = np.ascontiguousarray(indptr, dtype=np.int32)
        self.indices = np.ascontiguousarray(indices, dtype=np.int32)
        self.data = np.ascontiguousarray(data, dtype=np.float64)
        self.shape = (int(shape[0]), int(shape[1]))
        self._T = None

    @property
    def nnz(self):
        return int(self.indptr[-1])

    # y = A x, per row left to right (scipy csr_matvec order)
    def matvec(self, x, force_numpy=False):
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.shape == (self.shape[1],)
        lib = _clib()
        y = np.zeros(self.shape[0])
        if lib and not force_numpy:
            lib.ref_csr_matvec(self.shape[0], self.indptr.ctypes.data, self.indices.ctypes.data,
                               self.data.ctypes.data, x.ctypes.data, y.ctypes.data)
            return y
        start = self.indptr[:-1].astype(np.int64)
        length = np.diff(self.indptr).astype(np.int64)
        rows = np.arange(self.shape[0])
        k = 0
        while True:
            live = length > k
            if not live.any():
                break
            pos = start[live] + k
            y[rows[live]] = y[rows[live]] + self.data[pos] * x[self.indices[pos]]
            k += 1
        return y

    __call__ = matvec

    def transpose(self):
        if self._T is None:
            m, n = self.shape
            lib = _clib()
            if lib:
                tp = np.zeros(n + 1, dtype=np.int32)
                ti = np.zeros(self.nnz, dtype=np.int32)
                td = np.zeros(self.nnz)
                lib.ref_csr_transpose(m, n, self.indptr.ctypes.data, self.indices.ctypes.data,
                                      self.data.ctypes.data, tp.ctypes.data, ti.ctypes.data, td.ctypes.data)
            else:
                rows = np.repeat(np.arange(m, dtype=np.int32), np.diff(self.indptr))
                order = np.argsort(self.indices, kind="stable")
                ti = rows[order]
                td = self.data[order]
                tp = np.zeros(n + 1, dtype=np.int32)
                tp[1:] = np.cumsum(np.bincount(self.indices, minlength=n))
            self._T = RefCsr(tp, ti, td, (n, m))
        return self._T

    def rmatvec(self, u):
        return self.transpose().matvec(u)

    def to_dense(self):
        D = np.zeros(self.shape)
        rows = np.repeat(np.arange(self.shape[0]), np.diff(self.indptr))
        D[rows, self.indices] = self.data
        return D


def from_coo(rows, cols, vals, shape):
    """Sort by (row, col) and sum duplicates in input order (SciPy tocsr + sum_duplicates)."""
    rows = np.asarray(rows, dtype=np.int64)
    cols = np.asarray(cols, dtype=np.int64)
    vals = np.asarray(vals, dtype=np.float64)
    key = rows * shape[1] + cols
    order = np.argsort(key, kind="stable")
    key, vals = key[order], vals[order]
    first = np.ones(len(key), dtype=bool)
    first[1:] = key[1:] != key[:-1]
    ukey = key[first]
    seg = np.cumsum(first) - 1
    data = np.zeros(len(ukey))
    # sequential accumulation inside each duplicate group, in input order
    np.add.at(data, seg, vals)
    indptr = np.zeros(shape[0] + 1, dtype=np.int64)
    indptr[1:] = np.cumsum(np.bincount(ukey // shape[1], minlength=shape[0]))
    return RefCsr(indptr, ukey % shape[1], data, shape)


# ----------------------------------------------------------------------------- #
# synthetic matrices of BASELINE.md section 3 (independent of the product's builders)
# ----------------------------------------------------------------------------- #
def poisson1d(n):
    """tridiag(-1, 2, -1)   (reference gallery/gallery.py:3-8 as a matrix)."""
    i = np.arange(n)
    rows = np.concatenate([i[1:], i, i[:-1]])
    cols = np.concatenate([i[:-1], i, i[1:]])
    vals = np.concatenate([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)])
    return from_coo(rows, cols, vals, (n, n))


def poisson2d(m):
    """5-point Laplacian on an m x m grid, Dirichlet, row-major (gallery/gallery.py:10-29)."""
    n = m * m
    idx = np.arange(n)
    gx, gy = idx % m, idx // m
    parts = [(idx, idx, 4.0 * np.ones(n))]
    for ok, off in ((gy > 0, -m), (gx > 0, -1), (gx < m - 1, 1), (gy < m - 1, m)):
        parts.append((idx[ok], idx[ok] + off, -np.ones(int(ok.sum()))))
    return from_coo(*[np.concatenate(c) for c in zip(*parts)], shape=(n, n))


def poisson3d(mx, my=None, mz=None):
    """7-point Laplacian, x fastest then y then z."""
    my = mx if my is None else my
    mz = mx if mz is None else mz
    n = mx * my * mz
    idx = np.arange(n)
    gx, gy, gz = idx % mx, (idx // mx) % my, idx // (mx * my)
    parts = [(idx, idx, 6.0 * np.ones(n))]
    for ok, off in ((gz > 0, -mx * my), (gy > 0, -mx), (gx > 0, -1),
                    (gx < mx - 1, 1), (gy < my - 1, mx), (gz < mz - 1, mx * my)):
        parts.append((idx[ok], idx[ok] + off, -np.ones(int(ok.sum()))))
    return from_coo(*[np.concatenate(c) for c in zip(*parts)], shape=(n, n))


def cell_field(cells, seed):
    """k(c) = 0.5 + u(c), u a 53-bit uniform from a splitmix64 hash of the cell number (twin of mk_cell_field,
    pykrylov_amd/csrc/mk_core.hip)."""
    with np.errstate(over="ignore"):
        z = (np.asarray(cells).astype(np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return 0.5 + (z >> np.uint64(11)).astype(np.float64) * 2.0 ** -53


def poisson3d_varcoef(mx, my=None, mz=None, seed=7, rows=None):
    """Variable-coefficient 7-point operator -div(k grad u), Dirichlet: the sparsity of `poisson3d`; off-diagonal
    entries are minus the harmonic means ((2 ka) kb) / (ka + kb) of the two cells' coefficients, the diagonal the
    left-to-right sum over the directions (-z, -y, -x, +x, +y, +z) of that mean, or of k(c) where the neighbour is
    missing.  Twin of gen_poisson3d_varcoef (bit-identical arrays).  `rows=(a, b)`: only the rows [a, b) of the
    matrix, as a (b - a) x n RefCsr with global column ids (how the 512^3 matrix is checked slab by slab)."""
    my = mx if my is None else my
    mz = mx if mz is None else mz
    n = mx * my * mz
    a, b = (0, n) if rows is None else (int(rows[0]), int(rows[1]))
    assert 0 <= a <= b <= n
    idx = np.arange(a, b, dtype=np.int64)
    gx, gy, gz = idx % mx, (idx // mx) % my, idx // (mx * my)
    kc = cell_field(idx, seed)
    dirs = ((gz > 0, -mx * my), (gy > 0, -mx), (gx > 0, -1), (gx < mx - 1, 1), (gy < my - 1, mx), (gz < mz - 1, mx * my))
    terms = []
    for ok, off in dirs:
        kb = cell_field(np.where(ok, idx + off, idx), seed)
        terms.append(np.where(ok, ((2.0 * kc) * kb) / (kc + kb), kc))
    diag = ((((terms[0] + terms[1]) + terms[2]) + terms[3]) + terms[4]) + terms[5]
    parts = [(idx - a, idx, diag)]
    for (ok, off), t in zip(dirs, terms):
        parts.append((idx[ok] - a, idx[ok] + off, -t[ok]))
    return from_coo(*[np.concatenate(c) for c in zip(*parts)], shape=(b - a, n))


def poisson3d_varcoef_c(mx, my=None, mz=None, seed=7, rows=None):
    """The same matrix (or row block) written straight into CSR arrays by the C twin (csr_ref.c
    ref_poisson3d_varcoef_fill; OpenMP over rows): no COO triples, no sort -- what lets bench.py's CPU baseline hold
    the full 512^3 problem.  Bit-identical to :func:`poisson3d_varcoef` (tests/test_oracle_golden.py)."""
    lib = _clib()
    assert lib, "oracle/libcsr_ref.so is not built (make -C oracle)"
    my = mx if my is None else my
    mz = mx if mz is None else mz
    n = mx * my * mz
    a, b = (0, n) if rows is None else (int(rows[0]), int(rows[1]))
    indptr = np.empty(b - a + 1, dtype=np.int32)
    nnz = lib.ref_poisson3d_indptr(mx, my, mz, a, b, indptr.ctypes.data)
    assert nnz < 2 ** 31
    indices = np.empty(nnz, dtype=np.int32)
    data = np.empty(nnz, dtype=np.float64)
    lib.ref_poisson3d_varcoef_fill(mx, my, mz, seed, a, b, indptr.ctypes.data, indices.ctypes.data, data.ctypes.data)
    A = RefCsr.__new__(RefCsr)
    A.indptr, A.indices, A.data, A.shape, A._T = indptr, indices, data, (b - a, n), None
    return A


def poisson3d_c(mx, my=None, mz=None, rows=None):
    """:func:`poisson3d` (constant coefficients) written straight into CSR arrays by the C twin (csr_ref.c
    ref_poisson3d_const_fill): what lets bench.py's CPU baseline hold the literal BASELINE configs[4] matrix at 512^3.
    Bit-identical to :func:`poisson3d` (tests/test_oracle_golden.py)."""
    lib = _clib()
    assert lib, "oracle/libcsr_ref.so is not built (make -C oracle)"
    my = mx if my is None else my
    mz = mx if mz is None else mz
    n = mx * my * mz
    a, b = (0, n) if rows is None else (int(rows[0]), int(rows[1]))
    indptr = np.empty(b - a + 1, dtype=np.int32)
    nnz = lib.ref_poisson3d_indptr(mx, my, mz, a, b, indptr.ctypes.data)
    assert nnz < 2 ** 31
    indices = np.empty(nnz, dtype=np.int32)
    data = np.empty(nnz, dtype=np.float64)
    lib.ref_poisson3d_const_fill(mx, my, mz, a, b, indptr.ctypes.data, indices.ctypes.data, data.ctypes.data)
    A = RefCsr.__new__(RefCsr)
    A.indptr, A.indices, A.data, A.shape, A._T = indptr, indices, data, (b - a, n), None
    return A


def stencil27(mx, my=None, mz=None, seed=0):
    """27-point box stencil, columns ascending.  seed == 0: -1 off the diagonal, 26 on it; otherwise entry (a, b) is
    minus the harmonic mean of the two cells' coefficients and the diagonal the left-to-right sum, over the 26
    directions in column order, of that mean -- or of k(a) where the neighbour is missing.  Twin of gen_stencil27
    (pykrylov_amd/csrc/mk_core.hip; bit-identical arrays)."""
    my = mx if my is None else my
    mz = mx if mz is None else mz
    n = mx * my * mz
    idx = np.arange(n, dtype=np.int64)
    gx, gy, gz = idx % mx, (idx // mx) % my, idx // (mx * my)
    kc = cell_field(idx, seed) if seed else np.ones(n)
    diag = np.zeros(n)
    parts = []
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if dz == 0 and dy == 0 and dx == 0:
                    continue
                ok = ((gz + dz >= 0) & (gz + dz < mz) & (gy + dy >= 0) & (gy + dy < my) & (gx + dx >= 0) & (gx + dx < mx))
                off = dz * mx * my + dy * mx + dx
                if seed:
                    kb = cell_field(np.where(ok, idx + off, idx), seed)
                    h = np.where(ok, ((2.0 * kc) * kb) / (kc + kb), kc)
                else:
                    h = kc
                diag = diag + h
                parts.append((idx[ok], idx[ok] + off, -h[ok]))
    parts.append((idx, idx, diag))
    return from_coo(*[np.concatenate(c) for c in zip(*parts)], shape=(n, n))


def random_diagdom(n, seed=1, k=4):
    """k random off-diagonals per row (duplicates summed,
    accidental diagonal hits dropped), diagonal = sum|offdiag| + 1."""
    rng = np.random.default_rng(seed)
    rows = np.repeat(np.arange(n), k)
    cols = rng.integers(0, n, size=n * k)
    vals = rng.standard_normal(n * k)
    off = from_coo(rows, cols, vals, (n, n))
    r = np.repeat(np.arange(n), np.diff(off.indptr))
    keep = (off.indices != r) & (off.data != 0.0)
    r, c, v = r[keep], off.indices[keep], off.data[keep]
    diag = np.zeros(n)
    np.add.at(diag, r, np.abs(v))
    diag += 1.0
    return from_coo(np.concatenate([r, np.arange(n)]), np.concatenate([c, np.arange(n)]),
                    np.concatenate([v, diag]), (n, n))


def read_matrix_market(path):
    """Minimal coordinate-format reader (real/integer/pattern; general/symmetric)."""
    with open(path) as fh:
        header = fh.readline().lower().split()
        assert header[0] == "%%matrixmarket" and header[2] == "coordinate"
        field, symm = header[3], header[4]
        line = fh.readline()
        while line.startswith("%"):
            line = fh.readline()
        m, n, nz = (int(t) for t in line.split())
        body = np.loadtxt(fh, ndmin=2)
    rows = body[:, 0].astype(np.int64) - 1
    cols = body[:, 1].astype(np.int64) - 1
    vals = np.ones(nz) if field == "pattern" else body[:, 2].astype(np.float64)
    if symm == "symmetric":
        off = rows != cols
        rows, cols, vals = (np.concatenate([rows, cols[off]]), np.concatenate([cols, rows[off]]),
                            np.concatenate([vals, vals[off]]))
    return from_coo(rows, cols, vals, (m, n))


class Composed(object):
    """The reference's operator algebra on top of a RefCsr, restated as closures: `fn(y, x)` receives the product
    y = A x and the operand x and returns what the reference's composite operator returns (linop.py:307-330
    `alpha * (op * x)`, :375-426 `(op * x) +/- (other * x)`).  Test infrastructure only."""

    def __init__(self, A, fn):
        self.A, self.fn, self.shape = A, fn, A.shape

    def matvec(self, x):
        return self.fn(self.A.matvec(x), x)

    __call__ = matvec
