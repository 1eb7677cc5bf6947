"""Test operators: the Poisson matrices of the reference gallery, as device CSR operators.

The reference gallery holds two matrix-free products (pykrylov/gallery/gallery.py:3-29).  Here
the same matrices -- tridiag(-1, 2, -1) and the 5-point Laplacian with Dirichlet boundary on an
m x m row-major grid -- are built as canonical CSR, either on the host (NumPy, small sizes,
bit-checked against fixtures) or directly in HBM by generator kernels (the 512^3 7-point
matrix is 11 GB and never exists on the host).
"""
import ctypes
from math import sqrt

import numpy as np

from . import _lib
from .linop import CsrOperator
from .sparse import coo_to_csr


# -- matrix-free products kept for API parity (gallery.py:3-29), vectorised -----------------
def Poisson1dMatvec(x):
    "y = tridiag(-1, 2, -1) x"
    y = 2 * x
    y[:-1] -= x[1:]
    y[1:] -= x[:-1]
    return y


def Poisson2dMatvec(x):
    "y = (5-point Laplacian on a sqrt(len(x))-square grid) x"
    n = int(sqrt(x.shape[0]))
    g = x.reshape(n, n)
    y = 4 * g
    y[1:, :] -= g[:-1, :]
    y[:-1, :] -= g[1:, :]
    y[:, :-1] -= g[:, 1:]
    y[:, 1:] -= g[:, :-1]
    return y.reshape(-1)


# -- host CSR builders ---------------------------------------------------------------------
def _stencil_csr(dims, diag):
    """(2*len(dims)+1)-point Laplacian on a grid with extents `dims` (first index fastest)."""
    n = int(np.prod(dims))
    idx = np.arange(n, dtype=np.int64)
    rows, cols, vals = [idx], [idx], [np.full(n, float(diag))]
    stride = 1
    for ext in dims:
        coord = (idx // stride) % ext
        for ok, off in ((coord > 0, -stride), (coord < ext - 1, stride)):
            rows.append(idx[ok])
            cols.append(idx[ok] + off)
            vals.append(np.full(int(ok.sum()), -1.0))
        stride *= ext
    return coo_to_csr(np.concatenate(rows), np.concatenate(cols), np.concatenate(vals), (n, n)), n


def poisson1d_csr(n):
    (indptr, indices, data), n = _stencil_csr((n,), 2.0)
    return indptr, indices, data, (n, n)


def poisson2d_csr(m):
    (indptr, indices, data), n = _stencil_csr((m, m), 4.0)
    return indptr, indices, data, (n, n)


def poisson3d_csr(nx, ny=None, nz=None):
    ny = nx if ny is None else ny
    nz = nx if nz is None else nz
    (indptr, indices, data), n = _stencil_csr((nx, ny, nz), 6.0)
    return indptr, indices, data, (n, n)


def random_diagdom_csr(n, seed=1, k=4):
    """Nonsymmetric test matrix of BASELINE.md section 3 item 3: `k` off-diagonal entries per row
    at ``default_rng(seed).integers`` columns with ``standard_normal`` values (duplicates summed,
    diagonal hits dropped), diagonal = (row sum of |offdiag| in column order) + 1."""
    rng = np.random.default_rng(seed)
    rows = np.repeat(np.arange(n, dtype=np.int64), k)
    cols = rng.integers(0, n, size=n * k)
    vals = rng.standard_normal(n * k)
    indptr, indices, data = coo_to_csr(rows, cols, vals, (n, n))
    r = np.repeat(np.arange(n, dtype=np.int64), np.diff(indptr))
    keep = (indices != r) & (data != 0.0)
    r, c, v = r[keep], indices[keep].astype(np.int64), data[keep]
    diag = np.zeros(n)
    np.add.at(diag, r, np.abs(v))          # sequential, ascending column order within a row
    diag += 1.0
    d = np.arange(n, dtype=np.int64)
    indptr, indices, data = coo_to_csr(np.concatenate([r, d]), np.concatenate([c, d]),
                                       np.concatenate([v, diag]), (n, n))
    return indptr, indices, data, (n, n)


# -- operators -----------------------------------------------------------------------------
def poisson1d(n):
    indptr, indices, data, shape = poisson1d_csr(n)
    return CsrOperator(indptr, indices, data, shape, symmetric=True)


def poisson2d(m, on_device=True):
    """5-point Laplacian operator; with `on_device` the CSR arrays are generated in HBM."""
    if not on_device:
        indptr, indices, data, shape = poisson2d_csr(m)
        return CsrOperator(indptr, indices, data, shape, symmetric=True)
    lib = _lib.init()
    h = ctypes.c_void_p()
    _lib.check(lib.mk_csr_poisson2d(m, 0, m * m, ctypes.byref(h)))
    return CsrOperator.from_handle(h.value, symmetric=True)


def poisson3d(nx, ny=None, nz=None, on_device=True):
    """7-point Laplacian operator (x fastest); with `on_device` generated in HBM."""
    ny = nx if ny is None else ny
    nz = nx if nz is None else nz
    if not on_device:
        indptr, indices, data, shape = poisson3d_csr(nx, ny, nz)
        return CsrOperator(indptr, indices, data, shape, symmetric=True)
    lib = _lib.init()
    h = ctypes.c_void_p()
    _lib.check(lib.mk_csr_poisson3d(nx, ny, nz, 0, nx * ny * nz, ctypes.byref(h)))
    return CsrOperator.from_handle(h.value, symmetric=True)


def poisson3d_varcoef(nx, ny=None, nz=None, seed=7):
    """-div(k grad u) on the 7-point grid with a hashed positive cell field k (harmonic-mean face coefficients,
    Dirichlet): the sparsity of `poisson3d`, practically all values distinct.  Generated in HBM."""
    ny = nx if ny is None else ny
    nz = nx if nz is None else nz
    lib = _lib.init()
    h = ctypes.c_void_p()
    _lib.check(lib.mk_csr_poisson3d_varcoef(nx, ny, nz, seed, 0, nx * ny * nz, ctypes.byref(h)))
    return CsrOperator.from_handle(h.value, symmetric=True)


def stencil27(nx, ny=None, nz=None, seed=0):
    """27-point box stencil (HPCG's sparsity), generated in HBM.  seed == 0: -1 / 26 (HPCG's values); otherwise
    variable coefficients from the hashed cell field of `poisson3d_varcoef`."""
    ny = nx if ny is None else ny
    nz = nx if nz is None else nz
    lib = _lib.init()
    h = ctypes.c_void_p()
    _lib.check(lib.mk_csr_stencil27(nx, ny, nz, seed, 0, nx * ny * nz, ctypes.byref(h)))
    return CsrOperator.from_handle(h.value, symmetric=True)


def random_diagdom(n, seed=1, k=4):
    indptr, indices, data, shape = random_diagdom_csr(n, seed, k)
    return CsrOperator(indptr, indices, data, shape)
