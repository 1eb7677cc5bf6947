"""Small helpers of the solver path (reference pykrylov/tools/utils.py)."""
import ctypes

import numpy as np

from . import _lib


def machine_epsilon():
    "Double-precision machine epsilon (utils.py:7-9)."
    return np.finfo(np.double).eps


def check_symmetric(op, repeats=10):
    """Cheap randomized symmetry test (utils.py:63-85): for `repeats` random x checks
    ``<Ax, Ax> == <x, A(Ax)>`` up to ``(s + eps) * eps**(1/3)``.

    Like the reference it reseeds the global NumPy RNG with 1 and draws the vectors with
    ``np.random.random`` (so traces with ``check=True`` can be compared).  For a
    :class:`CsrOperator` both products and both inner products run on the GPU; the products
    are counted in ``op.nMatvec`` as the reference does.
    """
    from .linop import CsrOperator
    op = getattr(op, 'host_op', op)                           # (solver-side shell of a matrix-free operator)
    if getattr(op, 'local_size', None) is not None:
        return _check_symmetric_partitioned(op, repeats)
    m, n = op.shape
    if m != n:
        return False
    eps = machine_epsilon()
    np.random.seed(1)
    on_device = isinstance(op, CsrOperator)
    if on_device:
        lib = _lib.init()
        dx, dw, dr = (_lib.DeviceArray(n, zero=False) for _ in range(3))
        s, t = ctypes.c_double(), ctypes.c_double()
    for _ in range(repeats):
        x = np.random.random(n)
        if on_device:
            dx.upload(x)
            op.spmv_device(dx.ptr, dw.ptr)
            op.spmv_device(dw.ptr, dr.ptr)
            _lib.check(lib.mk_dot(n, dw.ptr, dw.ptr, ctypes.byref(s)))
            _lib.check(lib.mk_dot(n, dx.ptr, dr.ptr, ctypes.byref(t)))
            sv, tv = s.value, t.value
        else:                       # an operator defined by host callables lives on the host by definition
            w = op * x
            r = op * w
            sv, tv = np.dot(w, w), np.dot(x, r)
        if abs(sv - tv) > (sv + eps) * eps ** (1.0 / 3):
            return False
    return True


def _check_symmetric_partitioned(op, repeats):
    """The same test on a row-partitioned operator (pykrylov_amd.dist): every rank draws the same global random
    vector and keeps its slice, products are preceded by the operator's exchange, inner products are summed over
    the ranks -- so all ranks reach the same verdict.  Collective: call it on every rank."""
    lib = _lib.init()
    c0, c1 = op.row_range
    n_local, n_ext, n_global = c1 - c0, op.shape[1], op.global_size
    eps = machine_epsilon()
    np.random.seed(1)
    dx, dw = _lib.DeviceArray(n_ext), _lib.DeviceArray(n_ext)
    dr = _lib.DeviceArray(n_local, zero=False)
    pair = (ctypes.c_double * 2)()
    s, t = ctypes.c_double(), ctypes.c_double()
    ok = True
    for _ in range(repeats):
        x = np.random.random(n_global)[c0:c1]
        _lib.check(lib.mk_memcpy_h2d(dx.ptr, np.ascontiguousarray(x).ctypes.data, 8 * n_local))
        _lib.check(lib.mk_exchange(op.handle, dx.ptr))
        op.spmv_device(dx.ptr, dw.ptr)                        # w = A x   (local rows, written into [0, n_local))
        _lib.check(lib.mk_exchange(op.handle, dw.ptr))
        op.spmv_device(dw.ptr, dr.ptr)                        # r = A w
        _lib.check(lib.mk_dot(n_local, dw.ptr, dw.ptr, ctypes.byref(s)))
        _lib.check(lib.mk_dot(n_local, dx.ptr, dr.ptr, ctypes.byref(t)))
        pair[0], pair[1] = s.value, t.value
        _lib.check(lib.mk_comm_allreduce_host(pair, 2))
        if abs(pair[0] - pair[1]) > (pair[0] + eps) * eps ** (1.0 / 3):
            ok = False                                        # keep going: the collectives must stay matched
    for b in (dx, dw, dr):
        b.free()
    return ok


def block_jacobi(op, block_size):
    """Block-Jacobi preconditioner of a device matrix as a DEVICE operator: the diagonal blocks of `op` (size
    `block_size`, the last one possibly smaller) are inverted once on the host (NumPy, batched) and the inverses are
    stored as a block-diagonal :class:`CsrOperator`.  Passed as ``precon=`` it is applied inside the device loop as a
    product (``precon * r``, generic/generic.py:76; mk_solver_set_precon_csr) -- no host round trip per iteration,
    unlike a general callable preconditioner.  `block_size` = 1 gives the Jacobi (diagonal) preconditioner as a matrix.
    """
    from .linop import CsrOperator
    indptr, indices, data = op.to_csr_arrays()
    n = op.shape[0]
    nloc = getattr(op, 'local_size', None)
    if nloc is not None:
        # row-partitioned operator (pykrylov_amd.dist): the RANK-LOCAL preconditioner -- blocks of this rank's diagonal
        # block only (its columns are numbered [own | received], so owned columns are those below the local size);
        # blocks never straddle ranks, the result has no exchange plan and is applied without communication
        n = int(nloc)
        if getattr(op, 'exchange_mode', 0) == 1:              # all-gather layout: column = n_local + GLOBAL column
            c0 = int(op.row_range[0])
            indices = indices.astype(np.int64) - n - c0       # owned columns -> 0 .. n-1, everything else outside
            own = (indices >= 0) & (indices < n)
        else:
            own = indices < n
        rows_all = np.repeat(np.arange(n, dtype=np.int64), np.diff(indptr))
        counts = np.bincount(rows_all[own], minlength=n)
        indices, data = indices[own], data[own]
        indptr = np.concatenate([[0], np.cumsum(counts)])
    elif op.shape[0] != op.shape[1]:
        raise ValueError('block_jacobi needs a square operator')
    bs = int(block_size)
    if bs < 1:
        raise ValueError('block_size must be positive')
    nb = (n + bs - 1) // bs
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(indptr))
    cols = indices.astype(np.int64)
    inblk = (rows // bs) == (cols // bs)
    blocks = np.zeros((nb, bs, bs))
    blocks[rows[inblk] // bs, rows[inblk] % bs, cols[inblk] % bs] = data[inblk]
    tail = n - (nb - 1) * bs
    if tail < bs:                                            # pad the last block with an identity corner
        k = np.arange(tail, bs)
        blocks[nb - 1, k, k] = 1.0
    inv = np.linalg.inv(blocks)
    # block-diagonal CSR of the inverses (dense blocks; the padding of the last block is dropped)
    r = np.repeat(np.arange(nb * bs, dtype=np.int64), bs)
    c = (r // bs) * bs + np.tile(np.arange(bs, dtype=np.int64), nb * bs)
    v = inv.reshape(-1)
    keep = (r < n) & (c < n)
    r, c, v = r[keep], c[keep], v[keep]
    ip = np.zeros(n + 1, dtype=np.int64)
    ip[1:] = np.cumsum(np.bincount(r, minlength=n))
    return CsrOperator(ip, c, v, (n, n), symmetric=bool(getattr(op, 'symmetric', False)))
