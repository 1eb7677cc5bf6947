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
