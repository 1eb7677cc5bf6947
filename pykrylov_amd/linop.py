"""Operator protocol of pykrylov, with a device-resident CSR operator behind it.

Host-side mirror of the reference interface (reference ``pykrylov/linop/linop.py``):
``BaseLinearOperator`` (:14-104), ``LinearOperator`` (:107-452) and the helper
constructors (:455-754) keep their names, argument meaning and error behaviour
so that scripts written against pykrylov keep working.  What is new is
:class:`CsrOperator`: a ``LinearOperator`` whose matrix lives in HBM and whose
product is the hand-written gfx950 CSR-stream kernel ``mk_spmv`` (the reference
has no CSR product of its own -- SURVEY.md F2).

``op * ndarray -> new ndarray`` is preserved (the "plumbing" path: upload, SpMV on
the GPU, download), so reference-style Python loops, ``check_symmetric`` and the
operator algebra run unmodified on top of a ``CsrOperator``.  The solver classes
of this package do not use that path: they hand the device handle to the
device-resident loops in ``libmikrylov.so``.
"""
import ctypes
import logging

import numpy as np

from . import _lib

__docformat__ = 'restructuredtext'

null_log = logging.getLogger('linop')
null_log.setLevel(logging.INFO)
null_log.addHandler(logging.NullHandler())

_INT_KINDS, _REAL_KINDS, _CPLX_KINDS = "iub", "f", "c"


def _kind(dtype):
    try:
        return np.dtype(dtype).kind
    except TypeError:
        return None


def _is_complex(dtype):
    return _kind(dtype) in _CPLX_KINDS


class ShapeError(Exception):
    """Raised when operators or operands have incompatible shapes (linop.py:626-635)."""

    def __init__(self, value):
        super(ShapeError, self).__init__(value)
        self.value = value

    def __str__(self):
        return repr(self.value)


def _is_real_scalar(x):
    return isinstance(x, (int, float, np.integer, np.floating)) and not isinstance(x, (bool, np.bool_))


class BaseLinearOperator(object):
    """Shape / dtype / symmetry metadata and the product counter (linop.py:14-104)."""

    def __init__(self, nargin, nargout, symmetric=False, hermitian=False, **kwargs):
        self._nargin = nargin
        self._nargout = nargout
        self._symmetric = symmetric
        self._hermitian = hermitian
        self._shape = (nargout, nargin)
        self._dtype = kwargs.get('dtype', np.float64)
        self._nMatvec = 0
        self.logger = kwargs.get('logger', null_log)
        self.logger.info('New linear operator with shape ' + str(self.shape))

    nargin = property(lambda self: self._nargin, doc="Size of an input vector.")
    nargout = property(lambda self: self._nargout, doc="Size of an output vector.")
    symmetric = property(lambda self: self._symmetric)
    hermitian = property(lambda self: self._hermitian)
    shape = property(lambda self: self._shape)
    nMatvec = property(lambda self: self._nMatvec, doc="Products with vectors computed so far.")

    @property
    def dtype(self):
        return self._dtype

    @dtype.setter
    def dtype(self, value):
        if _kind(value) in _INT_KINDS + _REAL_KINDS + _CPLX_KINDS:
            self._dtype = value
        else:
            raise TypeError('Not a Numpy type')

    def reset_counters(self):
        self._nMatvec = 0

    def __call__(self, *args, **kwargs):
        return self.__mul__(*args, **kwargs)

    def __mul__(self, x):
        raise NotImplementedError('Please subclass to implement __mul__.')

    def __repr__(self):
        s = 'Symmetric' if self.symmetric else 'Unsymmetric'
        if self.hermitian:
            s += ' Hermitian'
        s += ' <' + self.__class__.__name__ + '>'
        s += ' of type %s' % self.dtype
        s += ' with shape (%d,%d)' % (self.nargout, self.nargin)
        return s


class LinearOperator(BaseLinearOperator):
    """Operator defined by a ``matvec`` callable and optional transpose / adjoint
    callables (linop.py:107-452).  ``symmetric=True`` makes ``op.T is op``."""

    def __init__(self, nargin, nargout, matvec, matvec_transp=None, matvec_adj=None, **kwargs):
        transpose_of = kwargs.pop('transpose_of', None)
        adjoint_of = kwargs.pop('adjoint_of', None)
        conjugate_of = kwargs.pop('conjugate_of', None)
        super(LinearOperator, self).__init__(nargin, nargout, **kwargs)
        self._matvec_fn = matvec
        self._T = self._sibling(self.symmetric, transpose_of, matvec_transp, 'transpose_of',
                                dict(matvec_transp=matvec), kwargs)
        self._H = self._sibling(self.hermitian, adjoint_of, matvec_adj, 'adjoint_of',
                                dict(matvec_adj=matvec), kwargs)
        if not _is_complex(self.dtype):
            # real data: transpose and adjoint coincide (linop.py:126-131)
            if self._T is not None and self._H is None:
                self._H = self._T
            elif self._T is None and self._H is not None:
                self._T = self._H
        elif transpose_of is None and adjoint_of is None and conjugate_of is None:
            conj = self.conjugate()
            if self._T is not None:
                self._T._H = conj
                if self._H is None and conj is not None:
                    self._H = conj.T
            if self._H is not None:
                self._H._T = conj
                if self._T is None and conj is not None:
                    self._T = conj.H

    def _sibling(self, is_self, given, fn, back_kw, fn_kw, kwargs):
        if is_self:
            return self
        if given is not None:
            if not isinstance(given, BaseLinearOperator):
                raise ValueError('kwarg %s must be a BaseLinearOperator. Got %s' % (back_kw, str(given.__class__)))
            return given
        if fn is None:
            return None
        kw = dict(kwargs)
        kw.update(fn_kw)
        kw[back_kw] = self
        return LinearOperator(self.nargout, self.nargin, fn, **kw)

    T = property(lambda self: self._T, doc="The transpose operator.")
    H = property(lambda self: self._H, doc="The adjoint operator.")
    bar = property(lambda self: self.conjugate(), doc="The complex conjugate operator.")

    def conjugate(self):
        if not _is_complex(self.dtype):
            return self

        def conj_of(op):
            def mv(x):
                if not _is_complex(x.dtype):
                    return (op * x).conjugate()
                return (op * x.conjugate()).conjugate()
            return mv

        if self._H is not None:
            mv_t = self._H._matvec_fn
            mv_h = self._T._matvec_fn if self._T is not None else conj_of(self._H)
        elif self._T is not None:
            mv_h = self._T._matvec_fn
            mv_t = conj_of(self._T)
        else:
            mv_t = mv_h = None
        return LinearOperator(self.nargin, self.nargout, matvec=conj_of(self), matvec_transp=mv_t,
                              matvec_adj=mv_h, transpose_of=self._H, adjoint_of=self._T,
                              conjugate_of=self, dtype=self.dtype)

    def to_array(self):
        "Dense matrix of the operator, one column per product (linop.py:256-269)."
        n, m = self.shape
        dense = np.empty((n, m), dtype=self.dtype)
        e = np.zeros(m, dtype=self.dtype)
        for j in range(m):
            e[j] = 1
            dense[:, j] = self * e
            e[j] = 0
        return dense

    full = to_array

    def _matvec(self, x):
        """Shape-checked call of the user callable (linop.py:271-298)."""
        x = np.asanyarray(x)
        nargout, nargin = self.shape
        try:
            x = x.reshape(nargin)
        except ValueError:
            raise ValueError('input array size incompatible with operator dimensions')
        y = self._matvec_fn(x)
        try:
            y = y.reshape(nargout)
        except ValueError:
            raise ValueError('output array size incompatible with operator dimensions')
        return y

    def rmatvec(self, x):
        "SciPy-style product with the conjugate transpose (linop.py:300-305)."
        return self._H.__mul__(x)

    # -- the three meanings of `*` (linop.py:307-369) ---------------------------------
    # `_mk_term = (diag or None, scale or None)` marks an operator whose product is
    # `[scale *] ([diag *] x)`: IdentityOperator, DiagonalOperator and one real scalar multiple of either.  A
    # CsrOperator combined with such an operand stays a device operator (see CsrOperator._compose).
    _mk_term = None

    def _times_scalar(self, x):
        result_type = np.result_type(self.dtype, type(x))
        if x == 0:
            return ZeroOperator(self.nargin, self.nargout, dtype=result_type)
        res = self._times_scalar_host(x, result_type)
        t = self._mk_term
        if t is not None and t[1] is None and _is_real_scalar(x) and not _is_complex(result_type):
            res._mk_term = (t[0], float(x))
        return res

    def _times_scalar_host(self, x, result_type):
        return LinearOperator(self.nargin, self.nargout,
                              symmetric=self.symmetric,
                              hermitian=(not _is_complex(result_type)) and self.hermitian,
                              matvec=lambda y: x * (self(y)),
                              matvec_transp=lambda y: x * (self.T(y)),
                              matvec_adj=lambda y: np.conjugate(x) * (self.H(y)),
                              dtype=result_type)

    def _times_linop(self, op):
        if self.nargin != op.nargout:
            raise ShapeError('Cannot multiply operators together')
        return LinearOperator(op.nargin, self.nargout, symmetric=False, hermitian=False,
                              matvec=lambda x: self(op(x)),
                              matvec_transp=lambda x: op.T(self.T(x)),
                              matvec_adj=lambda x: op.H(self.H(x)),
                              dtype=np.result_type(self.dtype, op.dtype))

    def _times_vector(self, x):
        self._nMatvec += 1
        result_type = np.result_type(self.dtype, x.dtype)
        return self._matvec(x).astype(result_type)

    def __mul__(self, x):
        if np.isscalar(x):
            return self._times_scalar(x)
        if isinstance(x, BaseLinearOperator):
            return self._times_linop(x)
        if isinstance(x, np.ndarray):
            return self._times_vector(x)
        raise ValueError('Cannot multiply')

    def __rmul__(self, x):
        if np.isscalar(x):
            return self._times_scalar(x)
        if isinstance(x, BaseLinearOperator):
            return x._times_linop(self)
        raise ValueError('Cannot multiply')

    def _combine(self, other, f):
        if not isinstance(other, BaseLinearOperator):
            raise ValueError('Cannot add')
        if self.shape != other.shape:
            raise ShapeError('Cannot add')
        if self._mk_term is not None and isinstance(other, CsrOperator):
            dev = other._compose_with_term(self._mk_term, 'add' if f is np.add else 'rsub')   # D + A, D - A
            if dev is not None:
                return dev
        return LinearOperator(self.nargin, self.nargout,
                              symmetric=self.symmetric and other.symmetric,
                              hermitian=self.hermitian and other.hermitian,
                              matvec=lambda x: f(self(x), other(x)),
                              matvec_transp=lambda x: f(self.T(x), other.T(x)),
                              matvec_adj=lambda x: f(self.H(x), other.H(x)),
                              dtype=np.result_type(self.dtype, other.dtype))

    def __add__(self, other):
        return self._combine(other, np.add)

    def __sub__(self, other):
        return self._combine(other, np.subtract)

    def __neg__(self):
        return self * (-1)

    def __truediv__(self, other):
        if not np.isscalar(other):
            raise ValueError('Cannot divide')
        return self * (1. / other)

    __div__ = __truediv__

    def __pow__(self, other):
        if not isinstance(other, int):
            raise ValueError('Can only raise to integer power')
        if other < 0:
            raise ValueError('Can only raise to nonnegative power')
        if self.nargin != self.nargout:
            raise ShapeError('Can only raise square operators to a power')
        if other == 0:
            return IdentityOperator(self.nargin)
        if other == 1:
            return self
        return self * self ** (other - 1)


class IdentityOperator(LinearOperator):
    "Identity of size `nargin` (linop.py:455-470)."

    def __init__(self, nargin, **kwargs):
        kwargs.pop('symmetric', None)
        kwargs.pop('matvec', None)
        super(IdentityOperator, self).__init__(nargin, nargin, symmetric=True, matvec=lambda x: x, **kwargs)
        self._mk_term = (None, None)


class DiagonalOperator(LinearOperator):
    "Diagonal operator from a 1-D array (linop.py:473-516)."

    def __init__(self, diag, **kwargs):
        kwargs.pop('symmetric', None)
        kwargs.pop('matvec', None)
        kwargs.pop('dtype', None)
        diag = np.asarray(diag)
        if diag.ndim != 1:
            raise ValueError('Input must be 1-d array')
        self._diag = diag.copy()
        super(DiagonalOperator, self).__init__(diag.shape[0], diag.shape[0], symmetric=True,
                                               matvec=lambda x: diag * x, dtype=diag.dtype, **kwargs)
        if _kind(diag.dtype) in _INT_KINDS + _REAL_KINDS:
            self._mk_term = (self._diag, None)

    diag = property(lambda self: self._diag, doc="The diagonal as a Numpy array.")

    def __abs__(self):
        return DiagonalOperator(np.abs(self._diag))

    def _sqrt(self):
        if self.dtype not in (np.complex64, np.complex128) and np.any(self._diag < 0):
            raise ValueError('Math domain error')
        return DiagonalOperator(np.sqrt(self._diag))


class ZeroOperator(LinearOperator):
    "The zero operator of shape `nargout`-by-`nargin` (linop.py:519-557)."

    def __init__(self, nargin, nargout, **kwargs):
        for k in ('matvec', 'matvec_transp'):
            kwargs.pop(k, None)

        def matvec(x):
            if x.shape != (nargin,):
                raise ShapeError('Input has shape ' + str(x.shape) + ' instead of (%d,)' % nargin)
            return np.zeros(nargout, dtype=np.result_type(self.dtype, x.dtype))

        def matvec_transp(x):
            if x.shape != (nargout,):
                raise ShapeError('Input has shape ' + str(x.shape) + ' instead of (%d,)' % nargout)
            return np.zeros(nargin, dtype=np.result_type(self.dtype, x.dtype))

        super(ZeroOperator, self).__init__(nargin, nargout, matvec=matvec, matvec_transp=matvec_transp, **kwargs)

    def __abs__(self):
        return self

    def _sqrt(self):
        return self


def ReducedLinearOperator(op, row_indices, col_indices):
    """Restriction of `op` to the given rows and columns (linop.py:560-587).  Of a device matrix: a device operator
    (`_ReducedCsrOperator`, scatter / product / gather on the GPU), accepted by the device solvers as it is."""
    dev = _ReducedCsrOperator.build(op, row_indices, col_indices, False) if isinstance(op, CsrOperator) else None
    if dev is not None:
        return dev
    nargin, nargout = len(col_indices), len(row_indices)
    m, n = op.shape

    def matvec(x):
        z = np.zeros(n, dtype=x.dtype)
        z[col_indices] = x[:]
        return (op * z)[row_indices]

    def matvec_transp(x):
        z = np.zeros(m, dtype=x.dtype)
        z[row_indices] = x[:]
        return (op.T * z)[col_indices]

    return LinearOperator(nargin, nargout, matvec=matvec, symmetric=False, matvec_transp=matvec_transp)


def SymmetricallyReducedLinearOperator(op, indices):
    "Restriction of `op` to the same rows and columns (linop.py:590-623); of a device matrix: a device operator."
    dev = _ReducedCsrOperator.build(op, indices, indices, bool(op.symmetric)) if isinstance(op, CsrOperator) else None
    if dev is not None:
        return dev
    nargin = len(indices)
    m, n = op.shape

    def matvec(x):
        z = np.zeros(n, dtype=x.dtype)
        z[indices] = x[:]
        return (op * z)[indices]

    def matvec_transp(x):
        z = np.zeros(m, dtype=x.dtype)
        z[indices] = x[:]
        return (op * z)[indices]

    return LinearOperator(nargin, nargin, matvec=matvec, symmetric=op.symmetric, matvec_transp=matvec_transp)


def linop_from_ndarray(A, symmetric=False, **kwargs):
    "Operator from a dense Numpy array (linop.py:723-745)."
    hermitian = kwargs.get('hermitian', symmetric)
    if _is_complex(A.dtype):
        return LinearOperator(A.shape[1], A.shape[0], lambda v: np.dot(A, v),
                              matvec_transp=lambda u: np.dot(A.T, u),
                              matvec_adj=lambda w: np.dot(A.conjugate().T, w),
                              symmetric=symmetric, hermitian=hermitian, dtype=A.dtype)
    if symmetric ^ hermitian:
        raise ValueError('For non-complex operators, transpose = adjoint.')
    return LinearOperator(A.shape[1], A.shape[0], lambda v: np.dot(A, v),
                          matvec_transp=lambda u: np.dot(A.T, u),
                          symmetric=symmetric or hermitian, hermitian=symmetric or hermitian, dtype=A.dtype)


def sqrt(op):
    "Operator square root, where the operator defines one (linop.py:748-754)."
    return op._sqrt()


# ======================================================================================
# device-resident CSR operator
# ======================================================================================
def _canonical_csr(indptr, indices, data, shape):
    indptr = np.ascontiguousarray(indptr)
    indices = np.ascontiguousarray(indices)
    if indptr.ndim != 1 or indptr.shape[0] != shape[0] + 1:
        raise ShapeError('indptr has length %d, expected %d' % (indptr.shape[0], shape[0] + 1))
    if indptr[0] != 0 or np.any(np.diff(indptr) < 0):
        raise ValueError('indptr must start at 0 and be non-decreasing')
    nnz = int(indptr[-1])
    if indices.shape[0] != nnz or np.asarray(data).shape[0] != nnz:
        raise ShapeError('indices/data must have indptr[-1] = %d entries' % nnz)
    if nnz > np.iinfo(np.int32).max or shape[1] > np.iinfo(np.int32).max:
        raise ValueError('matrix too large for int32 indices')
    if indices.dtype != np.int32:                            # (wider indices: the range check before they are narrowed;
        if nnz and (indices.min() < 0 or indices.max() >= shape[1]):   # int32 arrays are checked on the device)
            raise ValueError('column index out of range')
    return (indptr.astype(np.int32, copy=False), indices.astype(np.int32, copy=False),
            np.ascontiguousarray(data, dtype=np.float64), nnz)


class CsrOperator(LinearOperator):
    """``LinearOperator`` whose matrix is a canonical CSR (sorted columns, no duplicates,
    int32 indices, fp64 values) resident in HBM.

    :parameters:
        :indptr, indices, data:  host arrays (copied to the device once)
        :shape:                  ``(nargout, nargin)``

    :keywords:
        :symmetric:  declare A = A^T (then ``op.T is op``)

    ``op * x`` (x an ndarray) uploads x, runs the gfx950 CSR-stream kernel and returns a NEW
    ndarray, exactly like a reference operator; ``op.T`` is a second ``CsrOperator`` built on
    the device on first use (needed by the ``lls`` solvers, lls/lsqr.py:200).
    """

    def __init__(self, indptr, indices, data, shape, symmetric=False, **kwargs):
        indptr, indices, data, nnz = _canonical_csr(indptr, indices, data, shape)
        lib = _lib.init()
        h = ctypes.c_void_p()
        _lib.check(lib.mk_csr_create(shape[0], shape[1], nnz, indptr.ctypes.data, indices.ctypes.data,
                                     data.ctypes.data, ctypes.byref(h)))
        lo, hi = ctypes.c_int32(), ctypes.c_int32()
        _lib.check(lib.mk_csr_col_range(h, ctypes.byref(lo), ctypes.byref(hi)))
        if nnz and (lo.value < 0 or hi.value >= shape[1]):
            lib.mk_csr_destroy(h)
            raise ValueError('column index out of range')
        self._finish_init(h.value, shape, nnz, symmetric, kwargs)

    @classmethod
    def from_handle(cls, handle, symmetric=False, **kwargs):
        "Wrap a ``mk_csr*`` created in HBM (e.g. by the on-device generators in `gallery`)."
        lib = _lib.init()
        m, n, nnz = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        _lib.check(lib.mk_csr_shape(handle, ctypes.byref(m), ctypes.byref(n), ctypes.byref(nnz)))
        self = cls.__new__(cls)
        self._finish_init(handle, (m.value, n.value), nnz.value, symmetric, kwargs)
        return self

    @classmethod
    def from_coo(cls, rows, cols, vals, shape, symmetric=False, **kwargs):
        """Operator from 0-based coordinate triples, assembled into canonical CSR on the device
        (``mk_csr_from_coo``: columns sorted per row, duplicates summed in input order).  `symmetric` only declares
        A = A^T; mirror one-triangle storage yourself (see `CoordLinearOperator`)."""
        lib = _lib.init()
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        cols = np.ascontiguousarray(cols, dtype=np.int64)
        vals = np.ascontiguousarray(vals, dtype=np.float64)
        if not (rows.shape == cols.shape == vals.shape) or rows.ndim != 1:
            raise ValueError('rows, cols and vals must be 1-d arrays of equal length')
        m, n = int(shape[0]), int(shape[1])
        if rows.size and (rows.min() < 0 or rows.max() >= m or cols.min() < 0 or cols.max() >= n):
            raise ValueError('coordinate index out of range for shape %s' % (shape,))
        if rows.size and np.bincount(rows, minlength=m).max() > 16384:
            # the device builder sorts each row with one thread; very long rows are assembled on the host instead
            # (construction only -- products and solvers always run on the device)
            from .sparse import coo_to_csr
            indptr, indices, data = coo_to_csr(rows, cols, vals, (m, n))
            return cls(indptr, indices, data, (m, n), symmetric=symmetric, **kwargs)
        r32, c32 = rows.astype(np.int32), cols.astype(np.int32)
        h = ctypes.c_void_p()
        _lib.check(lib.mk_csr_from_coo(m, n, rows.size, r32.ctypes.data, c32.ctypes.data, vals.ctypes.data,
                                       ctypes.byref(h)))
        return cls.from_handle(h.value, symmetric=symmetric, **kwargs)

    def _finish_init(self, handle, shape, nnz, symmetric, kwargs):
        self._lib = _lib.init()
        self._handle = handle
        self._nnz = int(nnz)
        self._T_cache = None
        self._xbuf = self._ybuf = None
        transpose_of = kwargs.pop('transpose_of', None)
        LinearOperator.__init__(self, shape[1], shape[0], matvec=self._device_matvec, symmetric=symmetric,
                                dtype=np.float64, **kwargs)
        if transpose_of is not None:
            self._T_cache = transpose_of

    # device handle for the solver fast path
    handle = property(lambda self: self._handle, doc="Opaque ``mk_csr*`` for libmikrylov.")
    nnz = property(lambda self: self._nnz)

    @property
    def T(self):
        if self.symmetric:
            return self
        if getattr(self, 'local_size', None) is not None:
            raise NotImplementedError('the transpose of a row-partitioned operator is not available (its columns '
                                      'are localised); transpose before partitioning')
        if self._T_cache is None:
            h = ctypes.c_void_p()
            _lib.check(self._lib.mk_csr_transpose(self._handle, ctypes.byref(h)))
            t = type(self).from_handle(h.value, transpose_of=self) if isinstance(self, _ComposedCsrOperator) \
                else CsrOperator.from_handle(h.value, transpose_of=self)
            # (alpha A + D)^T = alpha A^T + D: the C handle carries the row program over; so must the Python object,
            # or a further composition would not see the steps already used
            for attr in ('_steps', '_diag_bufs'):
                if hasattr(self, attr):
                    setattr(t, attr, getattr(self, attr))
            if hasattr(self, '_base'):
                t._base = self._base.T if self._base is not self else t
            self._T_cache = t
        return self._T_cache

    H = T

    # -- operator algebra that stays on the device (reference linop.py:307-330, :375-426) --------
    # `alpha * A`, `-A`, `A / alpha`, `A + D`, `A - D`, `D + A`, `D - A` with D an IdentityOperator, a
    # DiagonalOperator or a real scalar multiple of one return a CsrOperator that shares A's arrays and applies the
    # reference's expression (`alpha * (A*x)`, `(A*x) + (D*x)`, ...) to every row sum inside the SpMV kernel
    # (mk_csr_compose): same bits as the host closures of the reference, and the result is still accepted by the
    # device solvers.  Anything else (A + B, A * B, complex scalars) composes on the host as in the reference.
    def _compose(self, steps, diag_bufs=()):
        if getattr(self, 'local_size', None) is not None:
            return None                                   # partitioned operators: compose before partitioning
        if len(getattr(self, '_steps', ())) + len(steps) > _lib.MK_ROWPROG_MAX:
            return None
        ops = (_lib.MkRowOp * len(steps))()
        for k, (code, has_scale, scale, dptr) in enumerate(steps):
            ops[k].code, ops[k].has_scale, ops[k].scale, ops[k].diag = code, has_scale, scale, dptr
        h = ctypes.c_void_p()
        if self._lib.mk_csr_compose(self._handle, len(steps), ops, ctypes.byref(h)) != 0:
            return None                                   # (too many steps, pair operator ...): host composition
        new = _ComposedCsrOperator.from_handle(h.value, symmetric=self.symmetric)
        new._base = self                                  # keeps the arrays (and earlier diagonals) alive
        new._diag_bufs = tuple(diag_bufs)
        new._steps = tuple(getattr(self, '_steps', ())) + tuple(steps)
        return new

    def _times_scalar(self, x):
        if x != 0 and _is_real_scalar(x):
            dev = self._compose([(_lib.MK_ROW_SCALE, 1, float(x), None)])
            if dev is not None:
                return dev
        return LinearOperator._times_scalar(self, x)

    def _compose_with_term(self, term, how):
        if self.nargin != self.nargout:
            return None
        diag, scale = term
        bufs, dptr = [], None
        if diag is not None:
            buf = _lib.DeviceArray.from_numpy(np.ascontiguousarray(diag, dtype=np.float64))
            bufs.append(buf)
            dptr = buf.ptr
        code = {'add': _lib.MK_ROW_ADD, 'sub': _lib.MK_ROW_SUB, 'rsub': _lib.MK_ROW_RSUB}[how]
        return self._compose([(code, int(scale is not None), float(scale or 0.0), dptr)], bufs)

    def _combine(self, other, f):
        if isinstance(other, BaseLinearOperator) and self.shape == other.shape and \
                getattr(other, '_mk_term', None) is not None:
            dev = self._compose_with_term(other._mk_term, 'add' if f is np.add else 'sub')
            if dev is not None:
                return dev
        if isinstance(other, CsrOperator) and self.shape == other.shape:
            dev = _PairCsrOperator.build(self, other, 'add' if f is np.add else 'sub')   # A + B, A - B on the device
            if dev is not None:
                return dev
        return LinearOperator._combine(self, other, f)

    def _times_linop(self, op):
        if isinstance(op, CsrOperator) and self.nargin == op.nargout:
            dev = _PairCsrOperator.build(self, op, 'mul')                                 # A * B on the device
            if dev is not None:
                return dev
        return LinearOperator._times_linop(self, op)

    def _pair_ok(self):
        "May this operator be an operand of a device sum / product?  (a matrix of its own, not partitioned)"
        return getattr(self, 'local_size', None) is None

    def _device_matvec(self, x):
        if getattr(self, 'local_size', None) is not None:
            raise NotImplementedError('a row-partitioned CsrOperator multiplies inside the device solvers only '
                                      '(its input needs the halo exchange); use the solver classes or '
                                      'pykrylov_amd.tools.check_symmetric')
        if x.dtype != np.float64:
            if _kind(x.dtype) not in _INT_KINDS + _REAL_KINDS:
                raise TypeError('CsrOperator is fp64-only on the device; got %s' % x.dtype)
            x = x.astype(np.float64)
        if self._xbuf is None:
            self._xbuf = _lib.DeviceArray(self.nargin, zero=False)
            self._ybuf = _lib.DeviceArray(self.nargout, zero=False)
        self._xbuf.upload(x)
        _lib.check(self._lib.mk_spmv(self._handle, self._xbuf.ptr, self._ybuf.ptr))
        return self._ybuf.to_numpy()          # a fresh ndarray every call (solvers update it in place)

    def _times_vector(self, x):
        # (the base class copies what the user's callable returned, linop.py:356-360; the device product already is a
        # fresh array: one gigabyte less to move per product at 512^3)
        self._nMatvec += 1
        return self._matvec(x).astype(np.result_type(self.dtype, x.dtype), copy=False)

    def spmv_device(self, x_ptr, y_ptr):
        "y = A x on device pointers (no host traffic); counts as a product."
        self._nMatvec += 1
        _lib.check(self._lib.mk_spmv(self._handle, x_ptr, y_ptr))

    def to_csr_arrays(self):
        "Download ``(indptr, indices, data)``."
        indptr = np.empty(self.nargout + 1, dtype=np.int32)
        indices = np.empty(self._nnz, dtype=np.int32)
        data = np.empty(self._nnz, dtype=np.float64)
        _lib.check(self._lib.mk_csr_download(self._handle, indptr.ctypes.data, indices.ctypes.data,
                                             data.ctypes.data))
        return indptr, indices, data

    def csr_rows(self, row_begin, row_end):
        """Download rows ``[row_begin, row_end)``: ``(indptr, indices, data)`` with the row pointers rebased to 0 (global
        column ids).  How a matrix too large for the host is compared with the oracle slab by slab."""
        cnt = int(row_end) - int(row_begin)
        indptr = np.empty(cnt + 1, dtype=np.int32)
        _lib.check(self._lib.mk_csr_download_rows(self._handle, int(row_begin), int(row_end), indptr.ctypes.data,
                                                  None, None))
        nz = int(indptr[-1]) - int(indptr[0])
        indices = np.empty(nz, dtype=np.int32)
        data = np.empty(nz, dtype=np.float64)
        _lib.check(self._lib.mk_csr_download_rows(self._handle, int(row_begin), int(row_end), None,
                                                  indices.ctypes.data, data.ctypes.data))
        return indptr - indptr[0], indices, data

    def free(self):
        if getattr(self, '_handle', None):
            try:
                self._lib.mk_csr_destroy(self._handle)
            except Exception:
                pass
            self._handle = None

    def __del__(self):
        self.free()


class _PairCsrOperator(CsrOperator):
    """`A + B`, `A - B`, `A * B` of two CsrOperators as ONE device operator (mk_csr_create_sum / _product): evaluated
    as the reference's closures are -- `(A*x) + (B*x)`, `A*(B*x)`, two complete products and one element-wise step
    (linop.py:332-354, :375-426) -- but on the device, inside the solver kernels' product site, so the result is
    accepted by the device solvers without host round trips.  Products are counted on this operator and on both
    operands, as the reference's closures do."""

    @classmethod
    def build(cls, a, b, how):
        if not (a._pair_ok() and b._pair_ok()):
            return None
        lib = _lib.init()
        h = ctypes.c_void_p()
        if how == 'mul':
            rc = lib.mk_csr_create_product(a.handle, b.handle, ctypes.byref(h))
        else:
            rc = lib.mk_csr_create_sum(a.handle, b.handle, 1 if how == 'add' else -1, ctypes.byref(h))
        if rc != 0:
            return None                                       # e.g. an operand is itself a pair: host closure
        sym = a.symmetric and b.symmetric and how != 'mul'
        self = cls.from_handle(h.value, symmetric=sym)
        self._parts = (a, b, how)                             # keeps the operands alive
        return self

    def _pair_ok(self):
        return False

    def _compose(self, steps, diag_bufs=()):
        return None                                           # alpha * (A + B): host closure (works through callbacks)

    def _get_count(self):
        return self.__dict__.get('_count', 0)

    def _set_count(self, v):
        delta = v - self.__dict__.get('_count', 0)
        self.__dict__['_count'] = v
        parts = self.__dict__.get('_parts')
        if parts is not None and delta:
            parts[0]._nMatvec += delta
            parts[1]._nMatvec += delta

    _nMatvec = property(_get_count, _set_count)

    @property
    def T(self):
        if self.symmetric:
            return self
        if self._T_cache is None:
            a, b, how = self._parts
            if how == 'mul':
                t = b.T * a.T                                 # (A B)^T = B^T A^T   (linop.py:350-351)
            else:
                t = (a.T + b.T) if how == 'add' else (a.T - b.T)
            if isinstance(t, _PairCsrOperator):
                t._T_cache = self
            self._T_cache = t
        return self._T_cache

    H = T

    def to_csr_arrays(self):
        raise NotImplementedError('a device sum / product has no matrix of its own; use its operands')


class _ReducedCsrOperator(CsrOperator):
    """`ReducedLinearOperator` of a device matrix as ONE device operator (mk_csr_create_reduced): z = 0, z[cols] = x,
    y = (A z)[rows] with every step on the device.  Products are counted on this operator and on the matrix it
    restricts, as the reference's closure does (it evaluates `op * z`)."""

    @classmethod
    def build(cls, base, row_indices, col_indices, symmetric):
        rows = np.ascontiguousarray(np.asarray(row_indices), dtype=np.int64).ravel()
        cols = np.ascontiguousarray(np.asarray(col_indices), dtype=np.int64).ravel()
        if not base._pair_ok() or len(np.unique(cols)) != len(cols):
            return None                                       # (repeated column indices: NumPy keeps the last -- host)
        m, n = base.shape
        if (rows.size and (rows.min() < 0 or rows.max() >= m)) or (cols.size and (cols.min() < 0 or cols.max() >= n)):
            return None                                       # (negative / out-of-range indices: NumPy semantics on the host)
        r32, c32 = rows.astype(np.int32), cols.astype(np.int32)
        h = ctypes.c_void_p()
        if _lib.init().mk_csr_create_reduced(base.handle, r32.size, r32.ctypes.data, c32.size, c32.ctypes.data,
                                             ctypes.byref(h)) != 0:
            return None
        self = cls.from_handle(h.value, symmetric=symmetric)
        self._parts = (base, rows, cols)
        return self

    def _pair_ok(self):
        return False

    def _compose(self, steps, diag_bufs=()):
        return None

    def _get_count(self):
        return self.__dict__.get('_count', 0)

    def _set_count(self, v):
        delta = v - self.__dict__.get('_count', 0)
        self.__dict__['_count'] = v
        parts = self.__dict__.get('_parts')
        if parts is not None and delta:
            parts[0]._nMatvec += delta

    _nMatvec = property(_get_count, _set_count)

    @property
    def T(self):
        if self.symmetric:
            return self
        if self._T_cache is None:
            base, rows, cols = self._parts
            t = ReducedLinearOperator(base.T, cols, rows)     # (linop.py:577-580: z[rows] = x ; (op.T * z)[cols])
            if isinstance(t, _ReducedCsrOperator):
                t._T_cache = self
            self._T_cache = t
        return self._T_cache

    H = T

    def to_csr_arrays(self):
        raise NotImplementedError('a reduced device operator has no matrix of its own; slice the base matrix')


class _BlockCsrOperator(CsrOperator):
    """A grid of device matrices as ONE device operator (mk_csr_create_block) -- what the device solvers see of a
    `blkop.BlockLinearOperator` / `BlockDiagonalLinearOperator` whose blocks all live on the device.  A product runs
    one launch per block and adds the block products to each block row one at a time, `y_i += B_ij * x_j`
    (reference blkop.py:86-96): the reference's roundings, no host round trip.  Products are counted on the block
    operator it was built from and on every block, as the reference's closures do."""

    @classmethod
    def build(cls, owner, grid, heights, widths):
        "`grid`: rows of CsrOperator / None (zero block).  Returns None if the library refuses (e.g. a composite block)."
        lib = _lib.init()
        nbr, nbc = len(grid), len(grid[0])
        flat = [blk for row in grid for blk in row]
        arr = (ctypes.c_void_p * len(flat))(*[None if b is None else b.handle for b in flat])
        hs = (ctypes.c_int64 * nbr)(*[int(h) for h in heights])
        ws = (ctypes.c_int64 * nbc)(*[int(w) for w in widths])
        h = ctypes.c_void_p()
        if lib.mk_csr_create_block(nbr, nbc, arr, hs, ws, ctypes.byref(h)) != 0:
            return None
        self = cls.from_handle(h.value, symmetric=bool(getattr(owner, 'symmetric', False)))
        self._owner = owner
        self._parts = [b for b in flat if b is not None]      # keeps the blocks alive
        return self

    def _pair_ok(self):
        return False

    def _compose(self, steps, diag_bufs=()):
        return None

    def _get_count(self):
        return self.__dict__.get('_count', 0)

    def _set_count(self, v):
        delta = v - self.__dict__.get('_count', 0)
        self.__dict__['_count'] = v
        if delta and self.__dict__.get('_owner') is not None:
            self._owner._nMatvec += delta
            for b in self.__dict__.get('_parts', ()):
                b._nMatvec += delta
                orig = b.__dict__.get('_count_owner')         # a Diagonal / Identity block seen through its device form
                if orig is not None:
                    orig._nMatvec += delta

    _nMatvec = property(_get_count, _set_count)

    @property
    def T(self):
        if self.symmetric:
            return self
        raise NotImplementedError('transpose the block operator itself (its transpose has its own device view)')

    H = T

    def to_csr_arrays(self):
        raise NotImplementedError('a device block operator has no matrix of its own; use its blocks')


def device_block(blk):
    """A block of a block operator as a device matrix, if it is one or has an exact device form: a CsrOperator as it
    is; a DiagonalOperator / IdentityOperator as the CSR matrix of its diagonal (`diag * x`, linop.py:503 -- one
    product per row either way; cached on the operator).  None for anything else."""
    if isinstance(blk, CsrOperator):
        return blk if (blk._pair_ok() and blk.handle) else None
    term = getattr(blk, '_mk_term', None)
    if term is None or term[1] is not None or not isinstance(blk, BaseLinearOperator) or blk.nargin != blk.nargout:
        return None                                           # (scaled diagonals round twice in the reference: host)
    cached = blk.__dict__.get('_device_diag_csr')
    if cached is None or not cached.handle:
        n = blk.nargin
        diag = np.ones(n) if term[0] is None else np.ascontiguousarray(term[0], dtype=np.float64)
        if _is_complex(np.asarray(diag).dtype):
            return None
        cached = CsrOperator(np.arange(n + 1), np.arange(n), diag, (n, n), symmetric=True)
        cached._count_owner = blk                             # (products are counted on the block the caller made)
        blk.__dict__['_device_diag_csr'] = cached
    return cached


class HostOperatorShell(object):
    """What the device solvers see of an operator that is NOT a CsrOperator: any object following the reference's
    protocol, ``op * ndarray -> ndarray`` (linop.py:271-298, e.g. ``LinearOperator(n, n, matvec=callable)`` or the
    gallery operators of cg/tests/test_diagdom.py:38-40).  The solver loop stays on the device; at every product site
    libmikrylov copies the loop's input vector to the host and calls back into `op` (mk_csr_create_callback), exactly
    when the reference would have evaluated ``op * v`` -- so ``op.nMatvec`` counts what it counts in the reference.
    An exception raised by the operator aborts the solve and is re-raised to the caller."""

    def __init__(self, op):
        self.host_op = op
        self.error = None
        dt = getattr(op, 'dtype', None)
        if dt is not None and _is_complex(np.dtype(dt)):     # (found here, not at the first callback)
            raise TypeError('complex operators are not supported on the device path (fp64 only)')
        nargout, nargin = op.shape
        self.shape = (int(nargout), int(nargin))
        self.symmetric = bool(getattr(op, 'symmetric', False))
        self.local_size = None
        self._nMatvec = 0                                   # (the wrapped operator keeps the real count itself)
        self._T = None
        self._lib = _lib.init()
        nrows, ncols = self.shape

        def call(user, transpose, xp, yp):
            try:
                x = np.ctypeslib.as_array(ctypes.cast(xp, ctypes.POINTER(ctypes.c_double)), shape=(ncols,)).copy()
                y = np.asarray(self.host_op * x)
                if _is_complex(y.dtype):
                    raise TypeError('operator returned complex data; the device solvers are fp64-only')
                if y.shape != (nrows,):
                    raise ValueError('operator returned shape %s, expected (%d,)' % (y.shape, nrows))
                np.ctypeslib.as_array(ctypes.cast(yp, ctypes.POINTER(ctypes.c_double)), shape=(nrows,))[:] = y
                return 0
            except BaseException as exc:                    # noqa: B902  (must not propagate through the C frames)
                if self.error is None:                      # (keep the root cause, not a follow-up failure)
                    self.error = exc
                return 1
        self._cb = _lib.MATVEC_FN(call)                     # keep the thunk alive
        h = ctypes.c_void_p()
        _lib.check(self._lib.mk_csr_create_callback(nrows, ncols, self._cb, None, 0, ctypes.byref(h)))
        self._handle = h.value

    handle = property(lambda self: self._handle)
    nnz = 0

    @property
    def T(self):
        if self.symmetric:
            return self
        if self._T is None:
            t = getattr(self.host_op, 'T', None)
            if t is None:                                    # (LinearOperator built without matvec_transp: linop.py:148-171)
                raise ValueError('the operator has no transpose (build it with matvec_transp=...): the least-squares '
                                 'solvers need A.T * u')
            self._T = HostOperatorShell(t)
            self._T._T = self
        return self._T

    def raise_pending(self):
        "Re-raise what the wrapped operator raised inside a callback (called when a libmikrylov call failed)."
        for sh in (self, self._T):
            if sh is not None and sh.error is not None:
                err, sh.error = sh.error, None
                raise err

    def free(self):
        if getattr(self, '_handle', None):
            try:
                self._lib.mk_csr_destroy(self._handle)
            except Exception:
                pass
            self._handle = None

    def __del__(self):
        self.free()


class _ComposedCsrOperator(CsrOperator):
    """A CsrOperator with a row program (see CsrOperator._compose).  Products are counted on this operator and on
    the matrix it was built from, as the reference's composite closures do (linop.py:356-360 via `self(y)`)."""

    def _get_count(self):
        return self.__dict__.get('_count', 0)

    def _set_count(self, v):
        delta = v - self.__dict__.get('_count', 0)
        self.__dict__['_count'] = v
        base = self.__dict__.get('_base')
        if base is not None and delta > 0:
            base._nMatvec += delta

    _nMatvec = property(_get_count, _set_count)

    def to_csr_arrays(self):
        "The arrays of the underlying matrix (the composed steps are not folded into them)."
        return self._base.to_csr_arrays()


def CoordLinearOperator(vals, rows, cols, nargin=0, nargout=0, symmetric=False):
    """Operator from a coordinate-format matrix (reference linop.py:638-685), compiled to a
    device CSR instead of the reference's per-nonzero Python loop.  If `symmetric`, the
    triples describe one triangle and are mirrored (MatrixMarket symmetric storage)."""
    vals = np.asarray(vals, dtype=np.float64)
    rows = np.asarray(rows).astype(np.int64)
    cols = np.asarray(cols).astype(np.int64)
    if nargin == 0:
        nargin = int(cols.max()) + 1
    if nargout == 0:
        nargout = int(rows.max()) + 1
    if symmetric:
        off = rows != cols
        rows, cols, vals = (np.concatenate([rows, cols[off]]), np.concatenate([cols, rows[off]]),
                            np.concatenate([vals, vals[off]]))
    return CsrOperator.from_coo(rows, cols, vals, (nargout, nargin), symmetric=symmetric)
