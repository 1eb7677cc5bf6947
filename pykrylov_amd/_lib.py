"""ctypes binding of libmikrylov.so (C ABI: include/mikrylov.h).

The shared object is the product: if it is missing or cannot be loaded this module
raises -- there is no CPU fallback anywhere in the package.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (MIKRYLOV_LIB: an experiment build of the same library, `python -m pykrylov_amd.build --tag X -DY`)
LIB_PATH = os.environ.get("MIKRYLOV_LIB") or os.path.join(_HERE, "libmikrylov.so")

c_i32, c_i64, c_f64 = ctypes.c_int32, ctypes.c_int64, ctypes.c_double
c_vp, c_sz = ctypes.c_void_p, ctypes.c_size_t
P = ctypes.POINTER

MK_CG, MK_BICGSTAB, MK_CGS, MK_TFQMR, MK_MINRES, MK_SYMMLQ = 1, 2, 3, 4, 5, 6
MK_LSQR, MK_LSMR, MK_CRAIG, MK_CRAIGMR = 7, 8, 9, 10


class MkParams(ctypes.Structure):
    _fields_ = [("struct_size", c_i32), ("kind", c_i32), ("abstol", c_f64), ("reltol", c_f64),
                ("matvec_max", c_i64), ("check_curvature", c_i32), ("has_shift", c_i32), ("shift", c_f64),
                ("rtol", c_f64), ("etol", c_f64), ("itnlim", c_i64), ("window", c_i32), ("spmv_event_stride", c_i32),
                ("damp", c_f64), ("atol", c_f64), ("btol", c_f64), ("conlim", c_f64)]


class MkRowOp(ctypes.Structure):
    "mk_rowop (include/mikrylov.h): one step of a composed operator."
    _fields_ = [("code", ctypes.c_int32), ("has_scale", ctypes.c_int32), ("scale", ctypes.c_double),
                ("diag", ctypes.c_void_p)]


MK_ROW_SCALE, MK_ROW_ADD, MK_ROW_SUB, MK_ROW_RSUB, MK_ROWPROG_MAX = 1, 2, 3, 4, 4


# host-staged transport callbacks (mk_comm_init_host)
HOST_ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int64)
HOST_EXCHANGE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64),
                                    ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p,
                                    ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64))
HOST_ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p)
# matrix-free operator callback (mk_csr_create_callback): fn(user, transpose, x_host, y_host) -> 0 on success
MATVEC_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)
# general preconditioner callback (mk_solver_set_precon_callback): fn(user, r_host, y_host) -> 0 on success
PRECON_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)


class MkResult(ctypes.Structure):
    _fields_ = [("struct_size", c_i32), ("halted", c_i32), ("nMatvec", c_i64), ("itn", c_i64),
                ("hist_len", c_i64), ("converged", c_i32), ("definite", c_i32), ("istop", c_i32),
                ("reserved", c_i32), ("residNorm", c_f64), ("residNorm0", c_f64), ("threshold", c_f64),
                ("Anorm", c_f64), ("Acond", c_f64), ("Arnorm", c_f64), ("ynorm", c_f64), ("xnorm", c_f64),
                ("aux", c_f64 * 8)]


# name -> (restype, argtypes); must list every function declared in include/mikrylov.h
PROTOTYPES = {
    "mk_version": (ctypes.c_int, []),
    "mk_build_info": (ctypes.c_char_p, []),
    "mk_init": (ctypes.c_int, [ctypes.c_int]),
    "mk_shutdown": (ctypes.c_int, []),
    "mk_last_error": (ctypes.c_char_p, []),
    "mk_device_info": (ctypes.c_int, [ctypes.c_char_p, P(ctypes.c_int), P(c_sz)]),
    "mk_sync": (ctypes.c_int, []),
    "mk_malloc": (ctypes.c_int, [P(c_vp), c_sz]),
    "mk_free": (ctypes.c_int, [c_vp]),
    "mk_memcpy_h2d": (ctypes.c_int, [c_vp, c_vp, c_sz]),
    "mk_memcpy_d2h": (ctypes.c_int, [c_vp, c_vp, c_sz]),
    "mk_memcpy_d2d": (ctypes.c_int, [c_vp, c_vp, c_sz]),
    "mk_memset": (ctypes.c_int, [c_vp, ctypes.c_int, c_sz]),
    "mk_arena_reserve": (ctypes.c_int, [c_sz]),
    "mk_calib_stream": (ctypes.c_int, [c_vp, c_i64, ctypes.c_int, ctypes.c_int]),
    "mk_csr_create": (ctypes.c_int, [c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, P(c_vp)]),
    "mk_csr_destroy": (ctypes.c_int, [c_vp]),
    "mk_csr_create_sum": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, P(c_vp)]),
    "mk_csr_create_product": (ctypes.c_int, [c_vp, c_vp, P(c_vp)]),
    "mk_csr_create_reduced": (ctypes.c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, P(c_vp)]),
    "mk_csr_create_block": (ctypes.c_int, [c_i32, c_i32, P(c_vp), P(c_i64), P(c_i64), P(c_vp)]),
    "mk_csr_create_callback": (ctypes.c_int, [c_i64, c_i64, MATVEC_FN, c_vp, ctypes.c_int, P(c_vp)]),
    "mk_csr_shape": (ctypes.c_int, [c_vp, P(c_i64), P(c_i64), P(c_i64)]),
    "mk_csr_col_range": (ctypes.c_int, [c_vp, P(c_i32), P(c_i32)]),
    "mk_csr_download": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp]),
    "mk_csr_download_rows": (ctypes.c_int, [c_vp, c_i64, c_i64, c_vp, c_vp, c_vp]),
    "mk_csr_from_coo": (ctypes.c_int, [c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, P(c_vp)]),
    "mk_csr_transpose": (ctypes.c_int, [c_vp, P(c_vp)]),
    "mk_csr_compose": (ctypes.c_int, [c_vp, ctypes.c_int32, P(MkRowOp), P(c_vp)]),
    "mk_csr_poisson2d": (ctypes.c_int, [c_i64, c_i64, c_i64, P(c_vp)]),
    "mk_csr_poisson3d": (ctypes.c_int, [c_i64, c_i64, c_i64, c_i64, c_i64, P(c_vp)]),
    "mk_csr_poisson3d_varcoef": (ctypes.c_int, [c_i64, c_i64, c_i64, ctypes.c_uint64, c_i64, c_i64, P(c_vp)]),
    "mk_csr_stencil27": (ctypes.c_int, [c_i64, c_i64, c_i64, ctypes.c_uint64, c_i64, c_i64, P(c_vp)]),
    "mk_csr_set_format": (ctypes.c_int, [c_vp, ctypes.c_int]),
    "mk_csr_format_info": (ctypes.c_int, [c_vp, P(c_i32), P(c_i64), P(c_i32), P(c_i32), P(c_i64)]),
    "mk_csr_launch_info": (ctypes.c_int, [c_vp, P(c_i32), P(c_i32)]),
    "mk_csr_pencil_info": (ctypes.c_int, [c_vp, P(c_i64), P(c_i64), P(c_i32), P(c_i32), P(c_i32), P(c_i32)]),
    "mk_csr_march_info": (ctypes.c_int, [c_vp, P(c_i64), c_i32]),
    "mk_csr_colblocks": (ctypes.c_int, [c_vp, P(c_i32)]),
    "mk_csr_set_tile_order": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32]),
    "mk_csr_tile_order": (ctypes.c_int, [c_vp, P(c_i32), P(c_i32), P(c_i32), P(c_i32)]),
    "mk_csr_set_colblocks": (ctypes.c_int, [c_vp, c_i32]),
    "mk_spmv": (ctypes.c_int, [c_vp, c_vp, c_vp]),
    "mk_dot": (ctypes.c_int, [c_i64, c_vp, c_vp, P(c_f64)]),
    "mk_nrm2": (ctypes.c_int, [c_i64, c_vp, P(c_f64)]),
    "mk_axpy": (ctypes.c_int, [c_i64, c_f64, c_vp, c_vp]),
    "mk_axpby": (ctypes.c_int, [c_i64, c_f64, c_vp, c_f64, c_vp]),
    "mk_scal": (ctypes.c_int, [c_i64, c_f64, c_vp]),
    "mk_comm_unique_id": (ctypes.c_int, [c_vp]),
    "mk_comm_init": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_vp]),
    "mk_comm_init_host": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, HOST_ALLREDUCE_FN, HOST_EXCHANGE_FN,
                                         HOST_ALLGATHER_FN]),
    "mk_comm_destroy": (ctypes.c_int, []),
    "mk_comm_info": (ctypes.c_int, [P(ctypes.c_int), P(ctypes.c_int)]),
    "mk_comm_transport": (ctypes.c_int, [P(ctypes.c_int), P(ctypes.c_int), P(ctypes.c_int)]),
    "mk_csr_set_exchange": (ctypes.c_int, [c_vp, ctypes.c_int, c_i64, c_i64, c_vp, c_vp, c_vp]),
    "mk_csr_localize": (ctypes.c_int, [c_vp, ctypes.c_int, c_i64, c_i64, c_i64, P(c_i64), P(c_i64)]),
    "mk_exchange": (ctypes.c_int, [c_vp, c_vp]),
    "mk_comm_allreduce_host": (ctypes.c_int, [P(c_f64), c_i64]),
    "mk_csr_overlap_info": (ctypes.c_int, [c_vp, P(c_i64), P(c_i64)]),
    "mk_comm_time_exchange": (ctypes.c_int, [c_vp, c_vp, c_i64, P(c_f64)]),
    "mk_comm_time_allreduce": (ctypes.c_int, [c_i64, c_i64, P(c_f64)]),
    "mk_csr_comm_last_us": (ctypes.c_int, [c_vp, P(c_f64)]),
    "mk_solver_create": (ctypes.c_int, [c_vp, P(MkParams), P(c_vp)]),
    "mk_solver_destroy": (ctypes.c_int, [c_vp]),
    "mk_solver_set_transpose": (ctypes.c_int, [c_vp, c_vp]),
    "mk_csr_set_row_block": (ctypes.c_int, [c_vp, ctypes.c_int]),
    "mk_solver_set_precon_diag": (ctypes.c_int, [c_vp, c_vp]),
    "mk_solver_set_precon_callback": (ctypes.c_int, [c_vp, PRECON_FN, c_vp]),
    "mk_solver_set_precon_csr": (ctypes.c_int, [c_vp, c_vp]),
    "mk_solver_set_lls_precon_callback": (ctypes.c_int, [c_vp, PRECON_FN, c_vp, PRECON_FN, c_vp]),
    "mk_solver_set_lls_precon": (ctypes.c_int, [c_vp, c_vp, c_vp]),
    "mk_solver_setup": (ctypes.c_int, [c_vp, c_vp, c_vp]),
    "mk_solver_iterate": (ctypes.c_int, [c_vp, c_i64, P(c_i64)]),
    "mk_solver_finish": (ctypes.c_int, [c_vp, P(MkResult)]),
    "mk_solver_x": (ctypes.c_int, [c_vp, P(c_vp)]),
    "mk_solver_history": (ctypes.c_int, [c_vp, c_vp, c_i64]),
    "mk_solver_fused": (ctypes.c_int, [c_vp, P(c_i32)]),
    "mk_solver_history2": (ctypes.c_int, [c_vp, c_vp, c_i64]),
    "mk_solver_vector": (ctypes.c_int, [c_vp, ctypes.c_int, P(c_vp), P(c_i64)]),
    "mk_solver_timing": (ctypes.c_int, [c_vp, P(c_f64), P(c_f64), P(c_i64)]),
    "mk_solver_time_spmv": (ctypes.c_int, [c_vp, c_i64, P(c_f64)]),
    "mk_solver_time_product": (ctypes.c_int, [c_vp, ctypes.c_int, c_i64, P(c_f64)]),
    "mk_solver_solve": (ctypes.c_int, [c_vp, c_vp, c_vp, P(MkResult)]),
}


class MkError(RuntimeError):
    """A libmikrylov call failed (the message is mk_last_error())."""


_lib = None
_device = None


def load():
    """Load the shared object and declare prototypes (no GPU needed for this step)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libmikrylov.so is not built: run `python -m pykrylov_amd.build` "
                              "(pykrylov_amd has no CPU fallback)")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        check_build(lib)
        _lib = lib
    return _lib


def tree_sha():
    """Digest of the library sources of THIS tree (pykrylov_amd/build.py `source_sha`)."""
    from . import build as _build
    return _build.source_sha()


def build_info(lib=None):
    return (lib or load()).mk_build_info().decode(errors="replace")


def check_build(lib):
    """Refuse a binary compiled from other sources than the tree's: libmikrylov.so is git-ignored and travels prebuilt, so
    nothing else ties what runs to what is read.  MIKRYLOV_ALLOW_STALE=1 (or an experiment build named by MIKRYLOV_LIB)
    overrides."""
    if os.environ.get("MIKRYLOV_ALLOW_STALE") == "1" or os.environ.get("MIKRYLOV_LIB"):
        return
    have, want = lib.mk_build_info().decode(errors="replace"), tree_sha()
    if have != want:
        raise ImportError("libmikrylov.so was built from other sources than this tree's (binary %s, tree %s): run "
                          "`python -m pykrylov_amd.build`, or set MIKRYLOV_ALLOW_STALE=1 to load it anyway" % (have, want))


def check(rc):
    if rc != 0:
        raise MkError("libmikrylov error %d: %s" % (rc, load().mk_last_error().decode(errors="replace")))


def init(device=None):
    """Bind this process to a GPU (default: LOCAL_RANK or 0).  Raises MkError without a GPU."""
    global _device
    lib = load()
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0")) if _device is None else _device
    if _device is None or _device != device:
        check(lib.mk_init(int(device)))
        _device = int(device)
    return lib


def device_info():
    lib = init()
    name = ctypes.create_string_buffer(256)
    cus, mem = ctypes.c_int(0), c_sz(0)
    check(lib.mk_device_info(name, ctypes.byref(cus), ctypes.byref(mem)))
    return {"name": name.value.decode(), "compute_units": cus.value, "hbm_bytes": mem.value}


class DeviceArray(object):
    """A typed 1-D array in HBM owned by this object (mk_malloc / mk_free)."""

    def __init__(self, n, dtype=np.float64, zero=True):
        self.lib = init()
        self.n = int(n)
        self.dtype = np.dtype(dtype)
        p = c_vp()
        check(self.lib.mk_malloc(ctypes.byref(p), max(self.nbytes, 16)))
        self.ptr = p.value
        if zero and self.nbytes:
            check(self.lib.mk_memset(self.ptr, 0, self.nbytes))

    @property
    def nbytes(self):
        return self.n * self.dtype.itemsize

    @classmethod
    def from_numpy(cls, a, dtype=None):
        a = np.ascontiguousarray(a, dtype=dtype or a.dtype)
        d = cls(a.size, a.dtype, zero=False)
        if a.size:
            check(d.lib.mk_memcpy_h2d(d.ptr, a.ctypes.data, a.nbytes))
        return d

    def upload(self, a, count=None):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        count = a.size if count is None else count
        assert count <= self.n
        if count:
            check(self.lib.mk_memcpy_h2d(self.ptr, a.ctypes.data, count * self.dtype.itemsize))

    def to_numpy(self, count=None, offset=0):
        count = self.n - offset if count is None else count
        out = np.empty(count, dtype=self.dtype)
        if count:
            check(self.lib.mk_memcpy_d2h(out.ctypes.data, self.ptr + offset * self.dtype.itemsize, out.nbytes))
        return out

    def free(self):
        if getattr(self, "ptr", None):
            try:
                self.lib.mk_free(self.ptr)
            except Exception:
                pass
            self.ptr = None

    def __del__(self):
        self.free()


def download(ptr, n, dtype=np.float64):
    out = np.empty(int(n), dtype=dtype)
    if n:
        check(init().mk_memcpy_d2h(out.ctypes.data, ptr, out.nbytes))
    return out
