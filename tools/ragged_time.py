"""fmt 0 against fmt 6 on banded matrices with ragged rows (row lengths uniform in [lo, hi], random columns within
+-spread of the diagonal): where does the padding of the ELL blocks stop paying?   gpurun: python tools/ragged_time.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pykrylov_amd import _lib, CsrOperator
from pykrylov_amd.generic import DeviceRun

lib = _lib.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
rng = np.random.default_rng(3)
for lo, hi in ((20, 20), (16, 24), (12, 24), (8, 24), (4, 28)):
    lens = rng.integers(lo, hi + 1, size=n)
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    nnz = int(indptr[-1])
    # columns: sorted distinct offsets within +-2000 of the diagonal (vectorised: a random start + strictly increasing steps)
    steps = rng.integers(1, 4000 // (hi + 1), size=nnz)
    rows = np.repeat(np.arange(n), lens)
    first = indptr[:-1][rows]
    csum = np.cumsum(steps)
    off = csum - csum[first] + steps[first]
    cols = np.clip(rows - 2000 + off, 0, n - 1)
    # make strictly increasing per row after clipping (rare at the ends): drop duplicates by re-sorting small rows is costly; tolerate
    data = rng.standard_normal(nnz)
    op = CsrOperator(indptr, cols.astype(np.int32), data, (n, n))
    x = _lib.DeviceArray.from_numpy(rng.standard_normal(n))
    y = _lib.DeviceArray(n)
    line = "rows %d..%d (nnz %.2e, max/mean %.2f):" % (lo, hi, nnz, hi / lens.mean())
    for f in (0, 6):
        _lib.check(lib.mk_csr_set_format(op.handle, f))
        op.spmv_device(x.ptr, y.ptr)
        fmt, by = ctypes.c_int32(), ctypes.c_int64()
        _lib.check(lib.mk_csr_format_info(op.handle, ctypes.byref(fmt), None, None, None, ctypes.byref(by)))
        run = DeviceRun(op, _lib.MK_CG, x, None, abstol=0.0, reltol=0.0, matvec_max=1 << 60, check_curvature=1)
        run.setup()
        avg = ctypes.c_double()
        _lib.check(lib.mk_solver_time_spmv(run.handle, 40, ctypes.byref(avg)))
        line += "   want %d -> fmt %d %.3f GB %7.1f us" % (f, fmt.value, by.value / 1e9, avg.value)
        del run
    print(line, flush=True)
    op.free()
