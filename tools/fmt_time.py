"""SpMV time of one gallery matrix under each storage format, in ONE process (same buffers for every format).

  gpurun: python tools/fmt_time.py <workload> [formats ...]
     workload: s27v-<nx> | s27c-<nx> (27-point, variable / constant coefficients) | p3dv-<nx> | p3d-<nx> | p2d-<m>
     formats: numbers for mk_csr_set_format (default: 0 and -1 = the library's choice)
Prints per format: format chosen, bytes of matrix data per product, kernel time (HIP events, the CG product kernel),
physical bandwidth.  Products are checked against each other bit for bit."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pykrylov_amd import _lib, gallery
from pykrylov_amd.generic import DeviceRun

lib = _lib.init(0)
wl = sys.argv[1] if len(sys.argv) > 1 else "s27v-128"
fmts = [int(a) for a in sys.argv[2:]] or [0, -1]
kind, size = wl.split("-")
m = int(size)
op = {"s27v": lambda: gallery.stencil27(m, seed=7), "s27c": lambda: gallery.stencil27(m, seed=0),
      "p3dv": lambda: gallery.poisson3d_varcoef(m), "p3d": lambda: gallery.poisson3d(m),
      "p2d": lambda: gallery.poisson2d(m)}[kind]()
n = op.shape[0]
rng = np.random.default_rng(1)
xh = rng.standard_normal(n)
x = _lib.DeviceArray.from_numpy(xh)
y = _lib.DeviceArray(n)
ref = None
print("workload %s: %d rows, %d nonzeros" % (wl, n, op.nnz))
for f in fmts:
    _lib.check(lib.mk_csr_set_format(op.handle, f))
    import time
    _lib.check(lib.mk_sync())
    t0 = time.perf_counter()
    op.spmv_device(x.ptr, y.ptr)                # (first product: builds the format)
    _lib.check(lib.mk_sync())
    t_build = time.perf_counter() - t0
    yh = y.to_numpy()
    if ref is None:
        ref = yh
    same = bool((yh.view(np.uint64) == ref.view(np.uint64)).all())
    fmt, tw, ch, nd, by = ctypes.c_int32(), ctypes.c_int64(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int64()
    _lib.check(lib.mk_csr_format_info(op.handle, ctypes.byref(fmt), ctypes.byref(tw), ctypes.byref(ch), ctypes.byref(nd), ctypes.byref(by)))
    run = DeviceRun(op, _lib.MK_CG, x, None, abstol=0.0, reltol=0.0, matvec_max=1 << 60, check_curvature=1)
    run.setup()
    avg = ctypes.c_double()
    _lib.check(lib.mk_solver_time_spmv(run.handle, 40, ctypes.byref(avg)))
    total = by.value + 16 * n
    print("  want %2d -> fmt %d  windowed tiles %d  chunks %d  dict %d  matrix bytes %.3f GB  spmv %9.1f us  physical %.2f TB/s  "
          "CSR-equivalent %.2f TB/s  build %.0f ms  bits %s" % (f, fmt.value, tw.value, ch.value, nd.value, by.value / 1e9, avg.value,
                                                  total / avg.value / 1e6, (12 * op.nnz + 4 * n + 16 * n) / avg.value / 1e6,
                                                  t_build * 1e3, "same" if same else "DIFFER"), flush=True)
    del run
