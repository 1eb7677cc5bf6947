"""Round 6: the brick march on general grid sides against the windowed formats.  Per grid and format: the CG product kernel
timed back to back (HIP events), CG passes per second, and the pass's physical iteration fraction of 8 TB/s -- (bytes of the
storage format + 72 n fused / 80 n three-kernel vector bytes) per pass.   python tools/r06_march_sizes.py [v:]nx,ny,nz ...
(prefix v: = variable coefficients; ny = 0: the 2-D 5-point matrix of side nx)."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pykrylov_amd import _lib, gallery                      # noqa: E402
from pykrylov_amd.generic import DeviceRun                  # noqa: E402

lib = _lib.init(0)
args = sys.argv[1:] or ["500,500,500", "384,300,200", "512,512,512", "v:500,500,500", "2000,0,0", "4000,0,0"]
steps = int(os.environ.get("STEPS", "200"))
for a in args:
    var = a.startswith("v:")
    g = tuple(int(t) for t in a.split(":")[-1].split(","))
    two_d = g[1] == 0
    n = g[0] * g[0] if two_d else g[0] * g[1] * g[2]
    row = []
    for fmt in ((5, 11) if var else (4, 9)):
        op = gallery.poisson2d(g[0]) if two_d else (gallery.poisson3d_varcoef(*g, seed=7) if var else gallery.poisson3d(*g))
        _lib.check(lib.mk_csr_set_format(op.handle, fmt))
        ones = _lib.DeviceArray.from_numpy(np.ones(n))
        rhs = _lib.DeviceArray(n)
        op.spmv_device(ones.ptr, rhs.ptr)
        run = DeviceRun(op, _lib.MK_CG, rhs, None, abstol=0.0, reltol=0.0, matvec_max=1 << 60, check_curvature=1)
        run.setup()
        run.iterate(20)
        _lib.check(lib.mk_sync())
        t0 = time.perf_counter()
        run.iterate(steps)
        _lib.check(lib.mk_sync())
        dt = time.perf_counter() - t0
        us = run.time_product(0, 100)
        f, fused = ctypes.c_int32(), ctypes.c_int32()
        mb = ctypes.c_int64()
        _lib.check(lib.mk_csr_format_info(op.handle, ctypes.byref(f), None, None, None, ctypes.byref(mb)))
        _lib.check(lib.mk_solver_fused(run.handle, ctypes.byref(fused)))
        info = (ctypes.c_int64 * 12)()
        _lib.check(lib.mk_csr_march_info(op.handle, info, 12))
        bytes_pass = mb.value + (72 if fused.value else 80) * n
        its = steps / dt
        prod_bytes = mb.value + (48 if fused.value else 16) * n
        row.append("fmt %2d%s%s product %8.1f us (%.2f)  CG %8.1f it/s  pass %.2f of 8 TB/s" %
                   (f.value, " gen" if info[9] else "    ", " fused" if fused.value else "      ", us, prod_bytes / (us * 1e-6) / 8e12,
                    its, bytes_pass * its / 8e12))
        run.close()
        op.free()
        ones.free()
        rhs.free()
    print("%-18s %10d rows: " % (a, n) + "  |  ".join(row), flush=True)
