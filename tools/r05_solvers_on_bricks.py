"""Every square solver loop on a 3-D 7-point matrix, windowed formats (MK_SPMV_FORMAT=8: formats 4 / 5) against the brick
march (formats 9 / 10): passes per second with tolerances at zero.  The epilogues of the non-CG loops load vectors inside
the march's pipelined loop; this checks that choosing the march per MATRIX does not cost those loops anything.
    python tools/r05_solvers_on_bricks.py [n] [varcoef]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pykrylov_amd import _lib, gallery                      # noqa: E402
from pykrylov_amd.generic import DeviceRun                  # noqa: E402

lib = _lib.init(0)
m = int(sys.argv[1]) if len(sys.argv) > 1 else 256
var = len(sys.argv) > 2
n = m ** 3
KINDS = [("cg", _lib.MK_CG, dict(abstol=0.0, reltol=0.0, matvec_max=1 << 60, check_curvature=1)),
         ("bicgstab", _lib.MK_BICGSTAB, dict(abstol=0.0, reltol=0.0, matvec_max=1 << 60)),
         ("cgs", _lib.MK_CGS, dict(abstol=0.0, reltol=0.0, matvec_max=1 << 60)),
         ("tfqmr", _lib.MK_TFQMR, dict(abstol=0.0, reltol=0.0, matvec_max=1 << 60)),
         ("minres", _lib.MK_MINRES, dict(shift=0.3, itnlim=1 << 60, rtol=0.0, etol=0.0, window=5)),
         ("symmlq", _lib.MK_SYMMLQ, dict(shift=0.3, has_shift=1, matvec_max=1 << 60, rtol=0.0))]
for fmt in (8, 10):
    op = gallery.poisson3d_varcoef(m, seed=7) if var else gallery.poisson3d(m)
    _lib.check(lib.mk_csr_set_format(op.handle, fmt))
    ones = _lib.DeviceArray.from_numpy(np.ones(n))
    rhs = _lib.DeviceArray(n)
    op.spmv_device(ones.ptr, rhs.ptr)
    row = []
    for name, kind, kw in KINDS:
        run = DeviceRun(op, kind, rhs, None, **kw)
        run.setup()
        run.iterate(6)
        _lib.check(lib.mk_sync())
        t0 = time.perf_counter()
        done = run.iterate(10)                              # (few passes: the non-symmetric loops stay finite on this matrix)
        _lib.check(lib.mk_sync())
        dt = time.perf_counter() - t0
        row.append("%s %.0f" % (name, done / dt))
        run.close()
    import ctypes
    f = ctypes.c_int32()
    _lib.check(lib.mk_csr_format_info(op.handle, ctypes.byref(f), None, None, None, None))
    print("%d^3 %s: format %d: " % (m, "varcoef" if var else "const", f.value) + "  ".join(row) + "   passes/s", flush=True)
    op.free(); ones.free(); rhs.free()
