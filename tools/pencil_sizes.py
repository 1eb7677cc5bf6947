"""Format 9 (z-marching bricks) against format 4 (windowed pattern tiles) over grid sizes: the CG product kernel timed back to
back (HIP events), and CG passes per second.  Decides MK_PENCIL_MIN_ROWS.   python tools/pencil_sizes.py [nx,ny,nz ...]"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pykrylov_amd import _lib, gallery                      # noqa: E402
from pykrylov_amd.generic import DeviceRun                  # noqa: E402

lib = _lib.init(0)
grids = [tuple(int(t) for t in a.split(",")) for a in sys.argv[1:]] or [(128, 128, 128), (256, 128, 128), (256, 256, 128),
                                                                          (256, 256, 256), (384, 384, 384), (512, 512, 256),
                                                                          (512, 512, 512), (640, 640, 512)]
for g in grids:
    n = g[0] * g[1] * g[2]
    row = []
    for fmt in (4, 9):
        op = gallery.poisson3d(*g)
        _lib.check(lib.mk_csr_set_format(op.handle, fmt))
        ones = _lib.DeviceArray.from_numpy(np.ones(n))
        rhs = _lib.DeviceArray(n)
        op.spmv_device(ones.ptr, rhs.ptr)
        run = DeviceRun(op, _lib.MK_CG, rhs, None, abstol=0.0, reltol=0.0, matvec_max=1 << 60, check_curvature=1)
        run.setup()
        run.iterate(20)
        _lib.check(lib.mk_sync())
        t0 = time.perf_counter()
        run.iterate(200)
        _lib.check(lib.mk_sync())
        dt = time.perf_counter() - t0
        us = run.time_product(0, 100)
        f = ctypes.c_int32()
        _lib.check(lib.mk_csr_format_info(op.handle, ctypes.byref(f), None, None, None, None))
        row.append((f.value, us, 200 / dt))
        run.close()
        op.free()
        ones.free()
        rhs.free()
    print("%4d x %4d x %4d (%9d rows): " % (g + (n,)) + "   ".join("fmt %d product %8.1f us, CG %9.1f it/s" % r for r in row), flush=True)
