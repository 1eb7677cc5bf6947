"""Per-pass wall time of the fused CG loop at 512^3 (one synchronised pass at a time): does the time alternate with the p
buffer in use (even passes read p / write p2, odd passes the reverse)?   python tools/r05_pass_times.py [varcoef]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pykrylov_amd import _lib, gallery                      # noqa: E402
from pykrylov_amd.generic import DeviceRun                  # noqa: E402

lib = _lib.init(0)
m = 512
op = gallery.poisson3d_varcoef(m, seed=7) if len(sys.argv) > 1 else gallery.poisson3d(m)
n = m ** 3
ones = _lib.DeviceArray.from_numpy(np.ones(n))
rhs = _lib.DeviceArray(n)
op.spmv_device(ones.ptr, rhs.ptr)
run = DeviceRun(op, _lib.MK_CG, rhs, None, abstol=0.0, reltol=0.0, matvec_max=1 << 60, check_curvature=1)
run.setup()
run.iterate(10)
_lib.check(lib.mk_sync())
ts = []
for k in range(40):
    t0 = time.perf_counter()
    run.iterate(1)
    _lib.check(lib.mk_sync())
    ts.append(1e6 * (time.perf_counter() - t0))
print("per-pass us:", " ".join("%.0f" % t for t in ts))
print("even passes mean %.0f us, odd passes mean %.0f us" % (np.mean(ts[0::2]), np.mean(ts[1::2])))
t0 = time.perf_counter()
run.iterate(200)
_lib.check(lib.mk_sync())
print("200 passes back to back: %.1f us per pass" % (1e6 * (time.perf_counter() - t0) / 200))
