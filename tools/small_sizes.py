import time, numpy as np, sys
sys.path.insert(0, '.')
from pykrylov_amd import CG, gallery, _lib
from pykrylov_amd.generic import DeviceRun
lib = _lib.init()
for m in (34, 100, 316):
    op = gallery.poisson2d(m); n = m*m
    ones = _lib.DeviceArray.from_numpy(np.ones(n)); rhs = _lib.DeviceArray(n); op.spmv_device(ones.ptr, rhs.ptr)
    run = DeviceRun(op, _lib.MK_CG, rhs, None, abstol=0.0, reltol=0.0, matvec_max=1<<60, check_curvature=1)
    run.setup(); run.iterate(200); lib.mk_sync()
    t0=time.perf_counter(); run.iterate(4000); lib.mk_sync(); dt=time.perf_counter()-t0
    print("n=%7d: %.2f us per CG pass (3 kernels)" % (n, 1e6*dt/4000))
    run.close()
