"""Does the SpMV time depend on WHERE the driver places the buffers?  Re-create the operator and the solver several
times inside one process and time the fused SpMV kernel each time (gpurun: python tools/alloc_lottery.py)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pykrylov_amd import _lib, gallery
from pykrylov_amd.generic import DeviceRun

lib = _lib.init(0)
m = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for rep in range(6):
    op = gallery.poisson3d_varcoef(m)
    n = op.shape[0]
    ones = _lib.DeviceArray.from_numpy(np.ones(n))
    rhs = _lib.DeviceArray(n)
    op.spmv_device(ones.ptr, rhs.ptr)
    run = DeviceRun(op, _lib.MK_CG, rhs, None, abstol=0.0, reltol=0.0, matvec_max=1 << 60, check_curvature=1)
    run.setup()
    run.iterate(20)
    out = []
    for k in range(3):
        avg = ctypes.c_double()
        _lib.check(lib.mk_solver_time_spmv(run.handle, 100, ctypes.byref(avg)))
        out.append(avg.value)
    run.iterate(100)
    t = run.timing()["iterate_ms"] / 100
    print("rep %d: spmv %s us, CG step %.3f ms" % (rep, " ".join("%.1f" % v for v in out), t), flush=True)
    run.close(); op.free(); ones.free(); rhs.free()
