"""How much does the CG pass time vary between vector sets inside ONE process (one matrix)?  K solver objects, each with
its own four vectors, all alive; 40 passes timed on each, twice.      gpurun: python tools/pass_lottery.py [K]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pykrylov_amd import _lib, gallery
from pykrylov_amd.generic import DeviceRun

lib = _lib.init(0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
op = gallery.poisson3d_varcoef(512)
n = op.shape[0]
ones = _lib.DeviceArray.from_numpy(np.ones(n))
rhs = _lib.DeviceArray(n)
op.spmv_device(ones.ptr, rhs.ptr)
runs, spacers = [], []
for k in range(K):
    run = DeviceRun(op, _lib.MK_CG, rhs, None, abstol=0.0, reltol=0.0, matvec_max=1 << 60, check_curvature=1)
    run.setup()
    run.iterate(10)
    runs.append(run)
    spacers.append(_lib.DeviceArray(((200 + 77 * k) << 20) // 8))
for rnd in range(2):
    for k, run in enumerate(runs):
        avg = ctypes.c_double()
        _lib.check(lib.mk_solver_time_spmv(run.handle, 30, ctypes.byref(avg)))
        run.iterate(40)
        print("round %d set %d: product %7.1f us   pass %6.3f ms" % (rnd, k, avg.value, run.timing()["iterate_ms"] / 40), flush=True)
