"""Grid sizes of the product kernel (and of the update kernels) swept inside ONE process on one matrix: the solver is set
up again for every grid (the partial-sum bookkeeping follows the grid), the matrix and its format stay.
   gpurun: MK_GRID_DYNAMIC=1 python tools/grid_sweep.py [varcoef|const|s27v|s27c] [rounds]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MK_GRID_DYNAMIC"] = "1"
import numpy as np
from pykrylov_amd import _lib, gallery
from pykrylov_amd.generic import DeviceRun

lib = _lib.init(0)
wl = sys.argv[1] if len(sys.argv) > 1 else "varcoef"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
m = int(os.environ.get("AB_M", "256" if wl.startswith("s27") else "512"))
op = {"varcoef": lambda: gallery.poisson3d_varcoef(m), "const": lambda: gallery.poisson3d(m),
      "s27v": lambda: gallery.stencil27(m, seed=7), "s27c": lambda: gallery.stencil27(m, seed=0)}[wl]()
n = op.shape[0]
ones = _lib.DeviceArray.from_numpy(np.ones(n))
rhs = _lib.DeviceArray(n)
op.spmv_device(ones.ptr, rhs.ptr)
grids = [None, 1024, 1280, 1536, 1792, 2048]
streams = [None, 256, 1024]
res = {}
for r in range(rounds):
    for g in grids:
        for sg in (streams if g is None else [None]):
            for k, v in (("MK_GRID_SPMV", g), ("MK_GRID_STREAM", sg)):
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = str(v)
            run = DeviceRun(op, _lib.MK_CG, rhs, None, abstol=0.0, reltol=0.0, matvec_max=1 << 60, check_curvature=1)
            run.setup()
            run.iterate(10)
            avg = ctypes.c_double()
            _lib.check(lib.mk_solver_time_spmv(run.handle, 40, ctypes.byref(avg)))
            run.iterate(30)
            res.setdefault((g, sg), []).append((avg.value, run.timing()["iterate_ms"] / 30))
            run.close()
print("workload %s (m = %d), medians over %d rounds" % (wl, m, rounds))
for (g, sg), v in res.items():
    a = np.array(v)
    print("  spmv grid %-8s stream grid %-8s  spmv %8.1f us   CG step %6.3f ms" % (g or "default", sg or "default", np.median(a[:, 0]), np.median(a[:, 1])))
