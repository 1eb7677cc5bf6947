"""A/B timing of SpMV launch variants on ONE matrix in ONE process (the placement of the buffers, which moves kernel
times by several percent from process to process, is the same for every variant): the variants are visited round robin
and the median over the rounds is reported.      gpurun: python tools/ab_spmv.py [workload] [rounds]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pykrylov_amd import _lib, gallery
from pykrylov_amd.generic import DeviceRun

lib = _lib.init(0)
wl = sys.argv[1] if len(sys.argv) > 1 else "varcoef"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
m = int(os.environ.get("AB_M", "256" if wl.startswith("s27") else "512"))
op = {"varcoef": lambda: gallery.poisson3d_varcoef(m), "const": lambda: gallery.poisson3d(m),
      "s27v": lambda: gallery.stencil27(m, seed=7), "s27c": lambda: gallery.stencil27(m, seed=0)}[wl]()
n = op.shape[0]
ones = _lib.DeviceArray.from_numpy(np.ones(n))
rhs = _lib.DeviceArray(n)
NO_ITER = bool(os.environ.get("AB_NO_ITER"))      # timing ablations that produce garbage: never let the loop see it
if NO_ITER:
    rhs = ones
else:
    op.spmv_device(ones.ptr, rhs.ptr)
run = DeviceRun(op, _lib.MK_CG, rhs, None, abstol=0.0, reltol=0.0, matvec_max=1 << 60, check_curvature=1)
run.setup()
if not NO_ITER:
    run.iterate(20)
P = m * m // 256
variants = [("order2", (2, 0, 0, 0)), ("order2+nt", (2, 0, 0, 1))]
if True:
    for S in (8, 16, 32, 64):
        variants += [("order4 S=%d" % S, (4, S, P, 0)), ("order4 S=%d +nt" % S, (4, S, P, 1))]
    variants += [("order3 S=128 +nt", (3, 128, 0, 1)), ("order0 +nt", (0, 0, 0, 1))]
if os.environ.get("AB_NT_Y"):                      # non-temporal loads of the value stream + stores of the product vector
    variants = [("order2", (2, 0, 0, 0)), ("order2+nt", (2, 0, 0, 1)), ("order4 S=32", (4, 32, P, 0)), ("order4 S=32 +nt", (4, 32, P, 1))]
variants = [(nm, par + (0,)) for nm, par in variants]
res = {k: [] for k, _ in variants}
step = {k: [] for k, _ in variants}
for r in range(rounds):
    for name, (o, s, p, nt, nty) in variants:
        _lib.check(lib.mk_csr_set_tile_order(op.handle, o, s, p, nt))
        avg = ctypes.c_double()
        _lib.check(lib.mk_solver_time_spmv(run.handle, 60, ctypes.byref(avg)))
        res[name].append(avg.value)
        if NO_ITER:
            step[name].append(float("nan"))
            continue
        run.iterate(30)
        step[name].append(run.timing()["iterate_ms"] / 30)
print("workload %s, %d rounds, medians (min .. max):" % (wl, rounds))
for name, _ in variants:
    a, b = np.array(res[name]), np.array(step[name])
    print("  %-20s spmv %7.1f us (%7.1f .. %7.1f)   CG step %6.3f ms" % (name, np.median(a), a.min(), a.max(), np.median(b)))
