"""Device-memory leak check of the round-4 paths: column blocks (automatic, resident blocks), pair / stepped scattered products (carry
buffer), the vector arena, row-range download, solver-borrowed operators destroyed before their solver."""
import ctypes, numpy as np, sys, gc
sys.path.insert(0, '.')
import bench
from pykrylov_amd import BiCGSTAB, CsrOperator, _lib, gallery
from pykrylov_amd.lls import LSQRFramework
from pykrylov_amd.generic import DeviceRun
hip = ctypes.CDLL("libamdhip64.so")
def free_mb():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)); return f.value / 2**20
lib = _lib.init()
rng = np.random.default_rng(0)
base = None
for rep in range(5):
    for _ in range(3):
        # long rows over a long x: automatic column blocks, resident
        m, ncols, k = 40000, 2200000, 16
        w = ncols // k
        cols = (rng.integers(0, w, size=(m, k)) + np.arange(k)[None, :] * w).reshape(-1)
        A = CsrOperator(np.arange(m + 1) * k, cols, rng.standard_normal(m * k), (m, ncols))
        A * rng.standard_normal(ncols)
        A.csr_rows(10, 2000)
        A.free()
        # tall 5-per-row matrix, stepped pair kernel + its transpose in the lls loop
        ip, ix, dv = bench.random_tall_csr(2300000, 700000)
        T = CsrOperator(ip, ix, dv, (2300000, 700000))
        LSQRFramework(T).solve(T * np.ones(700000), itnlim=3)
        T.free()
        # square scattered matrix: pair kernel in a solver, operator destroyed BEFORE the solver
        op = gallery.random_diagdom(700000, seed=rep + 1)
        rhs = op * np.ones(700000)
        run = DeviceRun(op, _lib.MK_BICGSTAB, rhs, None, abstol=0.0, reltol=1e-6, matvec_max=40)
        run.setup(); run.iterate(4)
        op.free()                                           # deferred: the solver still borrows it
        run.iterate(4); run.close()
        # arena
        _lib.check(lib.mk_arena_reserve(64 << 20))
        o2 = gallery.poisson2d(300)
        BiCGSTAB(o2).solve(o2 * np.ones(90000), matvec_max=10)
        o2.free()
        _lib.check(lib.mk_arena_reserve(0))
    gc.collect()
    f = free_mb()
    base = base or f
    print("after %2d rounds: free HBM %.1f MB (delta %.1f MB)" % ((rep + 1) * 3, f, f - base), flush=True)
assert abs(f - base) < 64, "device memory drifts"
print("no drift")
