"""HBM leak check of the wide storage formats: create / build every mode / multiply / solve / transpose / free in a loop."""
import ctypes, numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pykrylov_amd import CG, Minres, gallery, _lib
hip = ctypes.CDLL("libamdhip64.so")
def free_mb():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)); return f.value / 2**20
lib = _lib.init()
base = None
for rep in range(int(os.environ.get("REPS", "6"))):
    for _ in range(10):
        for seed in (0, 7):
            op = gallery.stencil27(64, 16, 6, seed=seed)
            n = op.shape[0]
            x = np.ones(n)
            for want in [int(w) for w in os.environ.get("WANTS", "6,7,8,0,-1").split(",")]:
                _lib.check(lib.mk_csr_set_format(op.handle, want))
                y = op * x
            if not os.environ.get("NO_SOLVE"):
                CG(op).solve(y, matvec_max=10)
                Minres(op).solve(y, show=False, check=False, itnlim=5)
            if not os.environ.get("NO_T"):
                op.T * x
                (2.0 * op) * x
            op.free()
        big = np.random.default_rng(1).standard_normal(9_000_000)          # staged host copies
        d = _lib.DeviceArray.from_numpy(big); d.to_numpy(); d.free()
    import gc; gc.collect(); _lib.check(lib.mk_sync())
    f = free_mb()
    base = base or f
    print("after %3d rounds: free HBM %.1f MB (delta %.1f MB)" % ((rep + 1) * 10, f, f - base), flush=True)
