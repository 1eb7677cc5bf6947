import ctypes, numpy as np, sys
sys.path.insert(0, '.')
from pykrylov_amd import CG, BiCGSTAB, Minres, Symmlq, gallery, DiagonalOperator, IdentityOperator, _lib
from pykrylov_amd.lls import LSQRFramework
hip = ctypes.CDLL("libamdhip64.so")
def free_mb():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)); return f.value / 2**20
_lib.init()
op = gallery.poisson2d(200); n = op.shape[0]; rhs = op * np.ones(n)
base = None
for rep in range(6):
    for _ in range(30):
        CG(op, precon=DiagonalOperator(np.full(n, 0.25))).solve(rhs, matvec_max=20)
        BiCGSTAB(op).solve(rhs, matvec_max=20)
        Minres(op - 0.5 * IdentityOperator(n)).solve(rhs, show=False, check=False, itnlim=10)
        Symmlq(op).solve(rhs, matvec_max=12)
        LSQRFramework(op).solve(rhs, itnlim=5)
        o2 = gallery.poisson2d(50); o2.T; (2.0 * o2) * np.ones(2500); o2.free()
    f = free_mb()
    base = base or f
    print("after %3d rounds: free HBM %.1f MB (delta %.1f MB)" % ((rep + 1) * 30, f, f - base))
