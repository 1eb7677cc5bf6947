"""Does a plain write probe predict how fast the product kernel runs when it writes into a given allocation?
One process, one matrix, K candidate vectors: time of a fill (hipMemsetAsync), of a device copy into it, and of the
product `y = A x` with y = the candidate.     gpurun: python tools/y_lottery.py [K]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pykrylov_amd import _lib, gallery

lib = _lib.init(0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m = int(os.environ.get("AB_M", "512"))
op = gallery.poisson3d_varcoef(m)
n = op.shape[0]
x = _lib.DeviceArray.from_numpy(np.ones(n))
src = _lib.DeviceArray(n)
cands = [_lib.DeviceArray(n) for _ in range(K)]


def timed(fn, reps=10):
    fn(); _lib.check(lib.mk_sync())
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    _lib.check(lib.mk_sync())
    return (time.perf_counter() - t0) / reps * 1e6


print("candidate      address     fill us   copy-into us   product us   product reading it as x")
for k, y in enumerate(cands):
    f = timed(lambda: _lib.check(lib.mk_memset(y.ptr, 0, 8 * n)))
    c = timed(lambda: _lib.check(lib.mk_memcpy_d2d(y.ptr, src.ptr, 8 * n)))
    p = timed(lambda: op.spmv_device(x.ptr, y.ptr), 20)
    _lib.check(lib.mk_memset(y.ptr, 0, 8 * n))
    q = timed(lambda: op.spmv_device(y.ptr, src.ptr), 20)
    print("  y%-2d  %#16x   %8.1f   %8.1f      %8.1f     %8.1f" % (k, y.ptr, f, c, p, q), flush=True)

# ---- large offsets inside ONE slab: does the state change along it?
slab_bytes = 6 << 30
slab = _lib.DeviceArray(slab_bytes // 8, zero=True)
class V:
    def __init__(self, ptr): self.ptr = ptr
print("y = slab + offset:")
steps = [k * (96 << 20) for k in range(0, 52)]
out = []
for off in steps:
    if off + 8 * n > slab_bytes:
        break
    out.append((off >> 20, timed(lambda: op.spmv_device(x.ptr, slab.ptr + off), 12)))
print("  " + "  ".join("%d MiB: %.0f" % o for o in out), flush=True)
fine = [k * (2 << 20) for k in range(0, 48)]
out = [(off >> 20, timed(lambda: op.spmv_device(x.ptr, slab.ptr + off), 12)) for off in fine]
print("  " + "  ".join("%d MiB: %.0f" % o for o in out), flush=True)
