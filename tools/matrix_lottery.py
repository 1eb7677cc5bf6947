"""Does re-creating the matrix (and its format arrays) in the same process change the product time?  Fixed x, y.
   gpurun: python tools/matrix_lottery.py [K]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pykrylov_amd import _lib, gallery

lib = _lib.init(0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
m = int(os.environ.get("AB_M", "512"))
n = m ** 3
x = _lib.DeviceArray.from_numpy(np.ones(n))
y = _lib.DeviceArray(n)


def timed(fn, reps=20):
    fn(); _lib.check(lib.mk_sync())
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    _lib.check(lib.mk_sync())
    return (time.perf_counter() - t0) / reps * 1e6


ballast = []
for k in range(K):
    op = gallery.poisson3d_varcoef(m)
    t1 = timed(lambda: op.spmv_device(x.ptr, y.ptr))
    # rebuild only the format arrays (the CSR arrays stay where they are)
    _lib.check(lib.mk_csr_set_format(op.handle, 5))
    t2 = timed(lambda: op.spmv_device(x.ptr, y.ptr))
    y2 = _lib.DeviceArray(n)
    t3 = timed(lambda: op.spmv_device(x.ptr, y2.ptr))
    print("matrix %d: product %7.1f us   format rebuilt %7.1f us   other y %7.1f us" % (k, t1, t2, t3), flush=True)
    op.free()
    if k % 2 == 0:
        ballast.append(_lib.DeviceArray((3 << 30) // 8 + 12345 * k))       # shift what the next matrix gets
