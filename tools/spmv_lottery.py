"""Which allocation decides the product kernel's time?  One process: K copies of the 512^3 variable-coefficient matrix
(each with its own format arrays) x K pairs of vectors; the plain product `y = A x` is timed for every combination.
   gpurun: python tools/spmv_lottery.py [K]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pykrylov_amd import _lib, gallery

lib = _lib.init(0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
m = int(os.environ.get("AB_M", "512"))
ops = [gallery.poisson3d_varcoef(m) for _ in range(K)]
n = ops[0].shape[0]
vecs = []
for k in range(K):
    x = _lib.DeviceArray.from_numpy(np.ones(n))
    y = _lib.DeviceArray(n)
    vecs.append((x, y))


def timed(op, x, y, reps=20):
    op.spmv_device(x.ptr, y.ptr)
    _lib.check(lib.mk_sync())
    t0 = time.perf_counter()
    for _ in range(reps):
        op.spmv_device(x.ptr, y.ptr)
    _lib.check(lib.mk_sync())
    return (time.perf_counter() - t0) / reps * 1e6


print("vector addresses: " + "  ".join("x%d=%#x y%d=%#x" % (k, x.ptr, k, y.ptr) for k, (x, y) in enumerate(vecs)))
print("rows: matrix copy, columns: vector pair; product time in us (plain epilogue, %d^3)" % m)
for i, op in enumerate(ops):
    print("  A%d: " % i + "  ".join("%7.1f" % timed(op, x, y) for x, y in vecs), flush=True)
print("x of pair i (rows) with y of pair j (columns), matrix A0 and A1:")
for a in (0, 1):
    for i in range(K):
        print("  A%d x%d: " % (a, i) + "  ".join("%7.1f" % timed(ops[a], vecs[i][0], vecs[j][1]) for j in range(K)), flush=True)
# offsets inside one over-sized buffer: y at byte offsets 0 .. 2 MiB in 256 KiB steps, then 4 KiB steps
big = _lib.DeviceArray(n + (4 << 20) // 8)
x0 = vecs[0][0]
class V:                                                     # a view into `big`
    def __init__(self, ptr): self.ptr = ptr
print("y = big + offset (matrix A0 / A1, x0):")
for off in [0, 4096, 8192, 65536, 262144, 524288, 1048576, 2097152, 3145728]:
    print("  offset %8d: A0 %7.1f   A1 %7.1f" % (off, timed(ops[0], x0, V(big.ptr + off)), timed(ops[1], x0, V(big.ptr + off))), flush=True)
