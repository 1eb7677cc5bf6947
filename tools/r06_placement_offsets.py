"""Round 6: does the OFFSET between the fused CG kernel's write streams select the fast / slow placement
state?  The loop's vectors are carved from ONE arena reserved before the matrix (mk_arena_reserve), 2 MiB granules, and the
k-th vector is shifted by a chosen skew (MK_ARENA_SKEW) -- allocation order of CG: x, r, p, Ap, p2, dump.  Every configuration
runs in its own process (placement states are per process), twice, with placement draws off; compared with the six-draw
search of the product.      python tools/r06_placement_offsets.py [m=512] > profiles/r06_placement_offsets.txt"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, time, ctypes
sys.path.insert(0, %r)
import numpy as np
from pykrylov_amd import _lib, gallery
from pykrylov_amd.generic import DeviceRun
lib = _lib.init(0)
m = int(sys.argv[1]); n = m ** 3
if os.environ.get("USE_ARENA") == "1":
    extra = sum(int(v) for v in os.environ.get("MK_ARENA_SKEW", "0").split(","))
    _lib.check(lib.mk_arena_reserve(7 * (8 * n + (8 << 20)) + extra))
op = gallery.poisson3d(m)
ones = _lib.DeviceArray.from_numpy(np.ones(n)); rhs = _lib.DeviceArray(n)
op.spmv_device(ones.ptr, rhs.ptr)
run = DeviceRun(op, _lib.MK_CG, rhs, None, abstol=0.0, reltol=0.0, matvec_max=1 << 60, check_curvature=1)
run.setup(); run.iterate(30); _lib.check(lib.mk_sync())
t0 = time.perf_counter(); run.iterate(300); _lib.check(lib.mk_sync()); dt = time.perf_counter() - t0
us = run.time_product(0, 100)
print("RESULT %%.4f ms/pass  %%.1f it/s  fused product %%.1f us  draws %%d" %% (1e3 * dt / 300, 300 / dt, us, run.placement["count"]))
''' % ROOT


def run(m, env):
    e = dict(os.environ, **env)
    p = subprocess.run([sys.executable, "-c", CHILD, str(m)], capture_output=True, text=True, env=e, timeout=600)
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
    return line[0][7:] if line else "FAILED " + p.stderr[-300:]


def big_skews(m):
    """second series: skews of megabytes to a gigabyte between the vectors (the channel / bank hash takes upper address bits)"""
    M = 1 << 20
    print("# second series: large skews")
    for sk in ([0, 16 * M, 32 * M, 48 * M, 64 * M], [0, 33 * M, 66 * M, 99 * M, 132 * M], [0, 100 * M, 200 * M, 300 * M, 400 * M],
               [0, 0, 0, 0, 512 * M], [0, 0, 256 * M, 0, 512 * M], [0, 341 * M, 682 * M, 1023 * M, 1364 * M], [0, 1000 * M, 0, 0, 0],
               [0, 7 * M + 4096, 14 * M + 8192, 21 * M + 12288, 28 * M + 16384]):
        for rep in range(2):
            print("arena, skew %-50s : %s" % (",".join(str(v) for v in sk),
                                               run(m, {"MK_PLACEMENT_DRAWS": "1", "USE_ARENA": "1", "MK_ARENA_SKEW": ",".join(str(v) for v in sk)})),
                  flush=True)
    for rep in range(3):
        print("separate allocations, draws off                                   : %s" % run(m, {"MK_PLACEMENT_DRAWS": "1"}), flush=True)


if __name__ == "__main__":
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    if len(sys.argv) > 2 and sys.argv[2] == "big":
        big_skews(m)
        sys.exit(0)
    K = 1024
    print("# CG %d^3, constant coefficients, fused passes; per configuration two processes" % m)
    for rep in range(3):
        print("product default (6 automatic draws)            : %s" % run(m, {}), flush=True)
    for rep in range(3):
        print("separate allocations, draws off                : %s" % run(m, {"MK_PLACEMENT_DRAWS": "1"}), flush=True)
    # skews: (x, r, p, Ap, p2): shift p2 and p against x
    for sk in ([0, 0, 0, 0, 0], [0, 0, 0, 0, 4 * K], [0, 0, 0, 0, 8 * K], [0, 0, 0, 0, 12 * K], [0, 0, 0, 0, 64 * K],
               [0, 0, 0, 0, 256 * K], [0, 0, 0, 0, 1024 * K], [0, 0, 4 * K, 0, 8 * K], [0, 0, 8 * K, 0, 16 * K],
               [0, 4 * K, 8 * K, 12 * K, 16 * K], [0, 1 * K, 2 * K, 3 * K, 4 * K], [0, 256, 512, 768, 1024],
               [0, 64 * K, 128 * K, 192 * K, 256 * K], [0, 512 * K, 1024 * K, 1536 * K, 0]):
        for rep in range(2):
            print("arena, skew %-34s : %s" % (",".join(str(v) for v in sk),
                                               run(m, {"MK_PLACEMENT_DRAWS": "1", "USE_ARENA": "1", "MK_ARENA_SKEW": ",".join(str(v) for v in sk)})),
                  flush=True)
