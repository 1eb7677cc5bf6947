"""Debug aid (round 5): where does the brick march on a slab differ from the scalar loop over the localised arrays?  Prints the
mismatching rows (plane, in-plane index, stored column order) per rank / format; this is how the storage order of a localised
row -- the order of its GLOBAL columns -- was found.  `python tools/dbg/slab_dbg.py` on a GPU box."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import test_gpu_slab_march as T
from pykrylov_amd import _lib

nx, ny = 128, 8
P = nx * ny
for (nr, rank, planes) in [(4, 2, 7), (4, 3, 7), (3, 1, 20), (4, 0, 7)]:
    for varcoef, fmt in [(False, 9), (True, 10)]:
        lib, world, op = T.build_slab(nx, ny, planes * nr, nr, rank, varcoef, fmt)
        n_local, ncols = int(op.shape[0]), int(op.shape[1])
        A = T.local_oracle(op)
        rng = np.random.default_rng(11 + rank)
        for trial in range(2):
            x = rng.standard_normal(ncols)
            if trial:
                x[::7] = 0.0
                x[5::11] *= 1e300
            xd = _lib.DeviceArray.from_numpy(x)
            yd = _lib.DeviceArray(n_local)
            op.spmv_device(xd.ptr, yd.ptr)
            y, y0 = yd.to_numpy(), A.matvec(x)
            bad = np.nonzero(y != y0)[0]
            print("nr %d rank %d planes %d fmt %d trial %d: %d mismatches" % (nr, rank, planes, T.fmt_of(lib, op), trial, len(bad)))
            if len(bad):
                pl = bad // P
                print("   planes of the mismatches:", np.unique(pl, return_counts=True))
                for r in bad[:6]:
                    lo, hi = A.indptr[r], A.indptr[r + 1]
                    print("   row %d (plane %d, in-plane %d): got %r want %r  cols %s" % (r, r // P, r % P, y[r], y0[r], A.indices[lo:hi] - r))
        op.free()
        lib.mk_comm_destroy()
