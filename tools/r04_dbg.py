import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from pykrylov_amd import _lib
from pykrylov_amd.linop import CsrOperator
from pykrylov_amd.generic import DeviceRun
lib = _lib.init(0)
m, n = 400000, 100000
indptr, indices, data = bench.random_tall_csr(m, n)
op = CsrOperator(indptr, indices, data, (m, n))
At = op.T
xs = _lib.DeviceArray.from_numpy(np.random.default_rng(12).standard_normal(n))
rhs = _lib.DeviceArray(m)
op.spmv_device(xs.ptr, rhs.ptr)
for kind in (_lib.MK_LSQR, _lib.MK_LSMR, _lib.MK_CRAIG, _lib.MK_CRAIGMR):
    run = DeviceRun(op, kind, rhs, None, transpose=At, itnlim=1 << 60, damp=0.0, atol=0.0, btol=0.0, conlim=0.0, etol=0.0, window=5)
    for cand in (16, 8, 4):
        run.setup()
        done = run.iterate(cand)
        r = run.finish()
        print(kind, cand, done, r.halted, r.istop, r.residNorm, r.itn, r.nMatvec, r.aux[0], r.aux[1], flush=True)
    run.close()
