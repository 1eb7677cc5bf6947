"""In-process A/B of the non-temporal product-vector stores in the MINRES / BiCGSTAB epilogues (round 4) at 512^3: the same
matrix, the same solver object, `mk_csr_set_tile_order(..., nontemporal)` toggled between runs (placement cancels)."""
import ctypes, sys, time
import numpy as np
sys.path.insert(0, '.')
from pykrylov_amd import _lib, gallery
from pykrylov_amd.generic import DeviceRun
lib = _lib.init(0)
for name, op in (("poisson3d-512 (fmt 4)", gallery.poisson3d(512)), ("poisson3d-512-varcoef (fmt 5)", gallery.poisson3d_varcoef(512, seed=7))):
    n = op.shape[0]
    ones = _lib.DeviceArray.from_numpy(np.ones(n)); rhs = _lib.DeviceArray(n)
    op.spmv_device(ones.ptr, rhs.ptr)
    for kind, kw, label in ((_lib.MK_MINRES, dict(itnlim=1 << 60, rtol=0.0, etol=0.0, window=5), "MINRES"),
                            (_lib.MK_BICGSTAB, dict(abstol=0.0, reltol=0.0, matvec_max=1 << 60), "BiCGSTAB")):
        run = DeviceRun(op, kind, rhs, None, **kw)
        for rep in range(2):
            for nt in (0, 1):
                _lib.check(lib.mk_csr_set_tile_order(op.handle, -1, 0, 0, nt))
                run.setup(); run.iterate(4)
                _lib.check(lib.mk_sync()); t0 = time.perf_counter()
                done = run.iterate(12)
                _lib.check(lib.mk_sync()); dt = (time.perf_counter() - t0) / max(1, done)
                us = [run.time_product(w, 20) for w in ((0, 1) if kind == _lib.MK_BICGSTAB else (0,))]
                print("%-30s %-8s nt=%d: pass %.3f ms, product kernel(s) %s us" % (name, label, nt, 1e3 * dt, [round(u, 1) for u in us]), flush=True)
        run.close()
    op.free()
