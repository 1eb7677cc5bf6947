"""Time-to-first-iteration at the BASELINE sizes: what a caller pays before the device loop runs.

  python tools/setup_time.py [nx]         (default 512: variable-coefficient 7-point problem, 134 M rows)

Phases: generate in HBM | first product (builds the storage format) | second product | device -> host CSR copy |
host -> device upload through CsrOperator(indptr, indices, data) | format build of the uploaded copy | CG(...).solve
set-up + 10 passes.  Each phase is bracketed by a device synchronisation.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

import pykrylov_amd as pk
from pykrylov_amd import gallery
from pykrylov_amd.linop import CsrOperator


def timed(label, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-58s %9.1f ms" % (label, dt * 1e3), flush=True)
    return out


def main():
    nx = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    torch.zeros(1, device="cuda")
    A = timed("generate %d^3 variable-coefficient CSR in HBM" % nx, lambda: gallery.poisson3d_varcoef(nx))
    n = A.shape[0]
    x = np.ones(n)
    timed("first product from a host vector (format build + 2 copies)", lambda: A * x)
    timed("second product from a host vector (2 copies of %d MB)" % (8 * n >> 20), lambda: A * x)
    ip, ix, da = timed("device -> host copy of the CSR arrays", A.to_csr_arrays)
    A.free()
    from pykrylov_amd import linop as _linop, _lib as _l
    import ctypes
    timed("   of which: host-side checks (_canonical_csr)", lambda: _linop._canonical_csr(ip, ix, da, (n, n)))
    def raw_upload():
        h = ctypes.c_void_p()
        _l.check(_l.load().mk_csr_create(n, n, len(ix), ip.ctypes.data, ix.ctypes.data, da.ctypes.data, ctypes.byref(h)))
        _l.load().mk_csr_destroy(h)
    timed("   of which: mk_csr_create (allocation + upload of %.1f GB)" % ((ip.nbytes + ix.nbytes + da.nbytes) / 1e9), raw_upload)
    B = timed("CsrOperator(indptr, indices, data): canonical check + upload",
              lambda: CsrOperator(ip, ix, da, (n, n), symmetric=True))
    timed("first product of the uploaded copy", lambda: B * x)
    rhs = B * x
    s = pk.CG(B)
    timed("CG(B).solve(rhs, matvec_max=10): set-up + 10 passes + result copy", lambda: s.solve(rhs, matvec_max=10))
    timed("the same again (format and handles exist)", lambda: s.solve(rhs, matvec_max=10))


if __name__ == "__main__":
    main()
