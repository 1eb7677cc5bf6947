import time, numpy as np, sys
sys.path.insert(0, '.')
from pykrylov_amd import CG, gallery, CsrOperator
op = gallery.poisson2d(1000)
n = op.shape[0]
rhs = op * np.ones(n)
for rep in range(3):
    s = CG(op)
    t0 = time.perf_counter(); s.solve(rhs); dt = time.perf_counter() - t0
    print("config2 solve(): %d products in %.2f ms -> %.0f it/s incl. H2D of rhs, D2H of x and history, Python" % (s.nMatvec, 1e3*dt, s.nMatvec/dt))
# host CSR arrays -> device (matrix upload over PCIe), then solve
indptr, indices, data, shape = gallery.poisson2d_csr(1000)
t0 = time.perf_counter(); op2 = CsrOperator(indptr, indices, data, shape, symmetric=True); dt = time.perf_counter() - t0
print("matrix upload (80 MB host CSR -> HBM, canonicalisation checks): %.1f ms" % (1e3*dt))
