#!/usr/bin/env python3
"""Golden capture of what the REAL reference prints with ``show=True``: one small MINRES run and one
run each of LSQR, LSMR and CRAIG.  Runs only in the build container (needs /root/reference; lib2to3 temp copy, nothing of
the reference is written into the repo).  Stored: the inputs (CSR arrays, right-hand sides, keywords) and the captured
stdout, line by line, in tests/golden/show_output.npz.

    OPENBLAS_NUM_THREADS=1 python tests/golden/make_golden_show.py
"""
import contextlib
import io
import os
import sys

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import load_reference, canon, tridiag            # noqa: E402


def capture(fn):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = fn()
    return out, np.array(buf.getvalue().split("\n"))


def main():
    load_reference()
    from pykrylov.linop import LinearOperator
    from pykrylov.minres import Minres
    from pykrylov.lls import LSQRFramework, LSMRFramework, CRAIGFramework

    out = {}
    # ---- MINRES: 2-D Poisson m = 12 (n = 144 > 40, so the reference's print rule is exercised), shift inside the spectrum
    m = 12
    T = tridiag(m)
    A = canon(sp.kron(sp.identity(m), T) + sp.kron(T, sp.identity(m)))
    n = A.shape[0]
    rhs = A @ np.ones(n) - 1.5 * np.ones(n)
    op = LinearOperator(n, n, matvec=lambda v: A @ v, symmetric=True)
    kw = dict(shift=1.5, show=True, check=False, rtol=1.0e-10, etol=0.0, itnlim=200)
    solver = Minres(op)
    _, lines = capture(lambda: solver.solve(rhs, **kw))
    out.update(minres_indptr=A.indptr, minres_indices=A.indices, minres_data=A.data, minres_rhs=rhs,
               minres_shift=1.5, minres_rtol=1.0e-10, minres_itnlim=200, minres_stdout=lines,
               minres_itn=solver.itn, minres_istop=solver.istop, minres_x=solver.x)

    # ---- least squares: seeded 60 x 40 sparse matrix, well conditioned (a strong identity block: the Golub-Kahan vectors
    # keep their orthogonality over the dozen steps the runs take, so the rows do not depend on the summation order of
    # the inner products beyond the printed digits), inconsistent right-hand side
    rng = np.random.default_rng(42)
    B = canon(0.3 * sp.random(60, 40, density=0.15, random_state=np.random.RandomState(7), format="csr")
              + sp.vstack([2.0 * sp.identity(40), sp.csr_matrix((20, 40))]))
    b = B @ rng.standard_normal(40) + 1e-3 * rng.standard_normal(60)
    opB = LinearOperator(40, 60, matvec=lambda v: B @ v, matvec_transp=lambda u: B.T @ u)
    out.update(lls_indptr=B.indptr, lls_indices=B.indices, lls_data=B.data, lls_shape=np.array(B.shape), lls_rhs=b)
    for name, cls, call in (("lsqr", LSQRFramework, lambda s: s.solve(b, show=True, atol=1e-7, btol=1e-7, etol=0.0)),
                            ("lsmr", LSMRFramework, lambda s: s.solve(b, show=True, atol=1e-7, btol=1e-7, etol=0.0)),
                            ("craig", CRAIGFramework, lambda s: s.solve(b, show=True, atol=1e-7, btol=1e-7, etol=0.0))):
        s = cls(opB)
        _, lines = capture(lambda: call(s))
        out[name + "_stdout"] = lines
        out[name + "_itn"] = s.itn if hasattr(s, "itn") else -1
        out[name + "_x"] = s.x
    np.savez_compressed(os.path.join(HERE, "show_output.npz"), **out)
    for k in ("minres", "lsqr", "lsmr", "craig"):
        print("---- %s: %d lines" % (k, len(out[k + "_stdout"])))
        print("\n".join(out[k + "_stdout"][:16]))


if __name__ == "__main__":
    main()
