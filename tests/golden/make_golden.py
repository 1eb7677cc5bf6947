#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference.

Runs only in the build container (needs /root/reference).  Nothing from the
reference is copied into the repo: the package is converted with lib2to3 into a
throw-away temp dir, imported from there, run on the inputs below, and only the
inputs/outputs (plain arrays and scalars) are written to ``*.npz``.

    OPENBLAS_NUM_THREADS=1 python tests/golden/make_golden.py

Why OPENBLAS_NUM_THREADS=1: a multi-threaded ddot changes the summation order
of ``np.dot`` and therefore the reference's own bits.

Fixture inventory:
  G1  cg_1138bus.npz        MatrixMarket -> CSR arrays + CG default trace
  G2  cg_poisson2d.npz      2-D Poisson m in {10,20,100}: CSR + CG traces
  G3  cg_poisson1d.npz      1-D Poisson n=100 (doc/source/introduction.rst:46-48)
  G4  nonsym_*.npz          BiCGSTAB / CGS / TFQMR on jpwh_991 + random matrix,
                            with the full reduction trace (every dot / norm)
  G5  minres_poisson2d.npz  MINRES m=100, shift in {0,1.5}, check in {F,T}
  G6  symmlq_poisson2d.npz  SYMMLQ m=30/100 (matvec attribute injected)
  G7  lls_*.npz             LSQR / LSMR / CRAIG / CRAIG-MR on seeded matrices
  G8  large_summaries.npz   n=1e6 CG summary + integer checksums of the matrices
  G9  api_contract.npz      LinearOperator protocol facts (dtype promotion, counters)
  G10 precon_jacobi.npz     all six solvers with a DiagonalOperator preconditioner;
                            `make_golden.py --only-precon` regenerates just this file and G11
  G11 composed_ops.npz      solvers on operators built with the reference's algebra (A + D, A - sigma*I,
                            alpha*A)
  G12 lls_precon.npz        LSQR / LSMR / CRAIG / CRAIG-MR with diagonal preconditioners M, N
"""
import contextlib
import hashlib
import io
import os
import shutil
import subprocess
import sys
import tempfile

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")

import numpy as np
import scipy.io
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


# --------------------------------------------------------------------------- #
# reference import (lib2to3 copy in a temp dir; never written into the repo)
# --------------------------------------------------------------------------- #
def load_reference():
    tmp = tempfile.mkdtemp(prefix="pkref_")
    shutil.copytree(os.path.join(REF, "pykrylov"), os.path.join(tmp, "pykrylov"))
    subprocess.run([sys.executable, "-W", "ignore", "-m", "lib2to3", "-w", "-n", "pykrylov"],
                   cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    # aliases removed from modern NumPy but used by tools/types.py:4-11, linop.py:33
    np.int, np.float, np.complex = int, float, complex
    np.float_, np.complex_ = np.float64, np.complex128
    sys.path.insert(0, tmp)
    import pykrylov  # noqa: F401
    return tmp


# --------------------------------------------------------------------------- #
# input matrices (SciPy builds them; the product's own builders are later
# checked bit-exactly against the arrays stored here)
# --------------------------------------------------------------------------- #
def canon(A):
    A = sp.csr_matrix(A)
    A.sum_duplicates()
    A.sort_indices()
    A = sp.csr_matrix((A.data.astype(np.float64), A.indices.astype(np.int32),
                       A.indptr.astype(np.int32)), shape=A.shape)
    return A


def tridiag(n):
    return sp.diags([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1], format="csr")


def poisson1d(n):
    return canon(tridiag(n))


def poisson2d(m):
    T, I = tridiag(m), sp.identity(m, format="csr")
    return canon(sp.kron(I, T) + sp.kron(T, I))


def poisson3d(m):
    T, I = tridiag(m), sp.identity(m, format="csr")
    return canon(sp.kron(sp.kron(I, I), T) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(T, I), I))


def random_diagdom(n, seed=1, k=4):
    """k random off-diagonals per row, diagonal = sum|off| + 1."""
    rng = np.random.default_rng(seed)
    rows = np.repeat(np.arange(n), k)
    cols = rng.integers(0, n, size=n * k)
    vals = rng.standard_normal(n * k)
    offd = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()
    offd.sum_duplicates()
    offd.sort_indices()
    offd.setdiag(0.0)
    offd.eliminate_zeros()
    # row sums of |a_ij| taken left to right in ascending column order (csr_matvec with ones)
    diag = abs(offd) @ np.ones(n) + 1.0
    return canon(offd + sp.diags(diag))


def mm_csr(name):
    return canon(scipy.io.mmread(os.path.join(REF, "examples", name)))


def csr_arrays(A, prefix):
    return {prefix + "indptr": A.indptr, prefix + "indices": A.indices, prefix + "data": A.data,
            prefix + "shape": np.array(A.shape, dtype=np.int64)}


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


# --------------------------------------------------------------------------- #
# reduction-trace recorder: every dot / norm result in call order
# --------------------------------------------------------------------------- #
class _LinalgProxy:
    def __init__(self, log):
        self._log = log

    def norm(self, a, *args, **kw):
        v = np.linalg.norm(a, *args, **kw)
        self._log.append(float(v))
        return v

    def __getattr__(self, k):
        return getattr(np.linalg, k)


def _exact_dot(a, b):
    return np.float64(np.dot(a.astype(np.longdouble), b.astype(np.longdouble)))


class _NumpyProxy:
    """Stands in for the module-level name ``np`` inside one solver module."""

    def __init__(self, log, exact=False):
        self._log = log
        self._exact = exact
        self.linalg = _LinalgProxy(log)

    def dot(self, a, b):
        v = _exact_dot(a, b) if self._exact else np.dot(a, b)
        self._log.append(float(v))
        return v

    def __getattr__(self, k):
        return getattr(np, k)


@contextlib.contextmanager
def traced(module, exact=False):
    """Record dots/norms made through ``np.*`` or bare ``dot``/``norm`` names of *module*."""
    log = []
    saved = {}
    if hasattr(module, "np"):
        saved["np"] = module.np
        module.np = _NumpyProxy(log, exact)
    if hasattr(module, "dot"):
        saved["dot"] = module.dot

        def _dot(a, b):
            v = np.dot(a, b)
            log.append(float(v))
            return v
        module.dot = _dot
    if hasattr(module, "norm"):
        saved["norm"] = module.norm

        def _norm(a, *args, **kw):
            v = np.linalg.norm(a, *args, **kw)
            log.append(float(v))
            return v
        module.norm = _norm
    try:
        yield log
    finally:
        for k, v in saved.items():
            setattr(module, k, v)


def quiet(fn, *a, **kw):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **kw)


def main():
    tmp = load_reference()
    from pykrylov.linop import LinearOperator
    import pykrylov.cg.cg as m_cg
    import pykrylov.bicgstab.bicgstab as m_bicgstab
    import pykrylov.cgs.cgs as m_cgs
    import pykrylov.tfqmr.tfqmr as m_tfqmr
    import pykrylov.minres.minres as m_minres
    import pykrylov.symmlq.symmlq as m_symmlq
    import pykrylov.lls.lsqr as m_lsqr
    import pykrylov.lls.lsmr as m_lsmr
    import pykrylov.lls.craig as m_craig
    import pykrylov.lls.craigmr as m_craigmr
    from pykrylov.gallery import Poisson1dMatvec, Poisson2dMatvec

    def csr_op(A, symmetric=False):
        At = A.T.tocsr()
        return LinearOperator(A.shape[1], A.shape[0], matvec=lambda v: A @ v,
                              matvec_transp=lambda u: At @ u, symmetric=symmetric)

    def save(name, **arrs):
        path = os.path.join(HERE, name)
        np.savez_compressed(path, **arrs)
        print("%-28s %8.1f KB" % (name, os.path.getsize(path) / 1024.0))

    # ---------------- G10: diagonal (Jacobi) preconditioning -------------- #
    def g10():
        from pykrylov.linop import DiagonalOperator
        out = {}
        m = 30
        n = m * m
        A = canon(poisson2d(m) + sp.diags(np.linspace(0.5, 6.0, n)))       # SPD, varying diagonal
        d = 1.0 / A.diagonal()
        out.update(csr_arrays(A, "spd_A_"))
        out["spd_d"] = d
        rhs = A @ np.ones(n)
        out["spd_rhs"] = rhs
        for gtag in ("zero", "guess"):
            kw = {} if gtag == "zero" else {"guess": 1.0 + np.arange(n)}
            with traced(m_cg) as log:
                s = m_cg.CG(csr_op(A, True), precon=DiagonalOperator(d))
                s.solve(rhs, **kw)
            k = "cg_%s_" % gtag
            out.update({k + "x": s.x, k + "nMatvec": s.nMatvec, k + "residNorm": s.residNorm,
                        k + "residNorm0": s.residNorm0, k + "residHistory": np.array(s.residHistory),
                        k + "converged": s.converged, k + "trace": np.array(log)})
        for shift in (0.0, 1.5):
            b = rhs - shift * np.ones(n)
            with traced(m_minres) as log:
                s = m_minres.Minres(csr_op(A, True))
                quiet(s.solve, b, precon=DiagonalOperator(d), shift=shift, show=False, check=False, etol=0.0,
                      rtol=1e-10)
            k = "minres_s%g_" % shift
            out.update({k + "rhs": b, k + "x": s.x, k + "istop": s.istop, k + "itn": s.itn,
                        k + "residHistory": np.array(s.residHistory), k + "rnorm": s.rnorm,
                        k + "Anorm": s.Anorm, k + "Acond": s.Acond, k + "Arnorm": s.Arnorm,
                        k + "ynorm": s.ynorm, k + "residNorm0": s.residNorm0, k + "trace": np.array(log)})
            op = csr_op(A, True)
            with traced(m_symmlq) as log:
                s = m_symmlq.Symmlq(op, precon=DiagonalOperator(d))
                s.matvec = lambda v, op=op: op * v        # symmlq.py:162 calls a missing attribute
                kw = {} if shift == 0.0 else {"shift": shift}
                s.solve(b, **kw)
            k = "symmlq_s%g_" % shift
            out.update({k + "x": s.x, k + "nMatvec": s.nMatvec, k + "residNorm": s.residNorm,
                        k + "xNorm": s.xNorm, k + "anorm": s.anorm, k + "acond": s.acond,
                        k + "trace": np.array(log)})
        A = random_diagdom(2000, seed=3)
        n = A.shape[0]
        d = 1.0 / A.diagonal()
        rhs = A @ np.ones(n)
        out.update(csr_arrays(A, "ns_A_"))
        out["ns_d"] = d
        out["ns_rhs"] = rhs
        for sname, mod, cls in (("bicgstab", m_bicgstab, "BiCGSTAB"), ("cgs", m_cgs, "CGS"),
                                ("tfqmr", m_tfqmr, "TFQMR")):
            for gtag in ("zero", "guess"):
                kw = dict(matvec_max=2 * n)
                if gtag == "guess":
                    kw["guess"] = 1.0 + np.arange(n)
                with traced(mod) as log:
                    s = getattr(mod, cls)(csr_op(A), reltol=1e-8, precon=DiagonalOperator(d))
                    s.solve(rhs, **kw)
                k = "%s_%s_" % (sname, gtag)
                out.update({k + "x": s.x, k + "nMatvec": s.nMatvec, k + "residNorm": s.residNorm,
                            k + "residNorm0": s.residNorm0, k + "converged": s.converged,
                            k + "trace": np.array(log)})
        save("precon_jacobi.npz", **out)

    # ---------------- G11: composed operators (linop.py:307-330, :375-426) -- #
    def g11():
        from pykrylov.linop import DiagonalOperator, IdentityOperator
        out = {}
        m = 30
        n = m * m
        A = poisson2d(m)
        dv = np.linspace(0.5, 6.0, n)
        out.update(csr_arrays(A, "A_"))
        out["dv"] = dv
        x = np.random.default_rng(5).standard_normal(n)
        out["x"] = x
        I = IdentityOperator(n)
        D = DiagonalOperator(dv)
        forms = {"Am15I": lambda o: o - 1.5 * I, "ApD": lambda o: o + D, "DmA": lambda o: D - o,
                 "2p5A": lambda o: 2.5 * o, "negA": lambda o: -o, "Adiv3": lambda o: o / 3.0,
                 "nested": lambda o: 2.0 * (o - 1.5 * I) + 0.25 * D}
        for k, f in forms.items():
            out["y_" + k] = f(csr_op(A, True)) * x
        # CG on A + D (SPD), MINRES on A - 1.5 I (indefinite) and on 0.5 * A
        op = csr_op(A, True) + D
        rhs = op * np.ones(n)
        with traced(m_cg) as log:
            s = m_cg.CG(op)
            s.solve(rhs)
        out.update({"cg_ApD_rhs": rhs, "cg_ApD_x": s.x, "cg_ApD_nMatvec": s.nMatvec,
                    "cg_ApD_residHistory": np.array(s.residHistory), "cg_ApD_trace": np.array(log)})
        for k, f in (("Am15I", forms["Am15I"]), ("halfA", lambda o: 0.5 * o)):
            op = f(csr_op(A, True))
            rhs = op * np.ones(n)
            with traced(m_minres) as log:
                s = m_minres.Minres(op)
                quiet(s.solve, rhs, show=False, check=False, etol=0.0, rtol=1e-10)
            kk = "minres_%s_" % k
            out.update({kk + "rhs": rhs, kk + "x": s.x, kk + "istop": s.istop, kk + "itn": s.itn,
                        kk + "residHistory": np.array(s.residHistory), kk + "Anorm": s.Anorm,
                        kk + "trace": np.array(log)})
        save("composed_ops.npz", **out)

    # ---------------- G12: lls solvers with diagonal M, N ------------------ #
    def g12():
        from pykrylov.linop import DiagonalOperator
        out = {}
        mm, nn = 60, 40
        rng = np.random.default_rng(7)
        A = canon(sp.random(mm, nn, density=0.2, random_state=rng, data_rvs=rng.standard_normal) + sp.eye(mm, nn))
        out.update(csr_arrays(A, "A_"))
        dm = np.linspace(0.5, 2.0, mm)
        dn = np.linspace(3.0, 0.25, nn)
        b_cons = A @ np.ones(nn)
        b_ls = b_cons + 0.1 * np.random.default_rng(8).standard_normal(mm)
        out.update(dm=dm, dn=dn, b_cons=b_cons, b_ls=b_ls)
        for sname, mod, cls in (("lsqr", m_lsqr, "LSQRFramework"), ("lsmr", m_lsmr, "LSMRFramework"),
                                ("craig", m_craig, "CRAIGFramework"), ("craigmr", m_craigmr, "CRAIGMRFramework")):
            for btag, b in (("cons", b_cons), ("ls", b_ls)):
                if sname.startswith("craig") and btag == "ls":
                    continue
                for ptag, kwp in (("MN", dict(M=DiagonalOperator(dm), N=DiagonalOperator(dn))),
                                  ("M", dict(M=DiagonalOperator(dm))), ("N", dict(N=DiagonalOperator(dn)))):
                    with traced(mod) as log:
                        s = getattr(mod, cls)(csr_op(A))
                        ret = quiet(s.solve, b.copy(), show=False, etol=0.0, **kwp)
                    k = "%s_%s_%s_" % (sname, btag, ptag)
                    rec = {k + "x": s.x, k + "trace": np.array(log)}
                    for attr in ("istop", "itn", "nMatvec", "r1norm", "r2norm", "Anorm", "Acond", "Arnorm", "xnorm"):
                        v = getattr(s, attr, None)
                        if v is not None and np.isscalar(v):
                            rec[k + attr] = v
                    if sname == "lsmr":
                        names = ("istop", "itn", "normr", "normar", "normA", "condA", "normx")
                        rec.update({k + nme: val for nme, val in zip(names, ret[1:])})
                    out.update(rec)
        save("lls_precon.npz", **out)

    g10()
    g11()
    g12()
    if "--only-precon" in sys.argv:
        shutil.rmtree(tmp, ignore_errors=True)
        return

    # ---------------- G1: CG on 1138bus --------------------------------- #
    A = mm_csr("1138bus.mtx")
    n = A.shape[0]
    rhs = A @ np.ones(n)
    with traced(m_cg) as log:
        s = m_cg.CG(csr_op(A, True))
        s.solve(rhs)
    save("cg_1138bus.npz", rhs=rhs, x=s.x, nMatvec=s.nMatvec, residNorm0=s.residNorm0,
         residNorm=s.residNorm, residHistory=np.array(s.residHistory), converged=s.converged,
         trace=np.array(log), **csr_arrays(A, "A_"))

    # ---------------- G2: CG on 2-D Poisson ------------------------------ #
    out = {}
    for m in (10, 20, 100):
        A = poisson2d(m)
        n = m * m
        out.update(csr_arrays(A, "m%d_A_" % m))
        for tag, rhs in (("ones", A @ np.ones(n)),
                         ("randn", np.random.default_rng(0).standard_normal(n))):
            with traced(m_cg) as log:
                s = m_cg.CG(csr_op(A, True))
                s.solve(rhs)
            k = "m%d_%s_" % (m, tag)
            out.update({k + "rhs": rhs, k + "x": s.x, k + "nMatvec": s.nMatvec,
                        k + "residHistory": np.array(s.residHistory), k + "residNorm": s.residNorm,
                        k + "trace": np.array(log)})
        # warm start + matrix-free gallery operator (test_diagdom.py:75-79 protocol)
        g = 1.0 + np.arange(n)
        s = m_cg.CG(csr_op(A, True))
        s.solve(A @ np.ones(n), guess=g.copy(), matvec_max=50)
        k = "m%d_guess_" % m
        out.update({k + "x": s.x, k + "nMatvec": s.nMatvec, k + "residHistory": np.array(s.residHistory)})
        s = m_cg.CG(LinearOperator(n, n, lambda v: Poisson2dMatvec(v), symmetric=True))
        s.solve(A @ np.ones(n))
        k = "m%d_gallery_" % m
        out.update({k + "x": s.x, k + "nMatvec": s.nMatvec, k + "residHistory": np.array(s.residHistory)})
    save("cg_poisson2d.npz", **out)

    # ---------------- G3: CG on 1-D Poisson ------------------------------ #
    out = {}
    for n in (10, 100, 1000):
        A = poisson1d(n)
        rhs = A @ np.ones(n)
        s = m_cg.CG(csr_op(A, True))
        s.solve(rhs)
        s2 = m_cg.CG(LinearOperator(n, n, lambda v: Poisson1dMatvec(v), symmetric=True))
        s2.solve(rhs)
        k = "n%d_" % n
        out.update(csr_arrays(A, k + "A_"))
        out.update({k + "x": s.x, k + "nMatvec": s.nMatvec, k + "residNorm": s.residNorm,
                    k + "residHistory": np.array(s.residHistory),
                    k + "gallery_nMatvec": s2.nMatvec, k + "gallery_residNorm": s2.residNorm})
    save("cg_poisson1d.npz", **out)

    # ---------------- G4: nonsymmetric solvers --------------------------- #
    solvers = (("bicgstab", m_bicgstab, "BiCGSTAB"), ("cgs", m_cgs, "CGS"), ("tfqmr", m_tfqmr, "TFQMR"))
    for mname, A in (("jpwh991", mm_csr("jpwh_991.mtx")), ("rand10k", random_diagdom(10000, seed=1))):
        n = A.shape[0]
        rhs = A @ np.ones(n)
        out = dict(rhs=rhs, **csr_arrays(A, "A_"))
        for sname, mod, cls in solvers:
            for tol in (1e-5, 1e-8):
                for gtag in ("guess", "zero"):
                    kw = dict(matvec_max=2 * n)
                    if gtag == "guess":
                        kw["guess"] = 1.0 + np.arange(n)      # examples/bmark.py:51
                    with traced(mod) as log:
                        s = getattr(mod, cls)(csr_op(A), reltol=tol)
                        s.solve(rhs, **kw)
                    k = "%s_%g_%s_" % (sname, tol, gtag)
                    out.update({k + "x": s.x, k + "nMatvec": s.nMatvec, k + "residNorm": s.residNorm,
                                k + "residNorm0": s.residNorm0, k + "converged": s.converged,
                                k + "trace": np.array(log)})
        save("nonsym_%s.npz" % mname, **out)

    # ---------------- G5: MINRES ---------------------------------------- #
    out = {}
    for m in (30, 100):
        A = poisson2d(m)
        n = m * m
        out.update(csr_arrays(A, "m%d_A_" % m))
        for shift in (0.0, 1.5):
            for check in (False, True):
                rhs = A @ np.ones(n) - shift * np.ones(n)
                with traced(m_minres) as log:
                    s = m_minres.Minres(csr_op(A, True))
                    quiet(s.solve, rhs, shift=shift, show=False, check=check, etol=0.0, rtol=1e-10)
                k = "m%d_s%g_c%d_" % (m, shift, check)
                out.update({k + "rhs": rhs, k + "x": s.x, k + "istop": s.istop, k + "itn": s.itn,
                            k + "residHistory": np.array(s.residHistory), k + "rnorm": s.rnorm,
                            k + "Anorm": s.Anorm, k + "Acond": s.Acond, k + "Arnorm": s.Arnorm,
                            k + "ynorm": s.ynorm, k + "residNorm0": s.residNorm0,
                            k + "trace": np.array(log)})
        # default etol (stops on the direct-error test, istop = 10; minres.py:309-310)
        rhs = A @ np.ones(n)
        s = m_minres.Minres(csr_op(A, True))
        quiet(s.solve, rhs, show=False, check=False)
        k = "m%d_etoldef_" % m
        out.update({k + "x": s.x, k + "istop": s.istop, k + "itn": s.itn, k + "rnorm": s.rnorm,
                    k + "residHistory": np.array(s.residHistory),
                    k + "dir_errors_window": np.array(s.dir_errors_window)})
    save("minres_poisson2d.npz", **out)

    # ---------------- G6: SYMMLQ ---------------------------------------- #
    out = {}
    for m in (30, 100):
        A = poisson2d(m)
        n = m * m
        for shift in (None, 1.5):
            rhs = A @ np.ones(n) - (shift or 0.0) * np.ones(n)
            op = csr_op(A, True)
            with traced(m_symmlq) as log:
                s = m_symmlq.Symmlq(op)
                s.matvec = lambda v, op=op: op * v        # symmlq.py:162 calls a missing attribute
                kw = {} if shift is None else {"shift": shift}
                s.solve(rhs, **kw)
            k = "m%d_s%g_" % (m, shift or 0.0)
            out.update({k + "rhs": rhs, k + "x": s.x, k + "nMatvec": s.nMatvec, k + "residNorm": s.residNorm,
                        k + "xNorm": s.xNorm, k + "anorm": s.anorm, k + "acond": s.acond,
                        k + "trace": np.array(log)})
    save("symmlq_poisson2d.npz", **out)

    # ---------------- G7: least-squares family --------------------------- #
    out = {}
    for tag, (mm, nn, dens) in (("s", (60, 40, 0.2)), ("l", (2000, 1500, 0.004))):
        rng = np.random.default_rng(7)
        A = canon(sp.random(mm, nn, density=dens, random_state=rng, data_rvs=rng.standard_normal)
                  + sp.eye(mm, nn))
        xs = np.ones(nn)
        b_cons = A @ xs                                        # consistent system
        b_ls = b_cons + 0.1 * np.random.default_rng(8).standard_normal(mm)
        out.update(csr_arrays(A, tag + "_A_"))
        out[tag + "_b_cons"] = b_cons
        out[tag + "_b_ls"] = b_ls
        for sname, mod, cls in (("lsqr", m_lsqr, "LSQRFramework"), ("lsmr", m_lsmr, "LSMRFramework"),
                                ("craig", m_craig, "CRAIGFramework"), ("craigmr", m_craigmr, "CRAIGMRFramework")):
            for btag, b in (("cons", b_cons), ("ls", b_ls)):
                if sname.startswith("craig") and btag == "ls":
                    continue                                    # CRAIG needs a consistent system
                for damp, etol in ((0.0, 1e-6), (0.1, 1e-6), (0.0, 0.0)):
                    if sname.startswith("craig") and damp != 0.0:
                        continue
                    with traced(mod) as log:
                        s = getattr(mod, cls)(csr_op(A))
                        kw = dict(damp=damp, show=False, etol=etol)
                        ret = quiet(s.solve, b.copy(), **kw)
                    k = "%s_%s_%s_d%g_e%g_" % (tag, sname, btag, damp, etol)
                    rec = {k + "x": s.x, k + "trace": np.array(log)}
                    for attr in ("istop", "itn", "nMatvec", "r1norm", "r2norm", "Anorm", "Acond", "Arnorm",
                                 "xnorm"):
                        v = getattr(s, attr, None)
                        if v is not None and np.isscalar(v):
                            rec[k + attr] = v
                    if sname == "lsmr":                     # lsmr.py:492 returns a tuple instead
                        names = ("istop", "itn", "normr", "normar", "normA", "condA", "normx")
                        rec.update({k + nme: val for nme, val in zip(names, ret[1:])})
                    out.update(rec)
    save("lls_random.npz", **out)

    # ---------------- G8: large-n summaries + integer checksums ---------- #
    out = {}
    A = poisson2d(1000)
    n = A.shape[0]
    out.update({"p2d1000_nnz": A.nnz, "p2d1000_indptr_sha": sha(A.indptr), "p2d1000_indices_sha": sha(A.indices),
                "p2d1000_indices_sum": np.int64(A.indices.astype(np.int64).sum())})
    rhs = A @ np.ones(n)
    s = m_cg.CG(csr_op(A, True))
    s.solve(rhs)
    out.update({"p2d1000_cg_nMatvec": s.nMatvec, "p2d1000_cg_residHistory": np.array(s.residHistory),
                "p2d1000_cg_x_sample": s.x[::997].copy(), "p2d1000_cg_residNorm": s.residNorm})
    # the same reference code with np.dot replaced by an exactly rounded inner product (80-bit
    # accumulation): separates the reference's own summation noise from the algorithm
    with traced(m_cg, exact=True):
        s = m_cg.CG(csr_op(A, True))
        s.solve(rhs)
    out.update({"p2d1000_cg_residHistory_exactdot": np.array(s.residHistory),
                "p2d1000_cg_nMatvec_exactdot": s.nMatvec})
    A = random_diagdom(1000000, seed=1)
    n = A.shape[0]
    out.update({"rand1m_nnz": A.nnz, "rand1m_indptr_sha": sha(A.indptr), "rand1m_indices_sha": sha(A.indices),
                "rand1m_data_sha": sha(A.data), "rand1m_indices_sum": np.int64(A.indices.astype(np.int64).sum())})
    rhs = A @ np.ones(n)
    with traced(m_bicgstab) as log:
        s = m_bicgstab.BiCGSTAB(csr_op(A), reltol=1e-10)
        s.solve(rhs)
    out.update({"rand1m_bicgstab_nMatvec": s.nMatvec, "rand1m_bicgstab_residNorm": s.residNorm,
                "rand1m_bicgstab_residNorm0": s.residNorm0, "rand1m_bicgstab_trace": np.array(log),
                "rand1m_bicgstab_x_sample": s.x[::997].copy()})
    A = poisson2d(2000)
    n = A.shape[0]
    out.update({"p2d2000_nnz": A.nnz, "p2d2000_indptr_sha": sha(A.indptr), "p2d2000_indices_sha": sha(A.indices)})
    rhs = A @ np.ones(n) - 1.5 * np.ones(n)
    s = m_minres.Minres(csr_op(A, True))
    quiet(s.solve, rhs, shift=1.5, show=False, check=False, etol=0.0, rtol=1e-8, itnlim=500)
    out.update({"p2d2000_minres_residHistory": np.array(s.residHistory), "p2d2000_minres_istop": s.istop,
                "p2d2000_minres_itn": s.itn, "p2d2000_minres_x_sample": s.x[::3989].copy()})
    for m in (8, 16):
        A = poisson3d(m)
        out.update(csr_arrays(A, "p3d%d_A_" % m))
        s = m_cg.CG(csr_op(A, True))
        s.solve(A @ np.ones(m ** 3))
        out.update({"p3d%d_cg_residHistory" % m: np.array(s.residHistory), "p3d%d_cg_x" % m: s.x})
    save("large_summaries.npz", **out)

    # ---------------- G9: operator protocol facts ------------------------ #
    out = {}
    B = np.array([[1.0, 2.0, 3.0], [4.0, 5.0, 6.0]])
    op = LinearOperator(3, 2, matvec=lambda v: B @ v, matvec_transp=lambda u: B.T @ u)
    v = np.array([1.0, 1.0, 1.0])
    out["mul_result"] = op * v
    out["T_result"] = op.T * np.array([1.0, 1.0])
    out["nMatvec_after_2"] = (op * v, op * v, op.nMatvec)[2]
    out["shape"] = np.array(op.shape)
    out["T_shape"] = np.array(op.T.shape)
    sym = LinearOperator(2, 2, matvec=lambda v: v, symmetric=True)
    out["sym_T_is_self"] = sym.T is sym
    names = []
    for dt in (np.int32, np.int64, np.float32, np.float64, np.complex64, np.complex128):
        r = op * v.astype(dt)
        names.append(np.dtype(r.dtype).name)
    out["promotion_vs_float64_op"] = np.array(names)
    try:
        op * np.ones(5)
        out["size_mismatch_exc"] = "none"
    except Exception as e:                                  # linop.py:283-285
        out["size_mismatch_exc"] = type(e).__name__
    try:
        op * "abc"
        out["bad_operand_exc"] = "none"
    except Exception as e:                                  # linop.py:369
        out["bad_operand_exc"] = type(e).__name__
    save("api_contract.npz", **out)

    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
