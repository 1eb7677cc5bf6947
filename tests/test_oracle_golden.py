"""Pins the CPU oracle (oracle/) against fixtures produced by the real reference.

Everything here is CPU-only.  Bit-exact equality is demanded wherever the oracle
and the reference perform the same floating-point operations in the same order
(iteration counts, reduction traces, residual histories, iterates).
"""
import os

import numpy as np
import pytest

from oracle import csr_ref, krylov_ref as kr

REF_EXAMPLES = "/root/reference/examples"


def csr_from(d, prefix):
    return csr_ref.RefCsr(d[prefix + "indptr"], d[prefix + "indices"], d[prefix + "data"], d[prefix + "shape"])


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


# ------------------------------------------------------------------ matrices
@pytest.mark.parametrize("m", [10, 20, 100])
def test_poisson2d_builder_bit_exact(golden, m):
    d = golden("cg_poisson2d.npz")
    A = csr_ref.poisson2d(m)
    assert same(A.indptr, d["m%d_A_indptr" % m]) and A.indptr.dtype == np.int32
    assert same(A.indices, d["m%d_A_indices" % m])
    assert same(A.data, d["m%d_A_data" % m])


@pytest.mark.parametrize("n", [10, 100, 1000])
def test_poisson1d_builder_bit_exact(golden, n):
    d = golden("cg_poisson1d.npz")
    A = csr_ref.poisson1d(n)
    for k in ("indptr", "indices", "data"):
        assert same(getattr(A, k), d["n%d_A_%s" % (n, k)])


@pytest.mark.parametrize("m", [8, 16])
def test_poisson3d_builder_bit_exact(golden, m):
    d = golden("large_summaries.npz")
    A = csr_ref.poisson3d(m)
    for k in ("indptr", "indices", "data"):
        assert same(getattr(A, k), d["p3d%d_A_%s" % (m, k)])


def test_random_builder_bit_exact(golden):
    d = golden("nonsym_rand10k.npz")
    A = csr_ref.random_diagdom(10000, seed=1)
    for k in ("indptr", "indices", "data"):
        assert same(getattr(A, k), d["A_" + k])


@pytest.mark.skipif(not os.path.isdir(REF_EXAMPLES), reason="reference examples not present on this box")
@pytest.mark.parametrize("name,fix", [("1138bus.mtx", "cg_1138bus.npz"), ("jpwh_991.mtx", "nonsym_jpwh991.npz")])
def test_matrix_market_reader_bit_exact(golden, name, fix):
    d = golden(fix)
    A = csr_ref.read_matrix_market(os.path.join(REF_EXAMPLES, name))
    for k in ("indptr", "indices", "data"):
        assert same(getattr(A, k), d["A_" + k])


# ------------------------------------------------------------------ SpMV
@pytest.mark.parametrize("force_numpy", [False, True])
def test_spmv_matches_scipy_products(golden, force_numpy):
    # rhs = A @ ones was produced by SciPy's csr_matvec in make_golden.py
    for fix in ("cg_1138bus.npz", "nonsym_jpwh991.npz", "nonsym_rand10k.npz"):
        d = golden(fix)
        A = csr_from(d, "A_")
        assert same(A.matvec(np.ones(A.shape[1]), force_numpy=force_numpy), d["rhs"])
    d = golden("cg_poisson2d.npz")
    A = csr_from(d, "m100_A_")
    x = d["m100_randn_x"]
    try:
        import scipy.sparse as sp
    except ImportError:
        return
    S = sp.csr_matrix((A.data, A.indices, A.indptr), shape=A.shape)
    assert same(A.matvec(x, force_numpy=force_numpy), S @ x)
    assert same(A.rmatvec(x), S.T.tocsr() @ x)


# ------------------------------------------------------------------ CG
def test_cg_1138bus(golden):
    d = golden("cg_1138bus.npz")
    A = csr_from(d, "A_")
    out = kr.cg(A, d["rhs"])
    assert out["nMatvec"] == int(d["nMatvec"]) == 1751
    assert same(out["residHistory"], d["residHistory"])
    assert same(out["x"], d["x"])
    assert same(out["trace"], d["trace"])
    assert out["residNorm0"] == float(d["residNorm0"])
    assert bool(out["converged"]) == bool(d["converged"])


@pytest.mark.parametrize("m", [10, 20, 100])
@pytest.mark.parametrize("tag", ["ones", "randn"])
def test_cg_poisson2d(golden, m, tag):
    d = golden("cg_poisson2d.npz")
    A = csr_from(d, "m%d_A_" % m)
    k = "m%d_%s_" % (m, tag)
    out = kr.cg(A, d[k + "rhs"])
    assert out["nMatvec"] == int(d[k + "nMatvec"])
    assert same(out["residHistory"], d[k + "residHistory"])
    assert same(out["x"], d[k + "x"])


@pytest.mark.parametrize("m", [10, 20, 100])
def test_cg_warm_start_and_matvec_max(golden, m):
    d = golden("cg_poisson2d.npz")
    A = csr_from(d, "m%d_A_" % m)
    n = m * m
    out = kr.cg(A, A.matvec(np.ones(n)), guess=1.0 + np.arange(n), matvec_max=50)
    k = "m%d_guess_" % m
    assert out["nMatvec"] == int(d[k + "nMatvec"])
    assert same(out["residHistory"], d[k + "residHistory"])
    assert same(out["x"], d[k + "x"])


def test_cg_poisson1d_doc_numbers(golden):
    d = golden("cg_poisson1d.npz")
    for n in (10, 100, 1000):
        A = csr_from(d, "n%d_A_" % n)
        out = kr.cg(A, A.matvec(np.ones(n)))
        assert out["nMatvec"] == int(d["n%d_nMatvec" % n])
        assert same(out["residHistory"], d["n%d_residHistory" % n])
    # doc/source/introduction.rst:46-48: 50 matvecs, residual 7.39e-14
    # (with the matrix-free gallery operator the docs used; the CSR product rounds differently)
    assert int(d["n100_nMatvec"]) == 50 == int(d["n100_gallery_nMatvec"])
    assert abs(float(d["n100_gallery_residNorm"]) - 7.39e-14) < 1e-16


# ------------------------------------------------------------------ nonsymmetric solvers
@pytest.mark.parametrize("fix", ["nonsym_jpwh991.npz", "nonsym_rand10k.npz"])
@pytest.mark.parametrize("solver", ["bicgstab", "cgs", "tfqmr"])
@pytest.mark.parametrize("tol", [1e-5, 1e-8])
@pytest.mark.parametrize("gtag", ["guess", "zero"])
def test_nonsymmetric(golden, fix, solver, tol, gtag):
    d = golden(fix)
    A = csr_from(d, "A_")
    n = A.shape[0]
    kw = dict(reltol=tol, matvec_max=2 * n)
    if gtag == "guess":
        kw["guess"] = 1.0 + np.arange(n)
    with np.errstate(all="ignore"):
        out = getattr(kr, solver)(A, d["rhs"], **kw)
    k = "%s_%g_%s_" % (solver, tol, gtag)
    assert out["nMatvec"] == int(d[k + "nMatvec"])
    assert same(out["trace"], d[k + "trace"])
    assert same(out["x"], d[k + "x"])
    assert same(out["residNorm"], d[k + "residNorm"])
    assert same(out["residNorm0"], d[k + "residNorm0"])
    assert bool(out["converged"]) == bool(d[k + "converged"])


def test_doc_numbers_jpwh991(golden):
    # doc/source/bmark.rst:52-54 (Pysparse matvec): 82 / 84 / 84 matvecs at reltol 1e-8
    d = golden("nonsym_jpwh991.npz")
    got = [int(d["%s_1e-08_guess_nMatvec" % s]) for s in ("cgs", "tfqmr", "bicgstab")]
    assert all(abs(g - w) <= 2 for g, w in zip(got, (82, 84, 84))), got


# ------------------------------------------------------------------ MINRES
@pytest.mark.parametrize("m", [30, 100])
@pytest.mark.parametrize("shift", [0.0, 1.5])
@pytest.mark.parametrize("check", [False, True])
def test_minres(golden, m, shift, check):
    d = golden("minres_poisson2d.npz")
    A = csr_from(d, "m%d_A_" % m)
    k = "m%d_s%g_c%d_" % (m, shift, check)
    out = kr.minres(A, d[k + "rhs"], shift=shift, check=check, etol=0.0, rtol=1e-10)
    assert (out["istop"], out["itn"]) == (int(d[k + "istop"]), int(d[k + "itn"]))
    assert same(out["residHistory"], d[k + "residHistory"])
    assert same(out["trace"], d[k + "trace"])
    assert same(out["x"], d[k + "x"])
    for name in ("rnorm", "Anorm", "Acond", "Arnorm", "ynorm", "residNorm0"):
        assert out[name] == float(d[k + name]), name


@pytest.mark.parametrize("m", [30, 100])
def test_minres_default_etol_stops_on_direct_error(golden, m):
    d = golden("minres_poisson2d.npz")
    A = csr_from(d, "m%d_A_" % m)
    k = "m%d_etoldef_" % m
    out = kr.minres(A, A.matvec(np.ones(m * m)), check=False)
    assert (out["istop"], out["itn"]) == (int(d[k + "istop"]), int(d[k + "itn"])) and out["istop"] == 10
    assert same(out["residHistory"], d[k + "residHistory"])
    assert same(out["dir_errors_window"], d[k + "dir_errors_window"])
    assert same(out["x"], d[k + "x"])


# ------------------------------------------------------------------ SYMMLQ
@pytest.mark.parametrize("m", [30, 100])
@pytest.mark.parametrize("shift", [0.0, 1.5])
def test_symmlq(golden, m, shift):
    d = golden("symmlq_poisson2d.npz")
    A = csr_from(golden("minres_poisson2d.npz"), "m%d_A_" % m)
    k = "m%d_s%g_" % (m, shift)
    out = kr.symmlq(A, d[k + "rhs"], shift=(shift or None))
    assert out["nMatvec"] == int(d[k + "nMatvec"])
    assert same(out["trace"], d[k + "trace"])
    assert same(out["x"], d[k + "x"])
    for name in ("residNorm", "xNorm", "anorm", "acond"):
        assert out[name] == float(d[k + name]), name


# ------------------------------------------------------------------ diagonal preconditioning
@pytest.mark.parametrize("gtag", ["zero", "guess"])
def test_precon_cg(golden, gtag):
    d = golden("precon_jacobi.npz")
    A = csr_from(d, "spd_A_")
    n = A.shape[0]
    dg = d["spd_d"]
    kw = {} if gtag == "zero" else {"guess": 1.0 + np.arange(n)}
    out = kr.cg(A, d["spd_rhs"], precon=lambda v: dg * v, **kw)
    k = "cg_%s_" % gtag
    # the reference builds p from r, not from y = precon*r (cg.py:104,150-151): with a preconditioner its CG
    # does not converge on this matrix -- reproduced, not repaired
    assert out["nMatvec"] == int(d[k + "nMatvec"]) == 2 * n and not bool(d[k + "converged"])
    assert same(out["residHistory"], d[k + "residHistory"])
    assert same(out["trace"], d[k + "trace"])
    assert same(out["x"], d[k + "x"])
    assert same(out["residNorm0"], d[k + "residNorm0"])


@pytest.mark.parametrize("solver", ["bicgstab", "cgs", "tfqmr"])
@pytest.mark.parametrize("gtag", ["zero", "guess"])
def test_precon_nonsymmetric(golden, solver, gtag):
    d = golden("precon_jacobi.npz")
    A = csr_from(d, "ns_A_")
    n = A.shape[0]
    dg = d["ns_d"]
    kw = dict(reltol=1e-8, matvec_max=2 * n, precon=lambda v: dg * v)
    if gtag == "guess":
        kw["guess"] = 1.0 + np.arange(n)
    out = getattr(kr, solver)(A, d["ns_rhs"], **kw)
    k = "%s_%s_" % (solver, gtag)
    assert out["nMatvec"] == int(d[k + "nMatvec"])
    assert same(out["trace"], d[k + "trace"])
    assert same(out["x"], d[k + "x"])
    assert same(out["residNorm"], d[k + "residNorm"])
    assert bool(out["converged"]) == bool(d[k + "converged"])


@pytest.mark.parametrize("shift", [0.0, 1.5])
def test_precon_minres_symmlq(golden, shift):
    d = golden("precon_jacobi.npz")
    A = csr_from(d, "spd_A_")
    dg = d["spd_d"]
    k = "minres_s%g_" % shift
    out = kr.minres(A, d[k + "rhs"], precon=lambda v: dg * v, shift=shift, check=False, etol=0.0, rtol=1e-10)
    assert (out["istop"], out["itn"]) == (int(d[k + "istop"]), int(d[k + "itn"]))
    assert same(out["residHistory"], d[k + "residHistory"])
    assert same(out["trace"], d[k + "trace"])
    assert same(out["x"], d[k + "x"])
    for name in ("rnorm", "Anorm", "Acond", "Arnorm", "ynorm", "residNorm0"):
        assert out[name] == float(d[k + name]), name
    k2 = "symmlq_s%g_" % shift
    out = kr.symmlq(A, d[k + "rhs"], precon=lambda v: dg * v, shift=(shift or None))
    assert out["nMatvec"] == int(d[k2 + "nMatvec"])
    assert same(out["trace"], d[k2 + "trace"])
    assert same(out["x"], d[k2 + "x"])
    for name in ("residNorm", "xNorm", "anorm", "acond"):
        assert out[name] == float(d[k2 + name]), name


# ------------------------------------------------------------------ composed operators
def test_composed_operator_closures_match_reference(golden):
    """oracle/csr_ref.Composed restates the reference's operator algebra; products and solver runs on composed
    operators reproduce the reference's bits."""
    d = golden("composed_ops.npz")
    A = csr_from(d, "A_")
    dv, x = d["dv"], d["x"]
    fns = {"Am15I": lambda y, x: y - 1.5 * x, "ApD": lambda y, x: y + dv * x, "DmA": lambda y, x: dv * x - y,
           "2p5A": lambda y, x: 2.5 * y, "negA": lambda y, x: -1 * y, "Adiv3": lambda y, x: (1. / 3.0) * y,
           "nested": lambda y, x: 2.0 * (y - 1.5 * x) + 0.25 * (dv * x)}
    for k, fn in fns.items():
        assert same(csr_ref.Composed(A, fn).matvec(x), d["y_" + k]), k
    out = kr.cg(csr_ref.Composed(A, fns["ApD"]), d["cg_ApD_rhs"])
    assert out["nMatvec"] == int(d["cg_ApD_nMatvec"])
    assert same(out["residHistory"], d["cg_ApD_residHistory"]) and same(out["trace"], d["cg_ApD_trace"])
    assert same(out["x"], d["cg_ApD_x"])
    for k, fn in (("Am15I", fns["Am15I"]), ("halfA", lambda y, x: 0.5 * y)):
        kk = "minres_%s_" % k
        out = kr.minres(csr_ref.Composed(A, fn), d[kk + "rhs"], check=False, etol=0.0, rtol=1e-10)
        assert (out["istop"], out["itn"]) == (int(d[kk + "istop"]), int(d[kk + "itn"]))
        assert same(out["residHistory"], d[kk + "residHistory"]) and same(out["trace"], d[kk + "trace"])
        assert same(out["x"], d[kk + "x"])


# ------------------------------------------------------------------ large-n summaries
def test_large_matrix_checksums(golden):
    import hashlib
    d = golden("large_summaries.npz")

    def sha(a):
        return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)

    A = csr_ref.poisson2d(1000)
    assert A.nnz == int(d["p2d1000_nnz"]) == 4996000
    assert same(sha(A.indptr), d["p2d1000_indptr_sha"]) and same(sha(A.indices), d["p2d1000_indices_sha"])


# ------------------------------------------------------------------ least-squares family
def _lls_cases():
    for tag in ("s", "l"):
        for solver in ("lsqr", "lsmr", "craig", "craigmr"):
            for btag in ("cons", "ls"):
                if solver.startswith("craig") and btag == "ls":
                    continue
                for damp, etol in ((0.0, 1e-6), (0.1, 1e-6), (0.0, 0.0)):
                    if solver.startswith("craig") and damp != 0.0:
                        continue
                    yield tag, solver, btag, damp, etol


@pytest.mark.parametrize("tag,solver,btag,damp,etol", list(_lls_cases()))
def test_lls(golden, tag, solver, btag, damp, etol):
    from oracle import lls_ref
    d = golden("lls_random.npz")
    A = csr_from(d, tag + "_A_")
    At = A.transpose()
    b = d[tag + "_b_" + btag]
    k = "%s_%s_%s_d%g_e%g_" % (tag, solver, btag, damp, etol)
    if solver == "lsqr":
        out = lls_ref.lsqr(A.matvec, At.matvec, A.shape, b.copy(), damp=damp, etol=etol)
        names = ("istop", "itn", "nMatvec", "r1norm", "r2norm", "Anorm", "Acond", "Arnorm", "xnorm")
    elif solver == "lsmr":
        out = lls_ref.lsmr(A.matvec, At.matvec, A.shape, b.copy(), damp=damp, etol=etol)
        names = ("istop", "itn", "normr", "normar", "normA", "condA", "normx")
    elif solver == "craig":
        out = lls_ref.craig(A.matvec, At.matvec, A.shape, b.copy(), etol=etol)
        names = ("istop", "itn", "nMatvec", "r1norm", "r2norm", "Arnorm", "xnorm")
    else:
        out = lls_ref.craigmr(A.matvec, At.matvec, A.shape, b.copy(), etol=etol)
        names = ("istop", "itn", "nMatvec")
    for name in names:
        assert out[name] == d[k + name].item(), name
    assert same(out["trace"], d[k + "trace"])
    assert same(out["x"], d[k + "x"])


def _lls_precon_cases():
    for solver in ("lsqr", "lsmr", "craig", "craigmr"):
        for btag in ("cons", "ls"):
            if solver.startswith("craig") and btag == "ls":
                continue
            for ptag in ("MN", "M", "N"):
                yield solver, btag, ptag


@pytest.mark.parametrize("solver,btag,ptag", list(_lls_precon_cases()))
def test_lls_diagonal_preconditioners(golden, solver, btag, ptag):
    """M (m-space) and N (n-space) as DiagonalOperators in the reference run (lsqr.py:189-190,201-202, ...)."""
    from oracle import lls_ref
    d = golden("lls_precon.npz")
    A = csr_from(d, "A_")
    At = A.transpose()
    dm, dn = d["dm"], d["dn"]
    kw = dict(etol=0.0)
    if "M" in ptag:
        kw["M"] = lambda u: dm * u
    if "N" in ptag:
        kw["N"] = lambda v: dn * v
    b = d["b_" + btag]
    k = "%s_%s_%s_" % (solver, btag, ptag)
    out = getattr(lls_ref, solver)(A.matvec, At.matvec, A.shape, b.copy(), **kw)
    names = {"lsqr": ("istop", "itn", "nMatvec", "r1norm", "r2norm", "Anorm", "Acond", "Arnorm", "xnorm"),
             "lsmr": ("istop", "itn", "normr", "normar", "normA", "condA", "normx"),
             "craig": ("istop", "itn", "nMatvec", "r1norm", "r2norm", "Arnorm", "xnorm"),
             "craigmr": ("istop", "itn", "nMatvec")}[solver]
    for name in names:
        assert out[name] == d[k + name].item(), name
    assert same(out["trace"], d[k + "trace"])
    assert same(out["x"], d[k + "x"])


# ------------------------------------------------------------------ round 3: reference-run fixtures for the new operator kinds
def test_round3_matrix_preconditioners_match_reference(golden):
    """`precon * r` with a MATRIX (the inverted 4 x 4 diagonal blocks) in all six solvers: the oracle reproduces the
    reference's counts, reduction traces and iterates bit for bit (tests/golden/make_golden_r3.py)."""
    d = golden("round3_ops.npz")
    V, MV = csr_from(d, "pc_A_"), csr_from(d, "pc_M_")
    rhs = d["pc_rhs"]
    out = kr.cg(V, rhs, precon=MV.matvec, matvec_max=60)
    assert out["nMatvec"] == int(d["pc_cg_nMatvec"]) and same(out["residHistory"], d["pc_cg_hist"])
    assert same(out["trace"], d["pc_cg_trace"]) and same(out["x"], d["pc_cg_x"])
    out = kr.minres(V, rhs, precon=MV.matvec, check=False, etol=0.0, rtol=1e-10)
    assert (out["istop"], out["itn"]) == (int(d["pc_minres_istop"]), int(d["pc_minres_itn"]))
    assert same(out["residHistory"], d["pc_minres_hist"]) and same(out["trace"], d["pc_minres_trace"])
    assert same(out["x"], d["pc_minres_x"])
    out = kr.symmlq(V, rhs, precon=MV.matvec)
    assert out["nMatvec"] == int(d["pc_symmlq_nMatvec"]) and same(out["trace"], d["pc_symmlq_trace"])
    assert same(out["x"], d["pc_symmlq_x"]) and out["residNorm"] == float(d["pc_symmlq_residNorm"])
    W, MW = csr_from(d, "pn_A_"), csr_from(d, "pn_M_")
    for solver in ("bicgstab", "cgs", "tfqmr"):
        out = getattr(kr, solver)(W, d["pn_rhs"], reltol=1e-10, matvec_max=400, precon=MW.matvec)
        k = "pn_%s_" % solver
        assert out["nMatvec"] == int(d[k + "nMatvec"]) and same(out["trace"], d[k + "trace"]), solver
        assert same(out["x"], d[k + "x"]) and same(out["residNorm"], d[k + "residNorm"]), solver
        assert bool(out["converged"]) == bool(d[k + "converged"])


def test_round3_variable_coefficient_cg_matches_reference(golden):
    d = golden("round3_ops.npz")
    C = csr_from(d, "vc_A_")
    mine = csr_ref.poisson3d_varcoef(24, 16, 8, seed=7)          # (the fixture's matrix IS this builder's output)
    assert same(mine.indptr, C.indptr) and same(mine.indices, C.indices) and same(mine.data, C.data)
    out = kr.cg(C, d["vc_rhs"])
    assert out["nMatvec"] == int(d["vc_cg_nMatvec"]) and bool(d["vc_cg_converged"])
    assert same(out["residHistory"], d["vc_cg_hist"]) and same(out["trace"], d["vc_cg_trace"]) and same(out["x"], d["vc_cg_x"])


def test_round3_reduced_and_block_operators_host_composition(golden):
    """The package's HOST compositions (what operators without a device form use) against the reference's results:
    ReducedLinearOperator / SymmetricallyReducedLinearOperator (linop.py:560-623) and a saddle-point block operator
    of sparse blocks (blkop.py:8-152), bit for bit."""
    from pykrylov_amd import DiagonalOperator, LinearOperator, ReducedLinearOperator, SymmetricallyReducedLinearOperator
    from pykrylov_amd.blkop import BlockLinearOperator
    d = golden("round3_ops.npz")

    def host_op(R, symmetric=False):
        Rt = R.transpose()
        return LinearOperator(R.shape[1], R.shape[0], matvec=R.matvec, matvec_transp=Rt.matvec, symmetric=symmetric)
    R = csr_from(d, "red_A_")
    red = ReducedLinearOperator(host_op(R), d["red_rows"], d["red_cols"])
    assert red.shape == tuple(d["red_shape"])
    assert same(red * d["red_x"], d["red_y"]) and same(red.T * d["red_u"], d["red_yt"])
    P = csr_from(d, "sred_A_")
    sred = SymmetricallyReducedLinearOperator(host_op(P, True), d["sred_idx"])
    assert same(sred * d["sred_x"], d["sred_y"]) and bool(sred.symmetric) == bool(d["sred_sym"][0])
    A, B = csr_from(d, "sp_A_"), csr_from(d, "sp_B_")
    K = BlockLinearOperator([[host_op(A, True), host_op(B).T], [DiagonalOperator(d["sp_d"])]], symmetric=True)
    assert same(K * d["sp_x"], d["sp_Kx"]) and same(K * np.ones(K.shape[1]), d["sp_rhs"])
    call = lambda self, v: K * v                                     # noqa: E731  (the oracle calls `A(v)`)
    out = kr.minres(type("Op", (), {"shape": K.shape, "matvec": call, "__call__": call})(), d["sp_rhs"], check=False,
                    etol=0.0, rtol=1e-10)
    assert (out["istop"], out["itn"]) == (int(d["sp_minres_istop"]), int(d["sp_minres_itn"]))
    assert same(out["residHistory"], d["sp_minres_hist"]) and same(out["x"], d["sp_minres_x"])


def test_stencil27_twin_against_a_brute_force_construction():
    """The 27-point test operator of the wide storage formats (no reference counterpart beyond "a user's matvec"):
    the vectorised NumPy twin against three nested loops, constant and variable coefficients."""
    from oracle import csr_ref
    for (mx, my, mz), seed in (((4, 3, 5), 0), ((3, 4, 2), 9), ((1, 1, 6), 2), ((5, 1, 1), 0)):
        A = csr_ref.stencil27(mx, my, mz, seed=seed)
        n = mx * my * mz
        k = csr_ref.cell_field(np.arange(n), seed) if seed else np.ones(n)
        D = np.zeros((n, n))
        for z in range(mz):
            for y in range(my):
                for x in range(mx):
                    r = (z * my + y) * mx + x
                    diag = 0.0
                    for dz in (-1, 0, 1):
                        for dy in (-1, 0, 1):
                            for dx in (-1, 0, 1):
                                if (dz, dy, dx) == (0, 0, 0):
                                    continue
                                inside = 0 <= z + dz < mz and 0 <= y + dy < my and 0 <= x + dx < mx
                                c = r + (dz * my + dy) * mx + dx
                                h = k[r]
                                if inside and seed:
                                    h = ((2.0 * k[r]) * k[c]) / (k[r] + k[c])
                                diag = diag + h
                                if inside:
                                    D[r, c] = -h
                    D[r, r] = diag
        assert np.array_equal(A.to_dense(), D)
        assert np.array_equal(D, D.T) and np.linalg.eigvalsh(D).min() > 0
        assert np.all(np.diff(A.indices[A.indptr[0]:A.indptr[1]]) > 0)


@pytest.mark.parametrize("dims,rows", [((8, 6, 5), None), ((8, 6, 5), (37, 141)), ((16, 16, 16), None),
                                       ((5, 7, 3), (0, 105)), ((12, 9, 4), (108, 324))])
def test_varcoef_row_blocks_and_c_generator_twin(dims, rows):
    """`poisson3d_varcoef(rows=(a, b))` is the row block of the whole matrix, and the C generator (csr_ref.c, what
    bench.py's cpu_baseline holds the 512^3 problem with) writes the same arrays bit for bit."""
    whole = csr_ref.poisson3d_varcoef(*dims, seed=7)
    a, b = rows if rows else (0, whole.shape[0])
    blk = csr_ref.poisson3d_varcoef(*dims, seed=7, rows=rows)
    lo, hi = whole.indptr[a], whole.indptr[b]
    assert blk.shape == (b - a, whole.shape[1])
    assert np.array_equal(blk.indptr, whole.indptr[a:b + 1] - lo)
    assert np.array_equal(blk.indices, whole.indices[lo:hi]) and np.array_equal(blk.data, whole.data[lo:hi])
    x = np.random.default_rng(3).standard_normal(whole.shape[1])
    assert np.array_equal(blk.matvec(x), whole.matvec(x)[a:b])
    c = csr_ref.poisson3d_varcoef_c(*dims, seed=7, rows=rows)
    assert c.shape == blk.shape
    assert np.array_equal(c.indptr, blk.indptr) and np.array_equal(c.indices, blk.indices)
    assert np.array_equal(c.data, blk.data)


@pytest.mark.parametrize("dims,rows", [((8, 6, 5), None), ((8, 6, 5), (37, 141)), ((16, 16, 16), None), ((12, 9, 4), (108, 324))])
def test_constant_coefficient_c_generator_twin(dims, rows):
    """The C generator of the literal BASELINE configs[4] matrix (csr_ref.c ref_poisson3d_const_fill: what bench.py's
    cpu_baseline holds the constant-coefficient 512^3 problem with) writes the arrays of `poisson3d` bit for bit."""
    whole = csr_ref.poisson3d(*dims)
    a, b = rows if rows else (0, whole.shape[0])
    lo, hi = whole.indptr[a], whole.indptr[b]
    c = csr_ref.poisson3d_c(*dims, rows=rows)
    assert c.shape == (b - a, whole.shape[1])
    assert np.array_equal(c.indptr, whole.indptr[a:b + 1] - lo)
    assert np.array_equal(c.indices, whole.indices[lo:hi]) and np.array_equal(c.data, whole.data[lo:hi])
    assert c.indptr.dtype == np.int32 and c.indices.dtype == np.int32 and c.data.dtype == np.float64
