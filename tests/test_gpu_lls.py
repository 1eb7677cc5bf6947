"""GPU parity of the device-resident LSQR / LSMR / CRAIG / CRAIG-MR with the oracle and the golden traces."""
import numpy as np
import pytest

from oracle import csr_ref, gpu_order, krylov_ref as kr, lls_ref

pytestmark = pytest.mark.gpu


def op_from(A, **kw):
    from pykrylov_amd import CsrOperator
    return CsrOperator(A.indptr, A.indices, A.data, A.shape, **kw)


def golden_csr(d, prefix):
    return csr_ref.RefCsr(d[prefix + "indptr"], d[prefix + "indices"], d[prefix + "data"], d[prefix + "shape"])


def cases():
    for tag in ("s", "l"):
        for solver in ("lsqr", "lsmr", "craig", "craigmr"):
            for btag in ("cons", "ls"):
                if solver.startswith("craig") and btag == "ls":
                    continue
                for damp, etol in ((0.0, 1e-6), (0.1, 1e-6), (0.0, 0.0)):
                    if solver.startswith("craig") and damp != 0.0:
                        continue
                    yield tag, solver, btag, damp, etol


def run_device(solver, op, b, damp, etol, **kw):
    from pykrylov_amd import lls
    if solver == "lsqr":
        s = lls.LSQRFramework(op)
        s.solve(b, damp=damp, etol=etol, **kw)
        return dict(x=s.x, istop=s.istop, itn=s.itn, r1norm=s.r1norm, r2norm=s.r2norm, Anorm=s.Anorm, Acond=s.Acond,
                    Arnorm=s.Arnorm, xnorm=s.xnorm), s
    if solver == "lsmr":
        s = lls.LSMRFramework(op)
        x, istop, itn, normr, normar, normA, condA, normx = s.solve(b, damp=damp, etol=etol, **kw)
        return dict(x=x, istop=istop, itn=itn, normr=normr, normar=normar, normA=normA, condA=condA, normx=normx), s
    if solver == "craig":
        s = lls.CRAIGFramework(op)
        s.solve(b, etol=etol, **kw)
        return dict(x=s.x, r=s.r, istop=s.istop, itn=s.itn, r1norm=s.r1norm, r2norm=s.r2norm, Arnorm=s.Arnorm,
                    xnorm=s.xnorm), s
    s = lls.CRAIGMRFramework(op)
    s.solve(b, etol=etol, **kw)
    return dict(x=s.x, istop=s.istop, itn=s.itn), s


def run_oracle(solver, A, b, damp, etol, red=None, **kw):
    At = A.transpose()
    if solver == "lsqr":
        return lls_ref.lsqr(A.matvec, At.matvec, A.shape, b.copy(), damp=damp, etol=etol, red=red, **kw)
    if solver == "lsmr":
        return lls_ref.lsmr(A.matvec, At.matvec, A.shape, b.copy(), damp=damp, etol=etol, red=red, **kw)
    if solver == "craig":
        return lls_ref.craig(A.matvec, At.matvec, A.shape, b.copy(), etol=etol, red=red, **kw)
    return lls_ref.craigmr(A.matvec, At.matvec, A.shape, b.copy(), etol=etol, red=red, **kw)


@pytest.mark.parametrize("tag,solver,btag,damp,etol", list(cases()))
def test_lls_bit_exact_with_emulated_dot_order(golden, tag, solver, btag, damp, etol, monkeypatch):
    monkeypatch.setattr(lls_ref, "_sq", lambda a: a * a)
    d = golden("lls_random.npz")
    A = golden_csr(d, tag + "_A_")
    b = d[tag + "_b_" + btag]
    got, s = run_device(solver, op_from(A), b, damp, etol)

    class Dots(gpu_order.GpuDots):
        """dots fused into an SpMV on A (rows = m) or on A' (rows = n) use that matrix's tile count"""
        def __call__(self, a, bb, site):
            if site.endswith(".beta"):
                return gpu_order.total(gpu_order.spmv_partials(a, bb, (A.shape[0] + 255) // 256))
            if site.endswith(".alpha") or site.endswith(".alpha0"):
                return gpu_order.total(gpu_order.spmv_partials(a, bb, (A.shape[1] + 255) // 256))
            return gpu_order.stream_dot(a, bb)
    ref = run_oracle(solver, A, b, damp, etol, red=kr.Reductions(Dots(0, [])))
    assert (got["istop"], got["itn"]) == (ref["istop"], ref["itn"])
    assert np.array_equal(got["x"], ref["x"])
    for k, v in got.items():
        if k not in ("x", "r", "istop", "itn"):
            assert v == ref[k], k
    if solver == "craig":
        assert np.array_equal(got["r"], ref["r"])
        assert np.allclose(s.dir_errors_d_window, ref["dir_errors_d_window"], rtol=1e-15, atol=0)
    else:
        assert len(s.dir_errors_window) == len(ref["dir_errors_window"])
        assert np.allclose(s.dir_errors_window, ref["dir_errors_window"], rtol=1e-15, atol=0)


@pytest.mark.parametrize("tag,solver,btag,damp,etol", list(cases()))
def test_lls_vs_golden(golden, tag, solver, btag, damp, etol):
    """Against the reference's own run (np.dot order): same istop family, iteration count within one, x to 1e-5 relative
    and the norm estimates to 1e-2 -- the fixtures stop at etol = 1e-6 on an ESTIMATE of the direct error, so x is only
    determined to about that accuracy, and the Golub-Kahan vectors of any two summation orders drift apart like every
    Lanczos process.  The tight statements are elsewhere: bit equality with the oracle in the device's summation order
    (test above) and <= 1e-12 against the order-independent anchor with etol = 0 (tests/test_gpu_anchors.py)."""
    d = golden("lls_random.npz")
    A = golden_csr(d, tag + "_A_")
    b = d[tag + "_b_" + btag]
    k = "%s_%s_%s_d%g_e%g_" % (tag, solver, btag, damp, etol)
    got, s = run_device(solver, op_from(A), b, damp, etol)
    # several stopping rules fire within an iteration of each other (etol window vs atol/btol): the winning
    # rule may differ by summation order, the iteration count may not by more than one
    assert got["istop"] in (int(d[k + "istop"]), 1, 2, 8) and got["istop"] > 0
    assert abs(got["itn"] - int(d[k + "itn"])) <= 1
    xref = d[k + "x"]
    # 60x40 runs ~n iterations, 2000x1500 runs 50-130: in both the Golub-Kahan vectors have lost orthogonality by
    # the end and the alpha/beta sequences of ANY two summation orders drift apart (bit-level agreement with the
    # oracle run in the device's order is the test above)
    xtol, ntol = 1e-5, 1e-2
    assert np.linalg.norm(got["x"] - xref) <= xtol * np.linalg.norm(xref)
    for name in ("Anorm", "normA"):
        if name in got and got["itn"] == int(d[k + "itn"]):
            assert abs(got[name] - float(d[k + name])) <= ntol * abs(float(d[k + name])), name


def test_lls_edge_cases(golden):
    from pykrylov_amd import lls
    d = golden("lls_random.npz")
    A = golden_csr(d, "s_A_")
    op = op_from(A)
    m, n = A.shape
    s = lls.LSQRFramework(op)
    s.solve(np.zeros(m))                                   # b = 0: x = 0, istop = 0
    assert s.istop == 0 and s.itn == 0 and np.array_equal(s.x, np.zeros(n)) and s.status == 'solution is zero'
    s.solve(d["s_b_ls"], itnlim=3, etol=0.0)
    ref = lls_ref.lsqr(A.matvec, A.transpose().matvec, A.shape, d["s_b_ls"].copy(), itnlim=3, etol=0.0)
    assert (s.istop, s.itn, s.nMatvec) == (ref["istop"], ref["itn"], 6) == (7, 3, 6)
    s.solve(d["s_b_ls"], store_iterates=True, store_resids=True)
    assert len(s.iterates) == s.itn + 1 and len(s.resids) == s.itn + 1 and np.array_equal(s.iterates[-1], s.x)
    assert op.nMatvec >= s.itn and op.T.nMatvec >= s.itn
    x, istop, itn = lls.LSMRFramework(op).solve(d["s_b_cons"], itnlim=2, etol=0.0)[:3]
    assert (istop, itn) == (7, 2)
    c = lls.CRAIGMRFramework(op)
    c.solve(d["s_b_cons"])
    assert c.x.shape == (m,)
    with pytest.raises(ValueError):                          # M is called back with m entries; this one takes n
        s.solve(d["s_b_ls"], M=op)
    with pytest.raises(NotImplementedError):
        s.solve(d["s_b_ls"], wantvar=True)


# ------------------------------------------------------------------ diagonal preconditioners M, N
def precon_cases():
    for solver in ("lsqr", "lsmr", "craig", "craigmr"):
        for btag in ("cons", "ls"):
            if solver.startswith("craig") and btag == "ls":
                continue
            for ptag in ("MN", "M", "N"):
                yield solver, btag, ptag


@pytest.mark.parametrize("solver,btag,ptag", list(precon_cases()))
def test_lls_diagonal_preconditioners(golden, solver, btag, ptag, monkeypatch):
    """`u = M(Mu)`, `v = N(Nv)` with DiagonalOperators (lsqr.py:189-190,201-202,253-254,265-266 and the same lines
    of the other three): (i) the reference's own run (fixture lls_precon.npz), (ii) bit equality with the oracle in
    the device's summation order."""
    from pykrylov_amd import DiagonalOperator
    d = golden("lls_precon.npz")
    A = golden_csr(d, "A_")
    dm, dn = d["dm"], d["dn"]
    b = d["b_" + btag]
    kw_dev, kw_ref = {}, {}
    if "M" in ptag:
        kw_dev["M"], kw_ref["M"] = DiagonalOperator(dm), (lambda u: dm * u)
    if "N" in ptag:
        kw_dev["N"], kw_ref["N"] = DiagonalOperator(dn), (lambda v: dn * v)
    got, s = run_device(solver, op_from(A), b, 0.0, 0.0, **kw_dev)
    k = "%s_%s_%s_" % (solver, btag, ptag)
    # these 60 x 40 runs take ~n iterations: the Golub-Kahan vectors lose orthogonality on the way and the tail is
    # sensitive to the summation order (see test_lls_vs_golden); istop and the solution are what is comparable
    assert got["istop"] == int(d[k + "istop"]) and abs(got["itn"] - int(d[k + "itn"])) <= 3
    # (LSMR and CRAIG-MR stop at itnlim = min(m, n) here, short of convergence: 1e-4 between summation orders)
    assert np.linalg.norm(got["x"] - d[k + "x"]) <= 1e-4 * np.linalg.norm(d[k + "x"])
    monkeypatch.setattr(lls_ref, "_sq", lambda a: a * a)

    class Dots(gpu_order.GpuDots):
        def __call__(self, a, bb, site):
            if site.endswith(".beta"):
                return gpu_order.total(gpu_order.spmv_partials(a, bb, (A.shape[0] + 255) // 256))
            if site.endswith(".alpha") or site.endswith(".alpha0"):
                return gpu_order.total(gpu_order.spmv_partials(a, bb, (A.shape[1] + 255) // 256))
            return gpu_order.stream_dot(a, bb)
    ref = run_oracle(solver, A, b, 0.0, 0.0, red=kr.Reductions(Dots(0, [])), **kw_ref)
    assert (got["istop"], got["itn"]) == (ref["istop"], ref["itn"])
    assert np.array_equal(got["x"], ref["x"])
    for key, v in got.items():
        if key not in ("x", "r", "istop", "itn"):
            assert v == ref[key], key


@pytest.mark.parametrize("solver", ["lsqr", "lsmr", "craig", "craigmr"])
@pytest.mark.parametrize("ptag", ["MN", "M", "N"])
def test_lls_general_preconditioners_through_callbacks(golden, solver, ptag, monkeypatch):
    """M and N as arbitrary callables (the reference applies them as functions, `u = M(Mu)`, `v = N(Nv)`:
    lsqr.py:190,202,254,266): here SPD tridiagonal operators, called back on the host while the loop stays on the
    device.  Bit equality with the oracle given the same callables and the device's summation order (the inner
    products that involve a callback's result are re-formed by a stream dot kernel)."""
    d = golden("lls_precon.npz")
    A = golden_csr(d, "A_")
    m, n = A.shape
    b = d["b_cons"]

    def spd(k, seed):
        rng = np.random.default_rng(seed)
        T = np.diag(2.0 + rng.random(k)) + np.diag(-0.5 * np.ones(k - 1), 1) + np.diag(-0.5 * np.ones(k - 1), -1)
        return T
    Mm, Nm = spd(m, 1), spd(n, 2)
    calls = {"M": 0, "N": 0}

    def M(u):
        calls["M"] += 1
        return Mm @ u

    def N(v):
        calls["N"] += 1
        return Nm @ v
    kw = {}
    if "M" in ptag:
        kw["M"] = M
    if "N" in ptag:
        kw["N"] = N
    got, s = run_device(solver, op_from(A), b, 0.0, 0.0, **kw)
    dev_calls = dict(calls)
    monkeypatch.setattr(lls_ref, "_sq", lambda a: a * a)

    class Dots(gpu_order.GpuDots):
        def __call__(self, a, bb, site):
            if site.endswith(".beta") and "M" not in ptag:
                return gpu_order.total(gpu_order.spmv_partials(a, bb, (m + 255) // 256))
            if (site.endswith(".alpha") or site.endswith(".alpha0")) and "N" not in ptag:
                return gpu_order.total(gpu_order.spmv_partials(a, bb, (n + 255) // 256))
            return gpu_order.stream_dot(a, bb)
    calls.update(M=0, N=0)
    ref = run_oracle(solver, A, b, 0.0, 0.0, red=kr.Reductions(Dots(0, [])), **kw)
    assert (got["istop"], got["itn"]) == (ref["istop"], ref["itn"]) and got["itn"] > 5
    assert np.array_equal(got["x"], ref["x"])
    # called as often as the reference calls them (once in the setup, once per pass)
    assert dev_calls == calls, (dev_calls, calls)


def test_lls_callback_errors_propagate(golden):
    from pykrylov_amd import lls
    d = golden("lls_precon.npz")
    A = golden_csr(d, "A_")

    def bad(v):
        if bad.n >= 3:
            raise RuntimeError("N broke")
        bad.n += 1
        return v
    bad.n = 0
    s = lls.LSQRFramework(op_from(A))
    with pytest.raises(RuntimeError, match="N broke"):
        s.solve(d["b_cons"], N=bad)
    with pytest.raises(TypeError):
        lls.LSQRFramework(op_from(A)).solve(d["b_cons"], M=object())
