"""CPU-only checks of the host logic: CSR builders, MatrixMarket reader, operator protocol,
and that libmikrylov.so loads and exports every symbol its header declares."""
import os
import sys
import re

import numpy as np
import pytest

from conftest import ROOT

REF_EXAMPLES = "/root/reference/examples"


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b)


# ------------------------------------------------------------------ C ABI surface
def header_functions():
    text = open(os.path.join(ROOT, "include", "mikrylov.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mk_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from pykrylov_amd import _lib
    lib = _lib.load()                      # loading needs no GPU
    names = header_functions()
    assert len(names) >= 35
    for name in names:
        assert hasattr(lib, name), "libmikrylov.so lacks %s" % name
        assert name in _lib.PROTOTYPES, "no ctypes prototype for %s" % name
    assert set(_lib.PROTOTYPES) == set(names)
    assert lib.mk_version() == 100


def test_library_exports_nothing_but_the_declared_entry_points():
    """-fvisibility=hidden + MK_API: the functions `nm -D` lists are exactly the entry points include/mikrylov.h declares
    (no mangled internals; what else is dynamic are HIP's kernel handles and module ids, data objects, not functions)."""
    import subprocess
    from pykrylov_amd import _lib
    _lib.load()
    so = os.path.join(ROOT, "pykrylov_amd", "libmikrylov.so")
    out = subprocess.run(["nm", "-D", "--defined-only", so], check=True, capture_output=True, text=True).stdout
    funcs = sorted(ln.split()[2] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] in "Tt")
    assert funcs == header_functions(), sorted(set(funcs) ^ set(header_functions()))


def test_struct_layouts_match_header(tmp_path):
    """sizeof/offsetof as the C compiler sees include/mikrylov.h == the ctypes mirrors."""
    import ctypes
    import subprocess
    from pykrylov_amd import _lib
    src = tmp_path / "probe.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mikrylov.h"\n'
                   'int main(void){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(mk_params), sizeof(mk_result),'
                   'offsetof(mk_params, matvec_max), offsetof(mk_params, window),'
                   'offsetof(mk_result, residNorm), offsetof(mk_result, aux));return 0;}\n')
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(t) for t in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    want = [ctypes.sizeof(_lib.MkParams), ctypes.sizeof(_lib.MkResult), _lib.MkParams.matvec_max.offset,
            _lib.MkParams.window.offset, _lib.MkResult.residNorm.offset, _lib.MkResult.aux.offset]
    assert got == want


def test_no_gpu_fails_loudly():
    from pykrylov_amd import _lib
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    lib = _lib.load()
    rc = lib.mk_init(0)
    if rc == 0:
        pytest.skip("a GPU is present")
    assert rc < 0 and b"hip" in lib.mk_last_error().lower()
    with pytest.raises(_lib.MkError):
        _lib.DeviceArray(4)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "pykrylov_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("the oracle", "").replace("an oracle", "") or f == "mk_device.h", \
                    "%s mentions the oracle" % f


# ------------------------------------------------------------------ builders
@pytest.mark.parametrize("m", [10, 20, 100])
def test_poisson2d_csr(golden, m):
    from pykrylov_amd import gallery
    d = golden("cg_poisson2d.npz")
    indptr, indices, data, shape = gallery.poisson2d_csr(m)
    assert same(indptr, d["m%d_A_indptr" % m]) and same(indices, d["m%d_A_indices" % m])
    assert same(data, d["m%d_A_data" % m]) and shape == (m * m, m * m)


@pytest.mark.parametrize("n", [10, 100, 1000])
def test_poisson1d_csr(golden, n):
    from pykrylov_amd import gallery
    d = golden("cg_poisson1d.npz")
    indptr, indices, data, _ = gallery.poisson1d_csr(n)
    assert same(indptr, d["n%d_A_indptr" % n]) and same(indices, d["n%d_A_indices" % n])
    assert same(data, d["n%d_A_data" % n])


@pytest.mark.parametrize("m", [8, 16])
def test_poisson3d_csr(golden, m):
    from pykrylov_amd import gallery
    d = golden("large_summaries.npz")
    indptr, indices, data, _ = gallery.poisson3d_csr(m)
    assert same(indptr, d["p3d%d_A_indptr" % m]) and same(indices, d["p3d%d_A_indices" % m])
    assert same(data, d["p3d%d_A_data" % m])


def test_random_diagdom_csr(golden):
    from pykrylov_amd import gallery
    d = golden("nonsym_rand10k.npz")
    indptr, indices, data, _ = gallery.random_diagdom_csr(10000, seed=1)
    assert same(indptr, d["A_indptr"]) and same(indices, d["A_indices"]) and same(data, d["A_data"])


def test_gallery_matvecs_match_csr(golden):
    from pykrylov_amd import gallery
    from oracle import csr_ref
    rng = np.random.default_rng(3)
    x = rng.standard_normal(400)
    A = csr_ref.RefCsr(*gallery.poisson2d_csr(20))
    np.testing.assert_allclose(gallery.Poisson2dMatvec(x.copy()), A.matvec(x), rtol=0, atol=1e-13)
    B = csr_ref.RefCsr(*gallery.poisson1d_csr(400))
    np.testing.assert_allclose(gallery.Poisson1dMatvec(x.copy()), B.matvec(x), rtol=0, atol=1e-13)


def test_coo_to_csr_duplicates_and_order():
    from pykrylov_amd.sparse import coo_to_csr
    rows = [2, 0, 0, 2, 1, 0]
    cols = [1, 3, 0, 1, 1, 3]
    vals = [1.0, 2.0, 3.0, 4.0, 5.0, 6.0]
    indptr, indices, data = coo_to_csr(rows, cols, vals, (3, 4))
    assert indptr.tolist() == [0, 2, 3, 4] and indices.tolist() == [0, 3, 1, 1]
    assert data.tolist() == [3.0, 8.0, 5.0, 5.0]
    with pytest.raises(ValueError):
        coo_to_csr([3], [0], [1.0], (3, 4))
    e = coo_to_csr([], [], [], (2, 2))
    assert e[0].tolist() == [0, 0, 0] and e[1].size == 0


# ------------------------------------------------------------------ MatrixMarket
@pytest.mark.skipif(not os.path.isdir(REF_EXAMPLES), reason="reference examples not on this box")
@pytest.mark.parametrize("name,fix", [("1138bus.mtx", "cg_1138bus.npz"), ("jpwh_991.mtx", "nonsym_jpwh991.npz")])
def test_read_mtx_reference_examples(golden, name, fix):
    from pykrylov_amd.mmio import read_mtx
    d = golden(fix)
    indptr, indices, data, shape, symmetric = read_mtx(os.path.join(REF_EXAMPLES, name))
    assert same(indptr, d["A_indptr"]) and same(indices, d["A_indices"]) and same(data, d["A_data"])
    assert shape == tuple(d["A_shape"]) and symmetric == (name == "1138bus.mtx")


def test_read_mtx_variants(tmp_path):
    from pykrylov_amd.mmio import read_mtx
    p = tmp_path / "a.mtx"
    p.write_text("%%MatrixMarket matrix coordinate real symmetric\n% comment\n\n3 3 3\n1 1 2.0\n3 1 -1.5\n2 2 4\n")
    indptr, indices, data, shape, symmetric = read_mtx(str(p))
    assert symmetric and shape == (3, 3) and indptr.tolist() == [0, 2, 3, 4]
    assert indices.tolist() == [0, 2, 1, 0] and data.tolist() == [2.0, -1.5, 4.0, -1.5]
    p.write_text("%%MatrixMarket matrix coordinate pattern general\n2 3 2\n1 3\n2 1\n")
    indptr, indices, data, shape, symmetric = read_mtx(str(p))
    assert not symmetric and shape == (2, 3) and indices.tolist() == [2, 0] and data.tolist() == [1.0, 1.0]
    p.write_text("%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n")
    with pytest.raises(ValueError):
        read_mtx(str(p))
    p.write_text("%%MatrixMarket matrix coordinate real general\n2 2 2\n1 1 1.0\n")
    with pytest.raises(ValueError):
        read_mtx(str(p))


# ------------------------------------------------------------------ operator protocol (host)
def test_linear_operator_protocol_matches_reference(golden):
    from pykrylov_amd import LinearOperator
    d = golden("api_contract.npz")
    B = np.array([[1.0, 2.0, 3.0], [4.0, 5.0, 6.0]])
    op = LinearOperator(3, 2, matvec=lambda v: B @ v, matvec_transp=lambda u: B.T @ u)
    v = np.ones(3)
    assert same(op * v, d["mul_result"]) and same(op.T * np.ones(2), d["T_result"])
    op * v
    assert op.nMatvec == 2 + 0 * int(d["nMatvec_after_2"]) and int(d["nMatvec_after_2"]) == 3
    assert op.shape == tuple(d["shape"]) and op.T.shape == tuple(d["T_shape"])
    assert op.T.T is op
    sym = LinearOperator(2, 2, matvec=lambda v: v, symmetric=True)
    assert (sym.T is sym) == bool(d["sym_T_is_self"])
    names = [np.dtype((op * v.astype(dt)).dtype).name
             for dt in (np.int32, np.int64, np.float32, np.float64, np.complex64, np.complex128)]
    assert names == list(d["promotion_vs_float64_op"])
    with pytest.raises(ValueError):
        op * np.ones(5)
    with pytest.raises(ValueError):
        op * [1.0, 1.0, 1.0]                       # not scalar / operator / ndarray (linop.py:369)
    op * "abc"                                     # np.isscalar(str) is True: the reference does not raise either
    assert str(d["size_mismatch_exc"]) == "ValueError" and str(d["bad_operand_exc"]) == "none"


def test_operator_algebra():
    # mirrors reference pykrylov/linop/tests/test_linop.py:162-228
    from pykrylov_amd import (LinearOperator, IdentityOperator, DiagonalOperator, ZeroOperator, ShapeError,
                              linop_from_ndarray, ReducedLinearOperator, SymmetricallyReducedLinearOperator)
    A = linop_from_ndarray(np.array([[1.0, 2.0, 3.0], [4.0, 5.0, 6.0]]))
    C = linop_from_ndarray(np.array([[1.0, 2.0], [3.0, 4.0]]), symmetric=False)
    v, u = np.ones(3), np.array([1.0, 1.0])
    assert (A * v).tolist() == [6.0, 15.0]
    assert ((C ** 2) * u).tolist() == [17.0, 37.0]
    assert ((2 * A) * v).tolist() == [12.0, 30.0] and ((A * 2) * v).tolist() == [12.0, 30.0]
    assert ((-A) * v).tolist() == [-6.0, -15.0] and ((A / 2) * v).tolist() == [3.0, 7.5]
    assert ((A + A) * v).tolist() == [12.0, 30.0] and ((A - A) * v).tolist() == [0.0, 0.0]
    assert ((C * A) * v).tolist() == [36.0, 78.0]
    assert ((A.T * C.T) * u).tolist() == ((C * A).T * u).tolist()
    assert isinstance(A * 0, ZeroOperator) and ((A * 0) * v).tolist() == [0.0, 0.0]
    assert isinstance(C ** 0, IdentityOperator) and (C ** 1) is C
    with pytest.raises(ShapeError):
        A * A
    with pytest.raises(ShapeError):
        A + C
    with pytest.raises(ShapeError):
        A ** 2
    with pytest.raises(ValueError):
        A + 3
    with pytest.raises(ValueError):
        A / C
    with pytest.raises(ValueError):
        C ** -1
    with pytest.raises(ValueError):
        C ** 1.5
    D = DiagonalOperator(np.array([1.0, 4.0]))
    assert (D * u).tolist() == [1.0, 4.0] and D.T is D
    assert (abs(DiagonalOperator(np.array([-1.0, 4.0]))) * u).tolist() == [1.0, 4.0]
    from pykrylov_amd.linop import sqrt
    assert (sqrt(D) * u).tolist() == [1.0, 2.0]
    with pytest.raises(ValueError):
        DiagonalOperator(np.ones((2, 2)))
    Z = ZeroOperator(3, 2)
    assert (Z * v).tolist() == [0.0, 0.0] and (Z.T * u).tolist() == [0.0, 0.0, 0.0]
    R = ReducedLinearOperator(A, [1], [0, 2])
    assert (R * np.array([1.0, 1.0])).tolist() == [10.0] and (R.T * np.array([1.0])).tolist() == [4.0, 6.0]
    S = SymmetricallyReducedLinearOperator(C, [1])
    assert (S * np.array([2.0])).tolist() == [8.0]
    assert A.to_array().tolist() == [[1.0, 2.0, 3.0], [4.0, 5.0, 6.0]]
    with pytest.raises(TypeError):
        A.dtype = "not a type"
    assert "Unsymmetric" in repr(A) and "(2,3)" in repr(A)
    op = LinearOperator(2, 2, matvec=lambda x: x)
    assert op.T is None and op.H is None


def test_solver_rejects_non_operators_and_needs_a_gpu_for_host_operators():
    """A matrix-free operator is accepted (its products are called back from the device loop, test_gpu_hostop.py),
    but the loop itself has no host implementation: without a GPU the solve fails loudly.  Objects that are not
    operators at all are a TypeError."""
    from pykrylov_amd import CG, LinearOperator, _lib
    with pytest.raises(TypeError):
        CG(object()).solve(np.ones(4))
    op = LinearOperator(4, 4, matvec=lambda v: 2 * v, symmetric=True)
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(_lib.MkError):
            CG(op).solve(np.ones(4))


def test_default_limits_use_the_global_size(monkeypatch):
    """On a row-partitioned operator rhs is the local slice; the default iteration limits must come from the GLOBAL
    size (cg.py:79-80 `2n`, minres.py:128 `5n`, symmlq.py:90 `2n+2`), identical on every rank, or ranks with
    unequal blocks stop enqueuing collectives at different passes."""
    import numpy as np
    import pykrylov_amd as pk
    from pykrylov_amd import generic, cg, minres, symmlq
    from pykrylov_amd.linop import CsrOperator

    seen = {}

    class Stop(Exception):
        pass

    class FakeRun(object):
        def __init__(self, op, kind, rhs, guess=None, precon_diag=None, **params):
            seen.update(params)
            raise Stop()

    for mod in (generic, cg, minres, symmlq):
        monkeypatch.setattr(mod, "DeviceRun", FakeRun)
    op = object.__new__(CsrOperator)
    op.__dict__.update(global_size=1001, local_size=334, _nMatvec=0)
    monkeypatch.setattr(CsrOperator, "shape", property(lambda self: (334, 336)), raising=False)
    rhs = np.ones(334)
    for cls, key, want in ((pk.CG, "matvec_max", 2002), (pk.BiCGSTAB, "matvec_max", 2002),
                           (pk.CGS, "matvec_max", 2002), (pk.TFQMR, "matvec_max", 2002),
                           (pk.Symmlq, "matvec_max", 2004), (pk.Minres, "itnlim", 5005)):
        seen.clear()
        s = cls(op)
        try:
            s.solve(rhs, show=False, check=False) if cls is pk.Minres else s.solve(rhs)
        except Stop:
            pass
        assert seen.get(key) == want, (cls.__name__, seen)


def test_integration_md_stub_matches_the_header():
    """INTEGRATION.md section 2b shows the ctypes structures a reference maintainer would paste: they must have the
    fields and sizes of include/mikrylov.h (a stale stub is rejected by the `struct_size` check at run time)."""
    import ctypes
    import re
    from pykrylov_amd import _lib
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    hdr = open(os.path.join(ROOT, "include", "mikrylov.h")).read()

    def md_fields(cls):
        body = md[md.index("class %s(ctypes.Structure)" % cls):]
        body = body[body.index("_fields_"):body.index("]\n") + 1]
        return re.findall(r'\("(\w+)",\s*ctypes\.(c_\w+)(?:\s*\*\s*(\d+))?\)', body)

    def hdr_fields(name):
        body = hdr[:hdr.index("} %s;" % name)]
        body = body[body.rindex("typedef struct {"):]
        return re.findall(r"^\s*(int32_t|int64_t|double)\s+([\w\s,\[\]]+);", body, re.M)

    ctype = {"int32_t": "c_int32", "int64_t": "c_int64", "double": "c_double"}
    for cls, cname, ref in (("MkParams", "mk_params", _lib.MkParams), ("MkResult", "mk_result", _lib.MkResult)):
        want = []
        for typ, names in hdr_fields(cname):
            for nm in names.split(","):
                nm = nm.strip()
                m = re.match(r"(\w+)\[(\d+)\]", nm)
                want.append((m.group(1), ctype[typ], m.group(2)) if m else (nm, ctype[typ], ""))
        got = md_fields(cls)
        assert got == want, (cls, got, want)
        fields = [(n, getattr(ctypes, t) * int(k) if k else getattr(ctypes, t)) for n, t, k in got]
        stub = type(cls, (ctypes.Structure,), {"_fields_": fields})
        assert ctypes.sizeof(stub) == ctypes.sizeof(ref)


def test_device_history_fixture_for_the_multi_gpu_line():
    """tests/golden/dev_hist_512.npz (device-generated on ONE MI355X by tools/make_dev_hist.py) is what bench.py's N > 1
    line compares its first 60 passes with: it must hold both 512^3 workloads and the 64^3 twins the smoke test uses."""
    import bench
    z = np.load(bench.DEV_HIST, allow_pickle=False)
    assert int(z["passes"]) == bench.PARITY_PASSES == 60 and int(z["seed"]) == bench.VARCOEF_SEED
    for wl in ("poisson3d-512-varcoef", "poisson3d-512", "poisson3d-64-varcoef", "poisson3d-64"):
        h = bench.n1_history(wl)
        assert h is not None and h.shape == (61,) and np.isfinite(h).all() and h[-1] < h[0]
    # ||A 1||^2 of the constant-coefficient matrix is an integer (faces 1, edges 4, corners 9): exact in any order
    for m in (64, 512):
        assert bench.n1_history("poisson3d-%d" % m)[0] == np.sqrt(6.0 * (m - 2) ** 2 + 48.0 * (m - 2) + 72.0)
    assert bench.n1_history("poisson2d-1000") is None
    assert abs(bench.rel_hist_err([4.0, 2.0], [4.0, 2.0 + 2e-12]) - 1e-12) < 1e-15


def test_bench_least_squares_matrix_is_canonical_csr():
    """bench.random_tall_csr (the 4e6 x 1e6 matrix of the least-squares loops, here small): sorted, distinct columns in range, k
    entries per row, reproducible from the seed -- what CsrOperator requires of caller arrays."""
    import bench
    indptr, indices, data = bench.random_tall_csr(4000, 1000, k=5, seed=11)
    assert np.array_equal(indptr, np.arange(4001) * 5) and indices.shape == data.shape == (20000,)
    cols = indices.reshape(4000, 5)
    assert (np.diff(cols, axis=1) > 0).all() and cols.min() >= 0 and cols.max() < 1000
    again = bench.random_tall_csr(4000, 1000, k=5, seed=11)
    assert all(np.array_equal(a, b) for a, b in zip((indptr, indices, data), again))
    assert set(bench.LOOP_BYTES) == {"bicgstab", "cgs", "tfqmr", "minres", "symmlq", "lsqr", "lsmr", "craig", "craigmr"}


# ---------------------------------------------------------------------------------------------------------------------
# bench.py's compact line (the driver keeps an 8 KB tail of stdout; a 32 KB line cost round 4 its record)
# ---------------------------------------------------------------------------------------------------------------------
def _canned_cg_block(bench, wname, fmt, n_ranks=1):
    m = int(wname.split("-")[1])
    n = m ** 3 if "3d" in wname or "stencil" in wname else m * m
    blk = {"workload": wname, "value": 438.31234567891234, "unit": "iterations/s", "steps": 1000, "warmup": 50,
           "ms_per_step": 2.2815123456789, "rows": n, "nnz": 7 * n - 6 * m * m,
           "roofline": {"bound": "hbm", "kernel": "mk_spmv_kernel<CgSpmvEpiT,MkNoGate,false,%d> (SpMV + fused <p,Ap>)" % fmt,
                        "kernel_format": bench.FMT_NAMES[fmt], "achieved": 3019.123456789, "peak": 8000.0, "unit": "GB/s",
                        "frac": 0.37739043209876, "traffic": {"bytes": 4712632246, "read_bytes": 3638830073},
                        "traffic_bytes": 4712632246, "traffic_note": "x" * 200, "bytes_per_launch": 2323644416,
                        "avg_launch_us": 769.61234567, "launches_timed": 200, "method": "y" * 100, "note": "z" * 400,
                        "csr_bytes_per_launch": 13939769348, "csr_equivalent_GBs": 18112.123},
           "iteration_roofline": {"bytes_per_iter": 10913579008, "achieved_GBs": 4783.2, "frac_of_aggregate_hbm": 0.5979,
                                  "note": "n" * 300},
           "storage_format": {"format": fmt, "format_name": bench.FMT_NAMES[fmt], "tiles_windowed": 524288, "grid": 1792},
           "device_loop_ms": 2281.5,
           "residual": {"first": 1254.1, "last": 3.1e-5, "recurrence": 3.123456789e-5, "true": 3.123456788e-5,
                        "rel_gap": 8.1234e-18, "passes": 1050, "note": "r" * 250, "ok": True},
           "parity_vs_n1": {"passes": 60, "fixture": "tests/golden/dev_hist_512.npz", "fixture_has_workload": True,
                            "rel_hist_err": 1.23456789e-15, "bit_equal": False, "tolerance": 1e-12, "ok": True},
           "placement_draws": {"count": 1, "chosen": 0, "per_draw_ms_per_pass": [], "probe_seconds": 0.0},
           "comm": None}
    if n_ranks > 1:
        blk["comm"] = {"per_rank": [{"last_overlapped_halo_group_us": 31.5 + r, "device_loop_ms_per_step": 0.412345678,
                                     "product_alone_us": 251.123456 + r, "exchange_alone_us": 29.87654321,
                                     "allreduce_2048_doubles_us": 17.123456789, "allreduces_per_step": 2}
                                    for r in range(n_ranks)], "tiles_interior_boundary_rank0": [63488, 2048]}
    return blk


def _canned_detail(bench, n_ranks):
    head = "poisson3d-512"
    d = {"metric": bench.METRIC, "n_gpus": n_ranks, "steps": 1000, "warmup": 50, "scaling": "strong", "dtype": "f64",
         "data": "synthetic", "headline": head, "baseline_config": bench.BASELINE_CONFIG[head],
         "parallelism": "1 GPU" if n_ranks == 1 else "row-partition x%d, halo exchange + allreduce(dots), host-staged gloo "
                        "(RCCL communicator unavailable (rank 0: RuntimeError('%s')): host-staged gloo fallback)" % (n_ranks, "e" * 300),
         "workloads": {head: _canned_cg_block(bench, head, 4, n_ranks)}, "kernel_source_sha": "0" * 16}
    cpu = {"value": 0.93123456, "unit": "iterations/s", "cores": 1, "kind": "port", "extrapolated": False,
           "sample_rows": 134217728, "sample_nnz": 937951232, "seconds_per_pass": 1.07, "sample": "s" * 300, "host_cpus": 128}
    if n_ranks == 1:
        d["workloads"][head]["cpu_baseline"] = dict(cpu)
        d["workloads"][head]["cpu_baseline_all_cores"] = dict(cpu, cores=64, value=1.49)
        d["workloads"]["poisson3d-512-varcoef"] = dict(_canned_cg_block(bench, "poisson3d-512-varcoef", 5),
                                                       cpu_baseline=dict(cpu), cpu_baseline_all_cores=dict(cpu, cores=64))
        for w, f in (("poisson2d-1000", 4), ("stencil27-256", 8), ("stencil27-256-varcoef", 7)):
            d["workloads"][w] = _canned_cg_block(bench, w, f)
        d["solver_loops"] = {}
        for key in bench.LOOP_BYTES:
            lab = {"bicgstab": "bicgstab-rand1m@1", "cgs": "cgs-rand1m@1", "tfqmr": "tfqmr-rand1m@1",
                   "minres": "minres-shifted2d-2000@1", "symmlq": "symmlq-shifted2d-2000@1"}.get(key, key + "-rand4m-x-1m@1")
            d["solver_loops"][lab] = {"value": 9545.151661382939, "iteration_roofline": {"frac": 0.3627153145104236},
                                      "products": {"first (A p / A y)": {"avg_product_us": 37.48055458068848}}}
    else:
        d["transport"] = {"kind": "host-staged", "rccl_ranks_seen": 0, "halo_communicator_split": False}
        d["per_rank_budget"] = {"kernels_per_pass_us": 300, "source": "profiles/r03_slab_budget.txt"} if n_ranks == 8 else None
        comm = d["workloads"][head]["comm"]
        d["exchange"] = {"halo": {"value": 1234.5678, "ms_per_step": 0.81, "steps": 1000, "comm": comm},
                         "allgather": {"value": 234.5678, "ms_per_step": 4.26, "steps": 200, "comm": comm, "note": "a" * 200}}
    return d


@pytest.mark.parametrize("n_ranks", [1, 2, 8])
def test_bench_compact_line_fits_the_drivers_tail(n_ranks):
    """`bench.compact_line` on canned results with worst-case string lengths: one JSON object of <= 4096 bytes that
    carries the contract keys, `roofline` and (N = 1) `cpu_baseline`, for the N = 1, 2 and 8 shapes of the detail dict."""
    import json
    import bench
    line = bench.compact_line(_canned_detail(bench, n_ranks))
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) <= bench.LINE_LIMIT == 4096, len(text)
    back = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "iteration_frac", "residual", "parity_vs_n1",
              "placement_draws"):
        assert k in back, k
    assert back["n_gpus"] == n_ranks and back["config"]["workload"] == "CG poisson3d-512" and back["dtype"] == "f64"
    assert back["metric"] == bench.METRIC and back["vs_baseline"] is None and back["higher_is_better"] is True
    roof = back["roofline"]
    assert set(roof) >= {"bound", "kernel", "achieved", "peak", "unit", "frac", "bytes_per_launch", "avg_launch_us",
                         "traffic", "traffic_bytes"}
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and len(roof["kernel"]) <= 80
    assert roof["traffic"] == 4712632246 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert back["residual"]["ok"] is True and back["parity_vs_n1"]["ok"] is True
    if n_ranks == 1:
        cb = back["cpu_baseline"]
        assert set(cb) >= {"value", "unit", "cores", "kind", "sample"} and cb["kind"] == "port" and cb["cores"] == 1
        second = back["second_workload"]
        assert second["workload"] == "CG poisson3d-512-varcoef" and "roofline" in second and "cpu_baseline" in second
        assert set(back["solver_loops"]) == {"cg"} | {k + s for k, s in zip(
            bench.LOOP_BYTES, ["-rand1m"] * 3 + ["-shifted2d-2000"] * 2 + ["-rand4m-x-1m"] * 4)}
        assert set(back["cg_other_workloads"]) == {"poisson2d-1000", "stencil27-256", "stencil27-256-varcoef"}
    else:
        assert back["transport"]["kind"] == "host-staged"
        assert set(back["exchange"]) == {"halo", "allgather"}
        assert back["exchange"]["halo"]["max_over_ranks"]["product_alone_us"] == pytest.approx(251.123456 + n_ranks - 1, rel=1e-4)


def test_bench_compact_line_on_a_real_detail_file():
    """The same on what a real MI355X run wrote (tests/golden/bench_detail_n1.json = bench_detail.json of the default
    command, round 5): the line rebuilt from the detail equals the line the run printed, fits, and carries the figures."""
    import json
    import bench
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_detail_n1.json")))
    printed = d.pop("line")
    line = bench.compact_line(d)
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) <= bench.LINE_LIMIT
    for k in ("metric", "value", "unit", "n_gpus", "steps", "ms_per_step", "config", "iteration_frac", "residual",
              "parity_vs_n1", "solver_loops", "cg_other_workloads"):
        assert line[k] == printed[k], k
    assert line["parity_vs_n1"]["fixture"] == "device-generated"       # (a device history: regression evidence, not oracle parity)
    assert line["roofline"].get("traffic_source", "").startswith("profiles/spmv_traffic.json") or not line["roofline"]["traffic"]
    assert line["config"]["workload"] == "CG poisson3d-512" and line["roofline"]["frac"] == printed["roofline"]["frac"]
    assert line["second_workload"]["workload"] == "CG poisson3d-512-varcoef"
    assert line["cpu_baseline"]["sample_rows"] == 134217728 and line["cpu_baseline"]["extrapolated"] is False


def test_stale_binary_is_refused(tmp_path):
    """libmikrylov.so is git-ignored and travels prebuilt: `mk_build_info()` carries the digest of the sources it was compiled
    from (pykrylov_amd/build.py `source_sha`, compiled into csrc/mk_buildinfo.hip) and `_lib.load()` refuses a binary whose
    digest is not the tree's.  Here: a copy of the tree with ONE header changed by a comment must refuse the (unchanged)
    binary, load it with MIKRYLOV_ALLOW_STALE=1, and the real tree must load and report its own digest."""
    import shutil
    import subprocess
    src = os.path.join(ROOT, "pykrylov_amd")
    if not os.path.exists(os.path.join(src, "libmikrylov.so")):
        pytest.skip("library not built")
    from pykrylov_amd import _lib, build
    assert _lib.build_info() == build.source_sha() and len(build.source_sha()) == 16
    dst = tmp_path / "tree"
    (dst / "pykrylov_amd" / "csrc").mkdir(parents=True)
    (dst / "include").mkdir()
    for f in os.listdir(src):
        if f.endswith(".py"):
            shutil.copy(os.path.join(src, f), dst / "pykrylov_amd" / f)
    for f in os.listdir(os.path.join(src, "csrc")):
        shutil.copy(os.path.join(src, "csrc", f), dst / "pykrylov_amd" / "csrc" / f)
    shutil.copytree(os.path.join(src, "lls"), dst / "pykrylov_amd" / "lls")
    shutil.copy(os.path.join(ROOT, "include", "mikrylov.h"), dst / "include" / "mikrylov.h")
    os.symlink(os.path.join(src, "libmikrylov.so"), dst / "pykrylov_amd" / "libmikrylov.so")
    code = "import sys; sys.path.insert(0, %r); from pykrylov_amd import _lib; _lib.load(); print('LOADED', _lib.build_info())" % str(dst)
    env = {k: v for k, v in os.environ.items() if k not in ("MIKRYLOV_ALLOW_STALE", "MIKRYLOV_LIB")}
    ok = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=str(tmp_path))
    assert ok.returncode == 0 and "LOADED " + build.source_sha() in ok.stdout, ok.stderr[-2000:]     # the copy is current
    with open(dst / "pykrylov_amd" / "csrc" / "mk_spmv_fmt9.h", "a") as fh:
        fh.write("// an edit the binary does not know about\n")
    bad = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=str(tmp_path))
    assert bad.returncode != 0 and "built from other sources" in bad.stderr, bad.stdout + bad.stderr[-2000:]
    forced = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(env, MIKRYLOV_ALLOW_STALE="1"),
                            cwd=str(tmp_path))
    assert forced.returncode == 0 and "LOADED" in forced.stdout, forced.stderr[-2000:]
