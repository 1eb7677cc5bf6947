"""One rank's slab of the 8-GPU BASELINE run (rows [3n/8, 4n/8) of 512^3, columns localised to [own | lower plane |
upper plane]) built on ONE GPU through the production partitioning path (tools/slab_budget.py, loopback transport):
the localised slab must still qualify for the compact storage formats, split into interior / boundary tiles as
DESIGN.md section 5 budgets it, and run the two-launch product."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_tool():
    spec = importlib.util.spec_from_file_location("slab_budget", os.path.join(ROOT, "tools", "slab_budget.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("kind,fmt", [("varcoef", 11), ("const", 9), ("varcoef", 10), ("varcoef", 5), ("const", 4)])
def test_rank3_slab_of_512cubed(kind, fmt):
    """fmt 11 / 9: what the slab gets by itself since round 5 (11: the symmetric twin of format 10) -- the brick march, the neighbours' planes taken from the
    received entries (tests/test_gpu_slab_march.py); fmt 5 / 4: the windowed formats, still there on request."""
    from pykrylov_amd import _lib
    from pykrylov_amd.generic import DeviceRun
    sb = load_tool()
    lib, world, op = sb.build_slab(kind)
    try:
        if fmt < 9 or (kind == "varcoef" and fmt == 10):
            _lib.check(lib.mk_csr_set_format(op.handle, fmt))
        info = sb.slab_info(lib, op)
        n_local = 512 ** 3 // 8
        assert info["rows"] == n_local and info["halo"] == 2 * 512 * 512
        assert info["format"] == fmt
        # 64 planes of 1024 tiles; the first and the last plane reference received entries
        assert (info["tiles_interior"], info["tiles_boundary"]) == (65536 - 2048, 2048)
        if fmt >= 9:
            assert info["matrix_bytes_per_product"] <= {9: 1, 10: 57, 11: 33}[fmt] * n_local + 64 * 256
        else:
            assert info["tiles_windowed"] == n_local // 256                                   # every tile windowed
            if fmt == 5:
                assert info["matrix_bytes_per_product"] < 8.3 * info["nnz"]                   # 8 B / nonzero + 1 B / row + descriptors
            else:
                assert info["matrix_bytes_per_product"] < 2 * n_local
        ones = _lib.DeviceArray.from_numpy(np.ones(op.shape[1]))
        rhs = _lib.DeviceArray(n_local)
        op.spmv_device(ones.ptr, rhs.ptr)
        # rows of the first / last plane see the looped-back planes: A * 1 is what the full matrix gives on interior
        # planes, i.e. zero wherever no Dirichlet face is involved (the slab has x / y faces only)
        r = rhs.to_numpy()
        assert np.isfinite(r).all()
        if kind == "const":
            inner = r.reshape(64, 512, 512)[:, 1:-1, 1:-1]
            assert not inner.any()                                                            # 6 - 6 * 1, first and last plane included
        run = DeviceRun(op, _lib.MK_CG, rhs, None, abstol=0.0, reltol=0.0, matvec_max=1 << 60, check_curvature=1)
        run.setup()
        assert sb.slab_info(lib, op)["format"] == fmt                                         # CG keeps the march
        assert run.iterate(10) == 10
        res = run.finish()
        assert np.isfinite(res.residNorm) and res.residNorm > 0 and res.definite == 1
        run.close()
    finally:
        op.free()
        lib.mk_comm_destroy()
