"""The round-3 device features against fixtures produced by RUNNING THE REFERENCE (tests/golden/make_golden_r3.py):
reduced operators and block operators as device composites (products bit for bit: the device's row sums have the
reference's rounding sequence), matrix preconditioners applied on the device, CG on a variable-coefficient matrix
(histories within north_star's 1e-12 of the reference's own np.dot order)."""
import numpy as np
import pytest

from oracle import csr_ref
from conftest import rel_hist_err

pytestmark = pytest.mark.gpu


def csr_from(d, prefix):
    return csr_ref.RefCsr(d[prefix + "indptr"], d[prefix + "indices"], d[prefix + "data"], d[prefix + "shape"])


def dev(R, symmetric=False):
    from pykrylov_amd import CsrOperator
    return CsrOperator(R.indptr, R.indices, R.data, R.shape, symmetric=symmetric)


def test_reduced_and_block_device_composites_have_the_references_bits(golden):
    from pykrylov_amd import DiagonalOperator, Minres, ReducedLinearOperator, SymmetricallyReducedLinearOperator
    from pykrylov_amd.blkop import BlockLinearOperator
    from pykrylov_amd.linop import _BlockCsrOperator, _ReducedCsrOperator
    d = golden("round3_ops.npz")
    op = dev(csr_from(d, "red_A_"))
    red = ReducedLinearOperator(op, d["red_rows"], d["red_cols"])
    assert isinstance(red, _ReducedCsrOperator) and red.shape == tuple(d["red_shape"])
    assert np.array_equal(red * d["red_x"], d["red_y"]) and np.array_equal(red.T * d["red_u"], d["red_yt"])
    p = dev(csr_from(d, "sred_A_"), True)
    sred = SymmetricallyReducedLinearOperator(p, d["sred_idx"])
    assert isinstance(sred, _ReducedCsrOperator) and np.array_equal(sred * d["sred_x"], d["sred_y"])
    A, B = dev(csr_from(d, "sp_A_"), True), dev(csr_from(d, "sp_B_"))
    K = BlockLinearOperator([[A, B.T], [DiagonalOperator(d["sp_d"])]], symmetric=True)
    view = K._device_view()
    assert isinstance(view, _BlockCsrOperator)
    assert np.array_equal(view * d["sp_x"], d["sp_Kx"]) and np.array_equal(view * np.ones(K.shape[1]), d["sp_rhs"])
    s = Minres(K)
    s.solve(d["sp_rhs"], show=False, check=False, etol=0.0, rtol=1e-10)
    assert (s.istop, s.itn) == (int(d["sp_minres_istop"]), int(d["sp_minres_itn"]))
    assert rel_hist_err(s.residHistory, d["sp_minres_hist"]) <= 1e-12
    assert np.linalg.norm(s.x - d["sp_minres_x"]) <= 1e-10 * np.linalg.norm(d["sp_minres_x"])
    for o in (op, p, A, B):
        o.free()


def test_matrix_preconditioners_on_the_device_against_the_reference(golden):
    import pykrylov_amd
    from pykrylov_amd.generic import DevicePrecon
    d = golden("round3_ops.npz")
    V, MV = dev(csr_from(d, "pc_A_"), True), dev(csr_from(d, "pc_M_"), True)
    rhs = d["pc_rhs"]
    s = pykrylov_amd.CG(V, precon=MV)
    assert isinstance(s._device_precon(MV), DevicePrecon)
    s.solve(rhs, matvec_max=60)
    assert s.nMatvec == int(d["pc_cg_nMatvec"]) and rel_hist_err(s.residHistory, d["pc_cg_hist"]) <= 1e-12
    assert np.linalg.norm(s.x - d["pc_cg_x"]) <= 1e-12 * np.linalg.norm(d["pc_cg_x"])
    s = pykrylov_amd.Minres(V)
    s.solve(rhs, precon=MV, show=False, check=False, etol=0.0, rtol=1e-10)
    assert (s.istop, s.itn) == (int(d["pc_minres_istop"]), int(d["pc_minres_itn"]))
    assert rel_hist_err(s.residHistory, d["pc_minres_hist"]) <= 1e-12
    assert np.linalg.norm(s.x - d["pc_minres_x"]) <= 1e-11 * np.linalg.norm(d["pc_minres_x"])
    s = pykrylov_amd.Symmlq(V, precon=MV)
    s.solve(rhs)
    assert abs(s.nMatvec - int(d["pc_symmlq_nMatvec"])) <= 1
    assert np.linalg.norm(s.x - d["pc_symmlq_x"]) <= 1e-8 * np.linalg.norm(d["pc_symmlq_x"])
    W, MW = dev(csr_from(d, "pn_A_")), dev(csr_from(d, "pn_M_"))
    for name, cls in (("bicgstab", pykrylov_amd.BiCGSTAB), ("cgs", pykrylov_amd.CGS), ("tfqmr", pykrylov_amd.TFQMR)):
        s = cls(W, reltol=1e-10, precon=MW)
        s.solve(d["pn_rhs"], matvec_max=400)
        k = "pn_%s_" % name
        # (a stopping test on ~1e-9 ||r0|| may move by one product with the summation order)
        assert abs(s.nMatvec - int(d[k + "nMatvec"])) <= 2 and bool(s.converged) == bool(d[k + "converged"]), name
        assert np.linalg.norm(s.x - d[k + "x"]) <= 1e-6 * np.linalg.norm(d[k + "x"]), name
    for o in (V, MV, W, MW):
        o.free()


def test_variable_coefficient_cg_against_the_reference(golden):
    from pykrylov_amd import CG, gallery
    from test_gpu_formats import fmt_info
    d = golden("round3_ops.npz")
    op = gallery.poisson3d_varcoef(24, 16, 8, seed=7)                       # generated in HBM
    ip, ix, dv = op.to_csr_arrays()
    assert np.array_equal(ip, d["vc_A_indptr"]) and np.array_equal(ix, d["vc_A_indices"]) and np.array_equal(dv, d["vc_A_data"])
    assert fmt_info(op)["fmt"] == 5
    s = CG(op)
    s.solve(d["vc_rhs"])
    assert s.converged and s.nMatvec == int(d["vc_cg_nMatvec"])
    assert rel_hist_err(s.residHistory, d["vc_cg_hist"]) <= 1e-12
    assert np.linalg.norm(s.x - d["vc_cg_x"]) <= 1e-12 * np.linalg.norm(d["vc_cg_x"])
    op.free()
