"""Worker of tests/test_gpu_multirank.py: several ranks share ONE GPU (launched with torch.distributed.run, backend
gloo) and run the row-partitioned device solvers through the host-staged transport (mk_comm_init_host).  Everything
except RCCL itself is the production multi-GPU path: partition plans, column localisation on the device, the pack
kernel, the [own | halo] vector layout, the all-reduce of the per-workgroup partial sums and the halting protocol
taking identical decisions on every rank."""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: F401,E402  (before libmikrylov: DESIGN.md section 5)
import torch.distributed as td  # noqa: E402

from oracle import csr_ref, krylov_ref as kr, lls_ref  # noqa: E402
from pykrylov_amd import _lib, dist  # noqa: E402


def rel_hist_err(h, href):
    h, href = np.asarray(h), np.asarray(href)
    if len(h) != len(href):
        return float("inf")
    return float(np.max(np.abs(h - href) / np.maximum(href, 1e-4 * href[0])))


def gather_x(world, x_local):
    parts = world.allgather_object(np.asarray(x_local))
    return np.concatenate(parts)


def main():
    td.init_process_group(backend="gloo")
    rank, nranks = td.get_rank(), td.get_world_size()
    _lib.init(0)                                            # every rank on GPU 0
    world = dist.World(rank, nranks, td)
    world.init_device_comm(transport="host")
    from pykrylov_amd import CG, BiCGSTAB, CGS, TFQMR, Minres, Symmlq, DiagonalOperator
    out = {}

    # ---- CG on the 3-D Poisson matrix generated per rank in HBM (the bench.py path), both exchange modes
    m = 24
    A = csr_ref.poisson3d(m)
    n = m ** 3
    rhs = A.matvec(np.ones(n))
    ref = kr.cg(A, rhs)
    for mode in ("halo", "allgather"):
        op, ranges = dist.partition_poisson3d(world, m, m, m, mode=mode)
        c0, c1 = ranges[rank]
        ni, nb_ = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(_lib.load().mk_csr_overlap_info(op.handle, ctypes.byref(ni), ctypes.byref(nb_)))
        if mode == "halo":          # the product runs as interior launch + boundary launch (1 or 2 neighbour planes)
            assert ni.value > 0 and nb_.value in (3, 6) and ni.value + nb_.value == (c1 - c0 + 255) // 256, \
                (ni.value, nb_.value)
        else:
            assert ni.value == 0 and nb_.value == 0
        s = CG(op)
        s.solve(rhs[c0:c1])
        x = gather_x(world, s.x)
        out["cg3d/" + mode] = dict(nMatvec=int(s.nMatvec), ref=int(ref["nMatvec"]),
                                   hist_err=rel_hist_err(s.residHistory, ref["residHistory"]),
                                   x_err=float(np.linalg.norm(x - ref["x"]) / np.linalg.norm(ref["x"])),
                                   halo=int(op.halo_size))
        # warm start + matvec_max: setup product goes through the exchange as well
        g = 1.0 + np.arange(n) / n
        s2 = CG(op)
        s2.solve(rhs[c0:c1], guess=g[c0:c1], matvec_max=30)
        ref2 = kr.cg(A, rhs, guess=g, matvec_max=30)
        out["cg3d_guess/" + mode] = dict(nMatvec=int(s2.nMatvec), ref=int(ref2["nMatvec"]),
                                         hist_err=rel_hist_err(s2.residHistory, ref2["residHistory"]), x_err=0.0)
        op.free()

    # ---- the brick march (storage formats 9 / 10) on the ranks' slabs: neighbours' planes from the received entries, the
    # product as interior planes + boundary planes around the exchange (tests/test_gpu_slab_march.py has the bit-level checks)
    nxm, nym, nzm = 128, 8, 14 * nranks
    Am = csr_ref.poisson3d(nxm, nym, nzm)
    rhs_m = Am.matvec(np.ones(Am.shape[0]))
    ref_m = kr.cg(Am, rhs_m, matvec_max=40)
    for name, seed, fmt in (("cg3d_march9", None, 9), ("cg3d_march10", 5, 10), ("cg3d_march11", 5, 11)):
        if seed is not None and fmt == 10:
            Am = csr_ref.poisson3d_varcoef(nxm, nym, nzm, seed=seed)
            rhs_m = Am.matvec(np.ones(Am.shape[0]))
            ref_m = kr.cg(Am, rhs_m, matvec_max=40)
        op, ranges = dist.partition_poisson3d(world, nxm, nym, nzm, mode="halo", varcoef_seed=seed)
        _lib.check(_lib.load().mk_csr_set_format(op.handle, fmt))
        c0, c1 = ranges[rank]
        s = CG(op)
        s.solve(rhs_m[c0:c1], matvec_max=40)
        got = ctypes.c_int32()
        _lib.check(_lib.load().mk_csr_format_info(op.handle, ctypes.byref(got), None, None, None, None))
        x = gather_x(world, s.x)
        out[name + "/halo"] = dict(nMatvec=int(s.nMatvec), ref=int(ref_m["nMatvec"]), fmt=int(got.value),
                                   hist_err=rel_hist_err(s.residHistory, ref_m["residHistory"]),
                                   x_err=float(np.linalg.norm(x - ref_m["x"]) / np.linalg.norm(ref_m["x"])))
        op.free()

    # ---- general matrices through the NumPy partition plans (pack kernel with a gather list)
    B = csr_ref.random_diagdom(2003, seed=3)              # prime: the all-gather path pads the last rank's block
    nb = B.shape[0]
    rhs_b = B.matvec(np.ones(nb))
    dgl = 1.0 / np.array([B.data[B.indptr[i]:B.indptr[i + 1]][B.indices[B.indptr[i]:B.indptr[i + 1]] == i][0]
                          for i in range(nb)])
    for mode in ("halo", "allgather"):
        for name, cls, fn in (("bicgstab", BiCGSTAB, kr.bicgstab), ("cgs", CGS, kr.cgs), ("tfqmr", TFQMR, kr.tfqmr)):
            op, ranges = dist.partition_host_csr(world, B.indptr, B.indices, B.data, nb, mode=mode)
            c0, c1 = ranges[rank]
            s = cls(op, reltol=1e-8)
            s.solve(rhs_b[c0:c1], matvec_max=200)
            r = fn(B, rhs_b, reltol=1e-8, matvec_max=200)
            x = gather_x(world, s.x)
            out["%s/%s" % (name, mode)] = dict(nMatvec=int(s.nMatvec), ref=int(r["nMatvec"]), hist_err=0.0,
                                               x_err=float(np.linalg.norm(x - r["x"]) / np.linalg.norm(r["x"])),
                                               conv=bool(s.converged))
            op.free()
        # diagonal preconditioner, sliced like every other vector
        op, ranges = dist.partition_host_csr(world, B.indptr, B.indices, B.data, nb, mode=mode)
        c0, c1 = ranges[rank]
        s = BiCGSTAB(op, reltol=1e-8, precon=DiagonalOperator(dgl[c0:c1]))
        s.solve(rhs_b[c0:c1], matvec_max=200)
        r = kr.bicgstab(B, rhs_b, reltol=1e-8, matvec_max=200, precon=lambda v: dgl * v)
        x = gather_x(world, s.x)
        out["bicgstab_precon/" + mode] = dict(nMatvec=int(s.nMatvec), ref=int(r["nMatvec"]), hist_err=0.0,
                                              x_err=float(np.linalg.norm(x - r["x"]) / np.linalg.norm(r["x"])))
        op.free()

    # ---- rank-local block-Jacobi preconditioners as DEVICE operators on partitioned matrices:
    # tools.block_jacobi of the partitioned operator inverts the blocks of this rank's diagonal block; the oracle applies
    # the same block-diagonal matrix (all ranks' blocks side by side) through its scalar CSR loop
    from pykrylov_amd.tools import block_jacobi
    bs = 4

    def local_bj(R, c0, c1):
        """NumPy twin of tools.block_jacobi for the diagonal block [c0, c1) of the RefCsr R."""
        nl = c1 - c0
        rows = np.repeat(np.arange(R.shape[0], dtype=np.int64), np.diff(R.indptr))
        cols = R.indices.astype(np.int64)
        sel = (rows >= c0) & (rows < c1) & (cols >= c0) & (cols < c1)
        r, c, v = rows[sel] - c0, cols[sel] - c0, R.data[sel]
        nblk = (nl + bs - 1) // bs
        inblk = (r // bs) == (c // bs)
        blocks = np.zeros((nblk, bs, bs))
        blocks[r[inblk] // bs, r[inblk] % bs, c[inblk] % bs] = v[inblk]
        tail = nl - (nblk - 1) * bs
        if tail < bs:
            k = np.arange(tail, bs)
            blocks[nblk - 1, k, k] = 1.0
        inv = np.linalg.inv(blocks)
        rr = np.repeat(np.arange(nblk * bs, dtype=np.int64), bs)
        cc = (rr // bs) * bs + np.tile(np.arange(bs, dtype=np.int64), nblk * bs)
        vv = inv.reshape(-1)
        keep = (rr < nl) & (cc < nl)
        return csr_ref.from_coo(rr[keep], cc[keep], vv[keep], (nl, nl))

    def oracle_precon(R, ranges):
        blocks = [(c0, c1, local_bj(R, c0, c1)) for c0, c1 in ranges]
        return lambda v: np.concatenate([P.matvec(v[c0:c1]) for c0, c1, P in blocks])

    P2 = csr_ref.poisson2d(37)
    n2 = P2.shape[0]
    rhs_p = P2.matvec(np.ones(n2))
    for mode in ("halo", "allgather"):
        for name, cls, fn, R, rhs_r, kw in (
                ("bicgstab", BiCGSTAB, kr.bicgstab, B, rhs_b, dict(matvec_max=200)),
                ("cgs", CGS, kr.cgs, B, rhs_b, dict(matvec_max=200)),
                ("tfqmr", TFQMR, kr.tfqmr, B, rhs_b, dict(matvec_max=200)),
                ("cg", CG, kr.cg, P2, rhs_p, dict(matvec_max=40))):
            op, ranges = dist.partition_host_csr(world, R.indptr, R.indices, R.data, R.shape[0], mode=mode)
            c0, c1 = ranges[rank]
            M = block_jacobi(op, bs)                         # rank local, on the device
            assert M.shape == (c1 - c0, c1 - c0)
            mi, mj, mv = M.to_csr_arrays()
            Mref = local_bj(R, c0, c1)
            assert np.array_equal(mi, Mref.indptr) and np.array_equal(mj, Mref.indices) and np.array_equal(mv, Mref.data)
            s = cls(op, reltol=1e-8, precon=M)
            s.solve(rhs_r[c0:c1], **kw)
            r = fn(R, rhs_r, reltol=1e-8, precon=oracle_precon(R, ranges), **kw)
            x = gather_x(world, s.x)
            ent = dict(nMatvec=int(s.nMatvec), ref=int(r["nMatvec"]), hist_err=0.0,
                       x_err=float(np.linalg.norm(x - r["x"]) / np.linalg.norm(r["x"])))
            if name == "cg":
                ent["hist_err"] = rel_hist_err(s.residHistory, r["residHistory"])
            out["%s_bjacobi/%s" % (name, mode)] = ent
            M.free()
            op.free()
        # MINRES with the SPD block-Jacobi of the 2-D Laplacian
        op, ranges = dist.partition_host_csr(world, P2.indptr, P2.indices, P2.data, n2, mode=mode)
        c0, c1 = ranges[rank]
        M = block_jacobi(op, bs)
        s = Minres(op)                                       # (a `solve` keyword in the reference, minres.py:121)
        s.solve(rhs_p[c0:c1], show=False, check=False, etol=0.0, rtol=1e-10, precon=M)
        r = kr.minres(P2, rhs_p, check=False, etol=0.0, rtol=1e-10, precon=oracle_precon(P2, ranges))
        x = gather_x(world, s.x)
        out["minres_bjacobi/" + mode] = dict(nMatvec=int(s.itn), ref=int(r["itn"]),
                                             hist_err=rel_hist_err(s.residHistory, r["residHistory"]),
                                             x_err=float(np.linalg.norm(x - r["x"]) / np.linalg.norm(r["x"])))
        M.free()
        op.free()

    # ---- CG on a partitioned 27-point operator: every rank's slab takes a wide storage format (rows of 27 entries),
    # interior / boundary launches and the halo windows included
    S = csr_ref.stencil27(64, 12, 9, seed=5)
    ns = S.shape[0]
    rhs_s = S.matvec(np.ones(ns))
    refc = kr.cg(S, rhs_s, reltol=1e-9)
    for mode in ("halo", "allgather"):
        op, ranges = dist.partition_host_csr(world, S.indptr, S.indices, S.data, ns, mode=mode)
        c0, c1 = ranges[rank]
        fmt = ctypes.c_int32()
        _lib.check(_lib.load().mk_csr_format_info(op.handle, ctypes.byref(fmt), None, None, None, None))
        s = CG(op, reltol=1e-9)
        s.solve(rhs_s[c0:c1])
        x = gather_x(world, s.x)
        out["cg27/" + mode] = dict(nMatvec=int(s.nMatvec), ref=int(refc["nMatvec"]),
                                   hist_err=rel_hist_err(s.residHistory, refc["residHistory"]),
                                   x_err=float(np.linalg.norm(x - refc["x"]) / np.linalg.norm(refc["x"])),
                                   fmt=int(fmt.value))
        op.free()

    # ---- MINRES / SYMMLQ on a partitioned 2-D Laplacian
    mm = 43
    C = csr_ref.poisson2d(mm)
    nc = mm * mm
    rhs_c = C.matvec(np.ones(nc))
    refm = kr.minres(C, rhs_c, check=False, etol=0.0, rtol=1e-10)
    refs = kr.symmlq(C, rhs_c)
    for mode in ("halo", "allgather"):
        op, ranges = dist.partition_host_csr(world, C.indptr, C.indices, C.data, nc, mode=mode)
        c0, c1 = ranges[rank]
        s = Minres(op)
        s.solve(rhs_c[c0:c1], show=False, check=False, etol=0.0, rtol=1e-10)
        x = gather_x(world, s.x)
        out["minres/" + mode] = dict(nMatvec=int(s.itn), ref=int(refm["itn"]),
                                     hist_err=rel_hist_err(s.residHistory, refm["residHistory"]),
                                     x_err=float(np.linalg.norm(x - refm["x"]) / np.linalg.norm(refm["x"])))
        s = Symmlq(op)
        s.solve(rhs_c[c0:c1])
        x = gather_x(world, s.x)
        out["symmlq/" + mode] = dict(nMatvec=int(s.nMatvec), ref=int(refs["nMatvec"]), hist_err=0.0,
                                     x_err=float(np.linalg.norm(x - refs["x"]) / np.linalg.norm(refs["x"])))
        op.free()

    # ---- check=True on a partitioned operator: the symmetry test is collective (global random vector sliced per
    # rank, exchange before each product, rank-summed inner products) and all ranks reach the same verdict
    from pykrylov_amd.tools import check_symmetric
    op, ranges = dist.partition_host_csr(world, C.indptr, C.indices, C.data, nc, mode="halo")
    c0, c1 = ranges[rank]
    assert check_symmetric(op) is True
    s = Minres(op)
    s.solve(rhs_c[c0:c1], show=False, check=True, etol=0.0, rtol=1e-10)
    x = gather_x(world, s.x)
    out["minres_check/halo"] = dict(nMatvec=int(s.itn), ref=int(refm["itn"]),
                                    hist_err=rel_hist_err(s.residHistory, refm["residHistory"]),
                                    x_err=float(np.linalg.norm(x - refm["x"]) / np.linalg.norm(refm["x"])))
    op.free()
    op, ranges = dist.partition_host_csr(world, B.indptr, B.indices, B.data, nb, mode="halo")   # nonsymmetric
    c0, c1 = ranges[rank]
    assert check_symmetric(op) is False
    s = Minres(op)
    s.solve(rhs_b[c0:c1], show=False, check=True)
    assert s.istop == 7 and s.itn == 0, (s.istop, s.itn)                  # minres.py:186-190
    op.free()

    # ---- default iteration limit on an UNEVEN split (n = 1001 over 2 or 3 ranks: local blocks differ): the limit is
    # 2 * n_global on every rank (cg.py:79-80), so all ranks take the same number of passes and the collectives stay
    # matched (with per-rank limits the ranks with the shorter block would stop enqueuing first and the job hangs)
    D = csr_ref.poisson1d(1001)
    rhs_d = D.matvec(np.sin(np.arange(1001.0)))
    op, ranges = dist.partition_host_csr(world, D.indptr, D.indices, D.data, 1001, mode="halo")
    c0, c1 = ranges[rank]
    s = CG(op, abstol=0.0, reltol=1e-300)                   # unreachable tolerance: runs until the limit (or r == 0)
    s.solve(rhs_d[c0:c1], check_curvature=False)
    counts = world.allgather_object(int(s.nMatvec))
    assert len(set(counts)) == 1 and 600 < counts[0] <= 2 * 1001, counts
    op.free()

    # ---- least-squares solvers on row blocks (mk_csr_set_row_block): m-space vectors sliced, n-space vectors whole
    # on every rank, A' u summed over the ranks.  Compared with the reference's own golden runs of the 2000 x 1500
    # problem (the same tolerances as the single-GPU test against them, tests/test_gpu_lls.py).
    from pykrylov_amd import lls
    g = np.load(os.path.join(ROOT, "tests", "golden", "lls_random.npz"), allow_pickle=False)
    L = csr_ref.RefCsr(g["l_A_indptr"], g["l_A_indices"], g["l_A_data"], g["l_A_shape"])
    mL, nL = L.shape
    cases = [(s_, b_, k_, False) for s_, b_, k_ in (("lsqr", "ls", dict(damp=0.0)), ("lsqr", "cons", dict(damp=0.1)),
                                                     ("lsmr", "ls", dict(damp=0.0)), ("craig", "cons", {}),
                                                     ("craigmr", "cons", {}))]
    # ... and with the n-space vectors sliced over the ranks as well (reduce-scatter of A' u, all-gather of v)
    cases += [(s_, b_, k_, True) for s_, b_, k_, _ in cases]
    for solver, btag, kw, sliced in cases:
        op, ranges = dist.partition_row_blocks(world, L.indptr, L.indices, L.data, L.shape, sliced=sliced)
        r0, r1 = ranges[rank]
        assert op.shape == (r1 - r0, nL) and op.global_shape == (mL, nL)
        b = g["l_b_" + btag]
        key = "l_%s_%s_d%g_e%g_" % (solver, btag, kw.get("damp", 0.0), 1e-6)
        cls = dict(lsqr=lls.LSQRFramework, lsmr=lls.LSMRFramework, craig=lls.CRAIGFramework,
                   craigmr=lls.CRAIGMRFramework)[solver]
        s = cls(op)
        ret = s.solve(b[r0:r1], etol=1e-6, **kw)
        if solver == "lsmr":                                # (returns its results, lsmr.py:492)
            s.istop, s.itn = ret[1], ret[2]
        x = np.asarray(s.x)
        if solver == "craigmr":                             # x has nargout entries (craigmr.py:112): sliced like rows
            x = gather_x(world, x)
        else:                                               # whole on every rank, and the same bits everywhere
            xs = world.allgather_object(x)
            assert all(np.array_equal(xs[0], xr) for xr in xs), solver
        xref = g[key + "x"]
        itn = int(s.itn)
        itns = world.allgather_object(itn)
        assert len(set(itns)) == 1, itns
        ent = dict(itn=itn, ref=int(g[key + "itn"]), istop=int(s.istop),
                   x_err=float(np.linalg.norm(x - xref) / np.linalg.norm(xref)))
        if solver == "craig":
            r = gather_x(world, s.r)                        # (the golden run kept no r: the oracle's)
            rref = lls_ref.craig(L.matvec, L.transpose().matvec, L.shape, b.copy(), etol=1e-6)["r"]
            ent["r_err"] = float(np.linalg.norm(r - rref) / np.linalg.norm(b))
        out["lls_%s_%s%s" % (solver, btag, "_sliced" if sliced else "")] = ent
        op.free()
    # diagonal preconditioners: M sliced like the rows, N whole
    dm = 1.0 + 0.5 * np.cos(np.arange(mL))
    dn = 1.0 + 0.25 * np.sin(np.arange(nL))
    Lt = L.transpose()
    ref = lls_ref.lsqr(L.matvec, Lt.matvec, L.shape, g["l_b_ls"].copy(), etol=0.0, M=lambda v: dm * v, N=lambda v: dn * v)
    for sliced in (False, True):
        op, ranges = dist.partition_row_blocks(world, L.indptr, L.indices, L.data, L.shape, sliced=sliced)
        r0, r1 = ranges[rank]
        s = lls.LSQRFramework(op)
        s.solve(g["l_b_ls"][r0:r1], M=DiagonalOperator(dm[r0:r1]), N=DiagonalOperator(dn), etol=0.0)
        xs = world.allgather_object(np.asarray(s.x))
        assert all(np.array_equal(xs[0], xr) for xr in xs)
        out["lls_lsqr_precon" + ("_sliced" if sliced else "")] = dict(
            itn=int(s.itn), ref=int(ref["itn"]), istop=int(s.istop),
            x_err=float(np.linalg.norm(s.x - ref["x"]) / np.linalg.norm(ref["x"])))
        op.free()

    _lib.load().mk_comm_destroy()
    if rank == 0:
        print("RESULT " + json.dumps(out))
    td.barrier()
    td.destroy_process_group()


if __name__ == "__main__":
    main()
