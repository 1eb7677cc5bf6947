"""Worker of tests/test_dist_cpu.py (launched with torch.distributed.run, backend gloo, CPU only)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch.distributed as td

from oracle import csr_ref, dist_ref, krylov_ref
from pykrylov_amd import dist


def main():
    td.init_process_group(backend="gloo")
    rank, nranks = td.get_rank(), td.get_world_size()
    world = dist.World(rank, nranks, td)
    out = {}
    cases = {"poisson2d": csr_ref.poisson2d(14), "poisson3d": csr_ref.poisson3d(6, 5, 8),
             "random": None}
    rnd = csr_ref.random_diagdom(300, seed=3)
    # symmetrise the random matrix so that CG applies: A + A^T + diagonal shift keeps it SPD-ish
    D = rnd.to_dense()
    S = D + D.T + 10.0 * np.eye(300)
    r, c = np.nonzero(S)
    cases["random"] = csr_ref.from_coo(r, c, S[r, c], S.shape)
    for name, A in cases.items():
        n = A.shape[0]
        rhs = A.matvec(np.ones(n))
        ref = krylov_ref.cg(A, rhs)
        for mode in ("halo", "allgather"):
            p = dist.plan_host_csr(world, A.indptr, A.indices, A.data, n, mode=mode)
            c0, c1 = p["ranges"][rank]
            # integer parts of the plan: every local column id points at the right global entry
            full = np.arange(n, dtype=np.float64) * 3.0 + 1.0
            ext = dist_ref.exchange(p, full[c0:c1], rank, nranks)
            sl = slice(A.indptr[c0], A.indptr[c1])
            assert np.array_equal(ext[p["indices"]], full[A.indices[sl]]), (name, mode)
            assert np.array_equal(p["indptr"], A.indptr[c0:c1 + 1] - A.indptr[c0])
            if mode == "halo":
                assert int(np.sum(p["recv_count"])) == p["n_halo"] and p["recv_count"][rank] == 0
                assert np.all(np.diff(p["halo_cols"]) > 0)
            res = dist_ref.dist_cg(p, rhs[c0:c1], rank, nranks, n_global=n)
            err = np.max(np.abs(res["residHistory"] - ref["residHistory"][:len(res["residHistory"])])
                         / np.maximum(ref["residHistory"][:len(res["residHistory"])], 1e-4 * ref["residHistory"][0]))
            out["%s/%s" % (name, mode)] = dict(nMatvec=int(res["nMatvec"]), ref=int(ref["nMatvec"]), hist_err=float(err),
                                              x_err=float(np.max(np.abs(res["x"] - ref["x"][c0:c1]))))
    # planning helpers
    assert dist.row_ranges(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert dist.row_ranges(8 * 6, 4, align=6) == [(0, 12), (12, 24), (24, 36), (36, 48)]
    assert dist.row_ranges(5 * 6, 4, align=6)[-1] == (24, 30)
    ob = world.allgather_object({"rank": rank})
    assert [o["rank"] for o in ob] == list(range(nranks))
    if rank == 0:
        print("RESULT " + json.dumps(out))
    td.barrier()
    td.destroy_process_group()


if __name__ == "__main__":
    main()
