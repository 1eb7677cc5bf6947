"""CPU oracle of the row-partitioned execution: NumPy vectors + torch.distributed (gloo).

TEST INFRASTRUCTURE ONLY.  Mirrors, collective for collective, what libmikrylov does on N GPUs
(csrc/mk_comm.hip): before every product the input vector is completed by the plan's exchange (halo
send/recv in rank order, or an all-gather), every inner product is a local sum followed by a sum
all-reduce.  Used by the world_size-2 CPU tests to check the plans of pykrylov_amd.dist.
"""
import numpy as np
import torch
import torch.distributed as td

from . import csr_ref, krylov_ref


def exchange(plan, x_local, rank, nranks):
    """Return [x_local | received entries] according to `plan` (dict of pykrylov_amd.dist.plan_host_csr)."""
    n_local, n_halo = plan["n_local"], plan["n_halo"]
    ext = np.zeros(n_local + n_halo)
    ext[:n_local] = x_local
    if plan["mode"] == 1:
        cnt = n_halo // nranks
        mine = torch.zeros(cnt, dtype=torch.float64)
        mine[:n_local] = torch.from_numpy(np.ascontiguousarray(x_local))
        parts = [torch.zeros(cnt, dtype=torch.float64) for _ in range(nranks)]
        td.all_gather(parts, mine)
        ext[n_local:] = torch.cat(parts).numpy()
        return ext
    send_off = np.concatenate([[0], np.cumsum(plan["send_count"])])
    recv_off = np.concatenate([[0], np.cumsum(plan["recv_count"])])
    reqs, bufs = [], {}
    for r in range(nranks):
        if plan["send_count"][r]:
            idx = plan["send_idx"][send_off[r]:send_off[r + 1]]
            reqs.append(td.isend(torch.from_numpy(np.ascontiguousarray(x_local[idx])), dst=r))
        if plan["recv_count"][r]:
            bufs[r] = torch.zeros(int(plan["recv_count"][r]), dtype=torch.float64)
            reqs.append(td.irecv(bufs[r], src=r))
    for q in reqs:
        q.wait()
    for r, b in bufs.items():
        ext[n_local + recv_off[r]:n_local + recv_off[r + 1]] = b.numpy()
    return ext


def dist_dot(a, b, site=""):
    t = torch.tensor([float(np.dot(a, b))], dtype=torch.float64)
    td.all_reduce(t)
    return float(t[0])


def dist_cg(plan, rhs_local, rank, nranks, **kw):
    """The oracle's CG on this rank's row block; returns the oracle result dict with the local x."""
    A = csr_ref.RefCsr(plan["indptr"], plan["indices"], plan["data"],
                       (plan["n_local"], plan["n_local"] + plan["n_halo"]))

    def matvec(v_local):
        return A.matvec(exchange(plan, v_local, rank, nranks))
    n_global = kw.pop("n_global")
    kw.setdefault("matvec_max", 2 * n_global)
    return krylov_ref.cg(matvec, rhs_local, red=krylov_ref.Reductions(dist_dot), **kw)
