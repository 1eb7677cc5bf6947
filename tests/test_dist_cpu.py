"""world_size-2 (and 3) CPU tests of the multi-GPU path's host logic: partition plans of
pykrylov_amd.dist, checked by running the oracle's CG row-partitioned over torch.distributed/gloo."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("nranks", [2, 3])
def test_partitioned_cg_over_gloo(nranks):
    env = dict(os.environ, OPENBLAS_NUM_THREADS="1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "tests", "_dist_worker.py")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    out = json.loads(line[len("RESULT "):])
    assert len(out) >= 4
    for key, r in out.items():
        assert r["nMatvec"] == r["ref"], (key, r)
        assert r["hist_err"] <= 1e-12 and r["x_err"] <= 1e-12, (key, r)


def test_planning_single_rank():
    import numpy as np
    from pykrylov_amd import dist
    from oracle import csr_ref
    A = csr_ref.poisson2d(9)
    w = dist.World()
    p = dist.plan_host_csr(w, A.indptr, A.indices, A.data, 81, mode="halo")
    assert p["n_halo"] == 0 and p["n_local"] == 81 and np.array_equal(p["indices"], A.indices)
    q = dist.plan_host_csr(w, A.indptr, A.indices, A.data, 81, mode="allgather")
    assert q["n_halo"] == 81 and np.array_equal(q["indices"], A.indices + 81)
    assert dist.equal_ranges(10, 4) == ([(0, 3), (3, 6), (6, 9), (9, 10)], 3)      # last block short: padded on the wire
    with pytest.raises(NotImplementedError):
        dist.equal_ranges(5, 4)                                                   # would leave a rank without rows
    need = dist.banded_needs(10, 20, 3, 2)
    assert need.tolist() == [7, 8, 9, 20, 21]
    sc, rc, si = dist.halo_plan(1, [(0, 10), (10, 20), (20, 30)], [np.array([10, 11]), need, np.array([19])])
    assert sc.tolist() == [2, 0, 1] and rc.tolist() == [3, 0, 2] and si.tolist() == [0, 1, 9]
