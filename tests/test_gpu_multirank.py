"""The multi-rank device path on a single-GPU box: 2 and 3 processes share GPU 0 (RCCL refuses that, so the
collectives travel through the host-staged transport, `mk_comm_init_host` + torch.distributed/gloo) and run the
row-partitioned solvers exactly as `bench.py --gpus N` does -- per-rank matrix generation / partition plans, column
localisation, pack kernel, [own | halo] vectors, all-reduced partial sums, identical halting on every rank.
Results are compared with the single-process CPU oracle on the unpartitioned problem."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("nranks", [2, pytest.param(3, marks=pytest.mark.slow)])
def test_partitioned_solvers_on_one_gpu(nranks):
    env = dict(os.environ, OPENBLAS_NUM_THREADS="1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "_gpu_rank_worker.py")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-5000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    out = json.loads(line[len("RESULT "):])
    assert len(out) >= 10, sorted(out)
    # rank-local block-Jacobi preconditioners applied on the device: all six square solvers' hook
    assert sum(k.split("/")[0].endswith("_bjacobi") for k in out) == 10, sorted(out)
    lls_keys = [k for k in out if k.startswith("lls_")]
    assert len(lls_keys) == 12 and sum(k.endswith("_sliced") for k in lls_keys) == 6, lls_keys
    for k in lls_keys:
        # (the tolerances of the single-GPU runs against the same golden traces, tests/test_gpu_lls.py)
        r = out.pop(k)
        assert abs(r["itn"] - r["ref"]) <= 1 and r["istop"] > 0 and r["x_err"] <= 1e-5, (k, r)
        assert r.get("r_err", 0.0) <= 1e-5, (k, r)
    for key, r in out.items():
        # partitioning changes the summation order of the dots (per-rank partials), nothing else: same counts on
        # these well-conditioned problems, 1e-12 on histories and iterates
        if key.split("/")[0] in ("bicgstab", "cgs", "tfqmr", "bicgstab_precon", "bicgstab_bjacobi", "cgs_bjacobi",
                                 "tfqmr_bjacobi"):
            # these stop on a tolerance after ~20 passes (reltol 1e-8): the count may move by one product with the
            # summation order, and with it the last update of x
            assert abs(r["nMatvec"] - r["ref"]) <= 1 and r["x_err"] <= 1e-7, (key, r)
            continue
        assert r["nMatvec"] == r["ref"], (key, r)
        assert r["hist_err"] <= 1e-12 and r["x_err"] <= 1e-11, (key, r)
    assert [out["cg3d_march%d/halo" % f]["fmt"] for f in (9, 10, 11)] == [9, 10, 11]          # the slabs keep the march
    assert out["cg3d/halo"]["halo"] in (576, 1152)              # one or two neighbour planes of 24 x 24
    assert out["cg27/halo"]["fmt"] >= 6 and out["cg27/allgather"]["fmt"] >= 6       # the slabs keep a wide format


@pytest.mark.parametrize("launcher", ["self", "torchrun", "rccl-fallback", "self-varcoef", "self-varcoef-8"])
def test_bench_two_rank_path_smoke(launcher):
    """bench.py's N > 1 branch (per-rank matrix generation, gloo bootstrap, barrier / max-over-ranks timing, both
    exchange modes, comm timings, JSON line) with two ranks on GPU 0 over the host-staged transport: a smoke test of
    the code path the driver runs on 2-8 GPUs.  `self`: the plain `python bench.py --gpus 2 ...` form spawns its
    ranks itself; `torchrun`: the external-launcher form of the driver contract."""
    env = dict(os.environ, OPENBLAS_NUM_THREADS="1", OMP_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    args = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "3", "--transport", "host",
            "--workload", "poisson3d-64", "--spmv-launches", "3"]
    nranks = 2
    if launcher.startswith("self-varcoef"):                 # the default line's operator family, partitioned
        args[args.index("poisson3d-64")] = "poisson3d-64-varcoef"
        if launcher.endswith("-8"):                         # ... over EIGHT ranks, the split the driver's scaling run ends on
            nranks = 8
            args[args.index("--gpus") + 1] = "8"
        launcher = "self"
    if launcher == "rccl-fallback":
        # RCCL requested with both ranks on GPU 0: ncclCommInitRank refuses, every rank falls back to the host-staged
        # transport and the line says so (a broken RCCL setup on a multi-GPU box must not cost the whole line)
        args[args.index("--transport"):args.index("--transport") + 2] = ["--one-device"]
    if launcher != "torchrun":
        cmd = [sys.executable] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port())] + args
    import tempfile
    dpath = os.path.join(tempfile.mkdtemp(prefix="mk_bench_"), "detail.json")
    env["BENCH_DETAIL"] = dpath
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    text = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    # the line the driver has to parse out of an 8 KB stdout tail: compact, for 2 AND 8 ranks
    assert len(text) <= 4096, len(text)
    line = json.loads(text)
    detail = json.load(open(dpath))
    assert detail["line"] == line
    assert line["n_gpus"] == nranks and line["steps"] == 12 and line["value"] > 0 and line["scaling"] == "strong"
    assert line["config"]["rows"] == 64 ** 3 and line["residual"]["recurrence"] > 0
    assert line["roofline"]["bound"] == "hbm"
    # (N > 1: the dominant kernel on rank 0's slab, timed alone after the run -- never null in a line the driver records)
    assert line["roofline"]["achieved"] > 0 and 0 < line["roofline"]["frac"] < 1 and line["roofline"]["avg_launch_us"] > 0
    # the N > 1 line validates itself: the first 60 passes against the committed single-GPU device history (1e-12), and
    # the true residual ||b - A x_k|| (exchange + product + all-reduced norm) against the recurrence's after the timed region
    par = line["parity_vs_n1"]
    assert par["ok"] and par["passes"] == 60 and par["rel_hist_err"] <= 1e-12, par
    assert detail["workloads"][detail["headline"]]["parity_vs_n1"]["fixture_has_workload"]
    assert line["residual"]["ok"] and line["residual"]["rel_gap"] <= 1e-10, line["residual"]
    res = detail["workloads"][detail["headline"]]["residual"]
    assert res["last"] < res["first"]
    ex = line["exchange"]
    assert set(ex) == {"halo", "allgather"} and ex["allgather"]["value"] > 0
    assert ex["halo"]["max_over_ranks"]["product_alone_us"] > 0 and ex["halo"]["max_over_ranks"]["exchange_alone_us"] > 0
    dex = detail["exchange"]                                # (per-rank timings live in the detail file)
    assert dex["halo"]["comm"]["per_rank"][0]["product_alone_us"] > 0
    assert len(dex["halo"]["comm"]["per_rank"]) == nranks and dex["halo"]["comm"]["per_rank"][1]["exchange_alone_us"] > 0
    assert "host-staged gloo" in line["config"]["parallelism"]
    # the line says what carried the collectives: a host-staged fallback can never pass for an RCCL measurement
    assert line["transport"] == {"kind": "host-staged", "rccl_ranks_seen": 0, "halo_communicator_split": False}
    assert ("RCCL communicator unavailable" in line["config"]["parallelism"]) == (launcher == "rccl-fallback")
