#!/usr/bin/env python3
"""bench.py -- Krylov iterations/s of the device-resident CG + roofline of its SpMV kernel.

    python bench.py --gpus 1 --steps K --warmup W            # configs[4] on one GPU (+ configs[1] as "extra")
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W   # configs[4]: 512^3 on N GPUs

A "step" is one pass of the CG loop body (reference pykrylov/cg/cg.py:113-158: 1 SpMV, 2 dots,
3 vector updates) on synthetic data resident in HBM.  Tolerances are set to zero so that exactly
K passes run inside the timed region.  Rank 0 prints ONE JSON line.

Workloads (BASELINE.json `configs`):
  poisson3d-512    configs[4]  CG, 3-D 7-point Poisson 512^3 (1.34e8 rows, 9.4e8 nnz) row-partitioned over N GPUs --
                               the configuration BASELINE.json's target is quoted on.  It fits one GPU (14 GB of
                               288 GB), so it is the default at EVERY N: the 1/2/4/8 series is one strong-scaling
                               series over a fixed problem.
  poisson2d-1000   configs[1]  CG, 2-D 5-point Poisson, n = 1e6, one GPU; also run at N = 1 and reported under
                               "extra" (value, ms_per_step and its own SpMV roofline), or alone with --workload.
"""
import argparse
import ctypes
import json
import os
import sys
import time

# the CPU baseline is a one-core number: keep OpenBLAS (np.dot in the oracle) from fanning out over the host
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s per GPU


def spmv_bytes(nrows, ncols, nnz):
    """Algorithmic bytes of one CSR SpMV launch: 12 nnz + 4 (n+1) + 8 ncols + 8 nrows."""
    return 12 * nnz + 4 * (nrows + 1) + 8 * ncols + 8 * nrows


def build_workload(name, world):
    from pykrylov_amd import gallery, dist
    if name.startswith("poisson2d-"):
        m = int(name.split("-")[1])
        n = m * m
        if world.nranks == 1:
            return gallery.poisson2d(m), n, {"grid": [m, m], "stencil": 5}
        indptr, indices, data, _ = gallery.poisson2d_csr(m)
        op, _ = dist.partition_host_csr(world, indptr, indices, data, n, mode=ARGS.exchange)
        return op, n, {"grid": [m, m], "stencil": 5}
    if name.startswith("poisson3d-"):
        m = int(name.split("-")[1])
        n = m ** 3
        if world.nranks == 1:
            return gallery.poisson3d(m), n, {"grid": [m, m, m], "stencil": 7}
        op, _ = dist.partition_poisson3d(world, m, m, m, mode=ARGS.exchange)
        return op, n, {"grid": [m, m, m], "stencil": 7}
    raise SystemExit("unknown workload %r" % name)


def cpu_baseline(name, seconds_budget=20.0):
    """The CPU oracle (NumPy restatement of the reference loop + C CSR product, one core) timed on a
    bounded sample of the same workload."""
    from oracle import csr_ref, krylov_ref
    m = int(name.split("-")[1])
    if name.startswith("poisson2d-"):
        A = csr_ref.poisson2d(m)
        scale, sample = 1.0, "first %%d CG iterations of %s (n=%d), rhs=A*1" % (name, m * m)
    else:
        ms = min(m, 128)                 # 512^3 does not fit a host-side sample: time 128^3 and scale by rows
        A = csr_ref.poisson3d(ms)
        scale = float(ms ** 3) / float(m ** 3)
        sample = ("first %%d CG iterations on %d^3 (%d rows), iterations/s scaled by rows ratio %.4g to %s"
                  % (ms, ms ** 3, scale, name))
    n = A.shape[0]
    rhs = A.matvec(np.ones(n))
    t0 = time.perf_counter()
    krylov_ref.cg(A, rhs, abstol=0.0, reltol=0.0, matvec_max=10)
    per_it = (time.perf_counter() - t0) / 10.0
    iters = int(max(20, min(2000, seconds_budget / max(per_it, 1e-6))))
    t0 = time.perf_counter()
    out = krylov_ref.cg(A, rhs, abstol=0.0, reltol=0.0, matvec_max=iters)
    dt = time.perf_counter() - t0
    return {"value": out["nMatvec"] / dt * scale, "unit": "iterations/s", "cores": 1, "kind": "port",
            "sample": sample % out["nMatvec"], "host_cpus": os.cpu_count(),
            "blas_threads": os.environ.get("OPENBLAS_NUM_THREADS", "default")}


def other_configs(lib, passes=400, warm=20):
    """BASELINE configs[2] and configs[3] on one GPU (`--all-configs`): loop passes per second with the
    tolerances at zero so that exactly `passes` passes run, plus the iteration roofline with the
    reference's op count."""
    from pykrylov_amd import _lib, gallery
    from pykrylov_amd.generic import DeviceRun
    out = {}

    def timed(op, kind, rhs, **params):
        run = DeviceRun(op, kind, rhs, None, **params)
        run.setup()
        assert run.iterate(warm) == warm
        _lib.check(lib.mk_sync())
        t0 = time.perf_counter()
        done = run.iterate(passes)
        _lib.check(lib.mk_sync())
        dt = time.perf_counter() - t0
        assert done == passes, (done, passes)
        run.close()
        return dt

    # configs[2]: BiCGSTAB, random nonsymmetric diagonally dominant CSR, n = 1e6, ~5 nnz/row.  With threshold 0
    # the residual reaches 0 after ~25 passes and the recurrence then divides 0 by 0 (as the reference would):
    # the kernels move the same bytes on NaNs, which is what is being timed.
    n = 1000000
    op = gallery.random_diagdom(n, seed=1)
    ones = _lib.DeviceArray.from_numpy(np.ones(n))
    rhs = _lib.DeviceArray(n)
    op.spmv_device(ones.ptr, rhs.ptr)
    dt = timed(op, _lib.MK_BICGSTAB, rhs, abstol=0.0, reltol=0.0, matvec_max=1 << 60)
    b_iter = 2 * spmv_bytes(n, n, op.nnz) + 224 * n
    out["bicgstab-rand1m@1"] = {"value": passes / dt, "unit": "iterations/s", "ms_per_step": 1e3 * dt / passes,
                                "rows": n, "nnz": int(op.nnz), "matvecs_per_iteration": 2,
                                "iteration_roofline": {"algorithmic_bytes_per_iter": b_iter,
                                                       "frac_of_hbm": b_iter * passes / dt / 1e9 / HBM_PEAK_GBS}}
    op.free()
    # configs[3]: MINRES, 2-D Laplacian m = 2000 (n = 4e6) with the keyword shift 1.5 (symmetric indefinite)
    m = 2000
    n = m * m
    op = gallery.poisson2d(m)
    ones = _lib.DeviceArray.from_numpy(np.ones(n))
    rhs_h = np.empty(n)
    rhs = _lib.DeviceArray(n)
    op.spmv_device(ones.ptr, rhs.ptr)
    rhs_h[:] = rhs.to_numpy() - 1.5
    dt = timed(op, _lib.MK_MINRES, rhs_h, shift=1.5, itnlim=1 << 60, rtol=0.0, etol=0.0, window=5)
    b_iter = spmv_bytes(n, n, op.nnz) + 176 * n                            # SURVEY.md 8d (kwarg shift)
    out["minres-shifted2d-2000@1"] = {"value": passes / dt, "unit": "iterations/s", "ms_per_step": 1e3 * dt / passes,
                                      "rows": n, "nnz": int(op.nnz), "shift": 1.5,
                                      "iteration_roofline": {"algorithmic_bytes_per_iter": b_iter,
                                                             "frac_of_hbm": b_iter * passes / dt / 1e9 / HBM_PEAK_GBS}}
    op.free()
    return out


def main():
    global ARGS
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="auto")
    ap.add_argument("--exchange", default="halo", choices=["halo", "allgather"])
    ap.add_argument("--event-stride", type=int, default=0,
                    help="additionally bracket the SpMV of every k-th pass with its own HIP event pair "
                         "(intrusive: each pair costs ~3-6 us; 0 = off)")
    ap.add_argument("--spmv-launches", type=int, default=200,
                    help="back-to-back launches of the fused SpMV kernel timed by one HIP event pair")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "host"],
                    help="host: collectives staged through host memory over gloo, all ranks on GPU 0 -- a smoke test "
                         "of the N > 1 path on a single-GPU box, not a measurement")
    ap.add_argument("--all-configs", action="store_true",
                    help="also time BASELINE configs[2] (BiCGSTAB, random n=1e6) and configs[3] (MINRES, n=4e6)")
    ARGS = ap.parse_args()

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != ARGS.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (ARGS.gpus, world_size, ARGS.gpus))
    td = torch = None
    if world_size > 1:
        import torch
        import torch.distributed as td
        if ARGS.transport == "host":
            local_rank = 0
            td.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            td.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from pykrylov_amd import _lib, dist
    from pykrylov_amd.generic import DeviceRun
    lib = _lib.init(local_rank)
    world = dist.World(rank, world_size, td)
    if world_size > 1:
        world.init_device_comm(transport=ARGS.transport)

    name = ARGS.workload
    if name == "auto":
        name = "poisson3d-512"

    def barrier():
        if ARGS.transport == "rccl":
            td.barrier(device_ids=[local_rank])
        else:
            td.barrier()

    def run_cg(workload, steps, warmup, stride):
        op, n_global, meta = build_workload(workload, world)
        n_local = getattr(op, "local_size", None) or op.shape[1]
        ones = _lib.DeviceArray.from_numpy(np.ones(op.shape[1]))
        rhs = _lib.DeviceArray(n_local)
        op.spmv_device(ones.ptr, rhs.ptr)                     # rhs = A * 1 (test_diagdom.py:78-79 convention)
        # tolerances 0: never converges, so exactly `steps` passes run inside the timed region
        run = DeviceRun(op, _lib.MK_CG, rhs, None, abstol=0.0, reltol=0.0, matvec_max=1 << 60,
                        check_curvature=1, spmv_event_stride=stride)
        run.setup()
        done_w = run.iterate(warmup)
        _lib.check(lib.mk_sync())
        if td is not None:
            torch.cuda.synchronize()
            barrier()
        t0 = time.perf_counter()
        done = run.iterate(steps)
        _lib.check(lib.mk_sync())
        if td is not None:
            torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if td is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if ARGS.transport == "rccl" else "cpu")
            td.all_reduce(t, op=td.ReduceOp.MAX)
            elapsed = float(t.item())
            barrier()
        timing = run.timing()
        res = run.finish()
        # dominant kernel: the loop's fused SpMV, launched back to back with one HIP-event pair around the
        # whole train (on the solver's stream), after the timed region so that it does not disturb `value`
        timing["spmv_b2b_us"] = None
        if stride >= 0 and ARGS.spmv_launches > 0:
            avg = ctypes.c_double()
            _lib.check(lib.mk_solver_time_spmv(run.handle, ARGS.spmv_launches, ctypes.byref(avg)))
            timing["spmv_b2b_us"] = avg.value
        assert done == steps and done_w == warmup, (done, steps, done_w, warmup)
        assert np.isfinite(res.residNorm), "CG diverged"
        hist = run.history()
        info = dict(op_shape=op.shape, nnz=op.nnz, n_local=n_local, n_global=n_global, meta=meta, elapsed=elapsed,
                    timing=timing, resid_first=float(hist[0]), resid_last=float(hist[-1]))
        run.close()
        op.free()
        return info

    def roofline_of(info, workload, steps):
        n_g, n_l = info["n_global"], info["n_local"]
        b_spmv = spmv_bytes(n_l, info["op_shape"][1], info["nnz"])
        tm = info["timing"]
        spmv_us = tm["spmv_b2b_us"]
        achieved = b_spmv / (spmv_us * 1e-6) / 1e9 if spmv_us else None
        inloop_us = 1e3 * tm["spmv_ms"] / tm["spmv_launches"] if tm["spmv_launches"] else None
        # whole-iteration roofline with the reference's op count: B_spmv + 104 n per pass
        stencil = info["meta"]["stencil"]
        nnz_global = stencil * n_g - 2 * sum(n_g // g for g in info["meta"]["grid"])
        iter_bytes = spmv_bytes(n_g, n_g, nnz_global) + 104 * n_g
        its = steps / info["elapsed"]
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "spmv_traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get("%s@%d" % (workload, world_size))
        roof = {"bound": "hbm", "kernel": "mk_spmv_kernel<CgSpmvEpi> (CSR-stream SpMV + fused <p,Ap>)",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                "bytes_per_launch": b_spmv, "avg_launch_us": spmv_us, "launches_timed": ARGS.spmv_launches,
                "method": "one hipEvent pair around back-to-back launches on the solver stream",
                "inloop_event_pair_us": inloop_us}
        it_roof = {"algorithmic_bytes_per_iter": iter_bytes, "achieved_GBs": iter_bytes * its / 1e9,
                   "frac_of_aggregate_hbm": iter_bytes * its / 1e9 / (HBM_PEAK_GBS * world_size)}
        return its, nnz_global, roof, it_roof

    info = run_cg(name, ARGS.steps, ARGS.warmup, ARGS.event_stride)
    elapsed = info["elapsed"]
    tm = info["timing"]
    n_g = info["n_global"]
    its, nnz_global, roof, it_roof = roofline_of(info, name, ARGS.steps)

    line = {
        "metric": "Krylov iters/sec + SpMV achieved HBM GB/s (% of peak), fp64",
        "value": its, "unit": "iterations/s", "n_gpus": world_size, "steps": ARGS.steps, "warmup": ARGS.warmup,
        "ms_per_step": 1e3 * elapsed / ARGS.steps, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "CG %s (%d rows, %d nnz), rhs=A*1, x0=0, tolerances 0" % (name, n_g, nnz_global),
                   "baseline_config": ("configs[4]: CG on 3-D 7-point Poisson 512^3, the configuration the target is "
                                       "quoted on (fits one GPU: same problem at every N, strong scaling); configs[1] "
                                       "is reported under extra") if name == "poisson3d-512" else name,
                   "solver": "cg", "rows": n_g, "nnz": nnz_global,
                   "parallelism": "1 GPU" if world_size == 1 else "row-partition x%d, %s exchange + allreduce(dots), %s"
                                  % (world_size, ARGS.exchange, "RCCL" if ARGS.transport == "rccl" else "host-staged gloo (smoke test)")},
        "roofline": roof,
        "iteration_roofline": it_roof,
        "device_loop_ms": tm["iterate_ms"],
        "residual": {"first": info["resid_first"], "last": info["resid_last"]},
    }
    if rank == 0 and world_size == 1 and not ARGS.no_cpu:
        line["cpu_baseline"] = cpu_baseline(name)
    if world_size == 1 and name == "poisson3d-512" and not ARGS.no_extra:
        # BASELINE configs[1] (CG, 2-D Poisson n = 1e6, one GPU): same measurement, reported beside the headline
        ex = run_cg("poisson2d-1000", 2000, 200, 0)
        e_its, e_nnz, e_roof, e_it = roofline_of(ex, "poisson2d-1000", 2000)
        line["extra"] = {"poisson2d-1000@1": {"value": e_its, "unit": "iterations/s", "steps": 2000, "warmup": 200,
                                              "ms_per_step": 1e3 * ex["elapsed"] / 2000, "roofline": e_roof,
                                              "iteration_roofline": e_it}}
    if world_size == 1 and ARGS.all_configs:
        line.setdefault("extra", {}).update(other_configs(lib))
    if rank == 0:
        print(json.dumps(line), flush=True)
    if td is not None:
        lib.mk_comm_destroy()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
