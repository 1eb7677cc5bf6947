#!/usr/bin/env python3
"""bench.py -- Krylov iterations/s of the device-resident CG + roofline of its SpMV kernel.

    python bench.py --gpus N --steps K --warmup W           # N = 1: in-process; N > 1: spawns N ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W   # same thing under an external launcher

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

Multi-GPU: one process per GPU.  torch.distributed is used with the gloo backend ONLY, for the bootstrap
(RCCL unique id, barriers, max over ranks of the elapsed time): the only RCCL instance in a process is the one
libmikrylov dlopens for the data path (halo send/recv or all-gather before each product, all-reduce of the dot
partials).  For N > 1 both exchange modes are measured back to back (`exchange`), `value` is the halo one.
"""
import argparse
import ctypes
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

# the CPU baseline's headline figure is a one-core number: keep OpenBLAS (np.dot in the oracle) from fanning out
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("OMP_NUM_THREADS", "1")           # (the oracle's C product is OpenMP-parallel over rows)

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s per GPU
METRIC = "Krylov iters/sec + SpMV achieved HBM GB/s (% of peak), fp64"
FMT_NAMES = {0: "csr (int32 columns + fp64 values, x gathered)",
             1: "windowed tiles (x windows in LDS, uint16 slots + fp64 values)",
             2: "windowed tiles + value dictionary (one packed 32-bit word per nonzero: LDS slot + value code)",
             3: "csr, tile resident in LDS, gathers ordered by column block (x longer than an L2)",
             4: "windowed tiles + value dictionary + row patterns (one byte per ROW: the number of its pattern of "
                "{LDS slot - lane, value code} words)"}


def spmv_bytes(nrows, ncols, nnz):
    """Algorithmic bytes of one CSR SpMV launch: 12 nnz + 4 (n+1) + 8 ncols + 8 nrows."""
    return 12 * nnz + 4 * (nrows + 1) + 8 * ncols + 8 * nrows


def kernel_source_sha():
    """Fingerprint of the SpMV kernel sources: PMC traffic figures under profiles/ are only quoted while it matches."""
    h = hashlib.sha256()
    for f in ("mk_device.h", "mk_format.hip", "mk_internal.h"):
        with open(os.path.join(ROOT, "pykrylov_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


# ======================================================================================
# launcher: `python bench.py --gpus N` without an external launcher spawns the N ranks itself
# ======================================================================================
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_spawn(nranks):
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE=str(nranks),
               LOCAL_WORLD_SIZE=str(nranks), HSA_ENABLE_IPC_MODE_LEGACY=env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    for r in range(nranks):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    deadline = time.time() + float(os.environ.get("BENCH_SPAWN_TIMEOUT", "3000"))
    alive = list(procs)
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is not None:
                alive.remove(p)
                if code != 0 and rc == 0:                  # one rank failed: the others would wait for it forever
                    rc = code
                    for q in alive:
                        q.kill()
        if time.time() > deadline:
            for q in alive:
                q.kill()
            rc = rc or 124
        time.sleep(0.05)
    sys.exit(rc)


# ======================================================================================
# CPU baselines (test infrastructure timed as a reported baseline, never the product path)
# ======================================================================================
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
            "blas_threads": os.environ.get("OPENBLAS_NUM_THREADS", "default"),
            "spmv_threads": os.environ.get("OMP_NUM_THREADS", "default")}


def cpu_baseline_all_cores(name, seconds_budget=9.0):
    """The same oracle on many host cores (SURVEY.md 8d asks for both figures): the C CSR product runs OpenMP-parallel
    over the rows (same bits); NumPy's dots and element-wise updates stay on one thread, which bounds the speed-up.
    8, 32 and all visible cores are tried (containers often see more CPUs than their quota lets them use at once) and
    the best is reported with the thread count it used.  Child processes: libgomp reads its thread count when it loads."""
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    tried = sorted({min(ncpu, 8), min(ncpu, 32), ncpu})
    code = ("import json,sys; sys.path.insert(0, %r); import bench; "
            "print(json.dumps(bench.cpu_baseline(%r, %r)))" % (ROOT, name, seconds_budget / len(tried)))
    best, err = None, None
    for threads in tried:
        env = dict(os.environ)
        env["OMP_NUM_THREADS"] = str(threads)
        env["OPENBLAS_NUM_THREADS"] = "1"
        env["BENCH_CHILD"] = "1"
        try:
            out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
            d = json.loads(out.stdout.strip().splitlines()[-1])
            d["cores"] = threads
            if best is None or d["value"] > best["value"]:
                best = d
        except Exception as e:                               # a baseline must never take the bench line down
            err = repr(e)[:200]
    if best is None:
        return {"value": None, "error": err}
    best["threads_tried"] = tried
    return best


# ======================================================================================
# workloads
# ======================================================================================
def build_workload(name, world, exchange):
    from pykrylov_amd import gallery, dist
    if name.startswith("poisson2d-"):
        m = int(name.split("-")[1])
        n = m * m
        if world.nranks == 1:
            return gallery.poisson2d(m), n, {"grid": [m, m], "stencil": 5}
        indptr, indices, data, _ = gallery.poisson2d_csr(m)
        op, _ = dist.partition_host_csr(world, indptr, indices, data, n, mode=exchange)
        return op, n, {"grid": [m, m], "stencil": 5}
    if name.startswith("poisson3d-"):
        m = int(name.split("-")[1])
        n = m ** 3
        if world.nranks == 1:
            return gallery.poisson3d(m), n, {"grid": [m, m, m], "stencil": 7}
        op, _ = dist.partition_poisson3d(world, m, m, m, mode=exchange)
        return op, n, {"grid": [m, m, m], "stencil": 7}
    raise SystemExit("unknown workload %r" % name)


def format_info(lib, op):
    from pykrylov_amd import _lib
    fmt, chunks, nd = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    tiles, mbytes = ctypes.c_int64(), ctypes.c_int64()
    _lib.check(lib.mk_csr_format_info(op.handle, ctypes.byref(fmt), ctypes.byref(tiles), ctypes.byref(chunks),
                                      ctypes.byref(nd), ctypes.byref(mbytes)))
    grid, tmap = ctypes.c_int32(), ctypes.c_int32()
    _lib.check(lib.mk_csr_launch_info(op.handle, ctypes.byref(grid), ctypes.byref(tmap)))
    return {"format": fmt.value, "format_name": FMT_NAMES[fmt.value], "tiles_windowed": tiles.value,
            "lds_window_chunks": chunks.value if fmt.value != 3 else 0,
            "column_phases": chunks.value if fmt.value == 3 else 0, "dictionary_size": nd.value,
            "matrix_bytes_per_product": mbytes.value, "grid": grid.value, "tile_order": tmap.value}


def colblocks(lib, op):
    from pykrylov_amd import _lib
    k = ctypes.c_int32()
    _lib.check(lib.mk_csr_colblocks(op.handle, ctypes.byref(k)))
    return k.value


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
        avg = ctypes.c_double(float("nan"))
        if kind == _lib.MK_BICGSTAB:                         # (the other solvers' product kernels: see profiles/)
            _lib.check(lib.mk_solver_time_spmv(run.handle, 200, ctypes.byref(avg)))
        run.close()
        return dt, avg.value

    # configs[2]: BiCGSTAB, random nonsymmetric diagonally dominant CSR, n = 1e6, ~5 nnz/row.  With threshold 0
    # the residual reaches 0 after ~25 passes and the recurrence then divides 0 by 0 (as the reference would):
    # the kernels move the same bytes on NaNs, which is what is being timed.
    n = 1000000
    op = gallery.random_diagdom(n, seed=1)
    ones = _lib.DeviceArray.from_numpy(np.ones(n))
    rhs = _lib.DeviceArray(n)
    op.spmv_device(ones.ptr, rhs.ptr)
    dt, spmv_us = timed(op, _lib.MK_BICGSTAB, rhs, abstol=0.0, reltol=0.0, matvec_max=1 << 60)
    b_spmv = spmv_bytes(n, n, op.nnz)
    b_iter = 2 * b_spmv + 224 * n
    out["bicgstab-rand1m@1"] = {"value": passes / dt, "unit": "iterations/s", "ms_per_step": 1e3 * dt / passes,
                                "rows": n, "nnz": int(op.nnz), "matvecs_per_iteration": 2,
                                "format": format_info(lib, op),
                                "column_blocks": colblocks(lib, op),
                                "spmv": {"avg_product_us": spmv_us, "achieved_GBs": b_spmv / spmv_us / 1e3,
                                         "frac": b_spmv / spmv_us / 1e3 / HBM_PEAK_GBS,
                                         "note": "one product = all column-block launches of the first product's kernel"},
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
    dt, spmv_us = timed(op, _lib.MK_MINRES, rhs_h, shift=1.5, itnlim=1 << 60, rtol=0.0, etol=0.0, window=5)
    b_spmv = spmv_bytes(n, n, op.nnz)
    b_iter = b_spmv + 176 * n                                              # SURVEY.md 8d (kwarg shift)
    out["minres-shifted2d-2000@1"] = {"value": passes / dt, "unit": "iterations/s", "ms_per_step": 1e3 * dt / passes,
                                      "rows": n, "nnz": int(op.nnz), "shift": 1.5, "format": format_info(lib, op),
                                      "iteration_roofline": {"algorithmic_bytes_per_iter": b_iter,
                                                             "frac_of_hbm": b_iter * passes / dt / 1e9 / HBM_PEAK_GBS}}
    op.free()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="auto")
    ap.add_argument("--exchange", default="both", choices=["both", "halo", "allgather"],
                    help="N > 1: iterate exchange before each product.  both: halo is `value`, all-gather is timed "
                         "beside it with a fifth of the steps")
    ap.add_argument("--event-stride", type=int, default=0,
                    help="additionally bracket the SpMV of every k-th pass with its own HIP event pair "
                         "(intrusive: each pair costs ~3-6 us; 0 = off)")
    ap.add_argument("--spmv-launches", type=int, default=200,
                    help="back-to-back launches of the fused SpMV kernel timed by one HIP event pair")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--one-device", action="store_true",
                    help="test aid: every rank on GPU 0 (RCCL refuses that, which exercises the host-staged fallback)")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "host"],
                    help="host: collectives staged through host memory over gloo, all ranks on GPU 0 -- a smoke test "
                         "of the N > 1 path on a single-GPU box, not a measurement")
    ap.add_argument("--all-configs", action="store_true",
                    help="also time BASELINE configs[2] (BiCGSTAB, random n=1e6) and configs[3] (MINRES, n=4e6)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_spawn(args.gpus)                                 # does not return
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world_size))
    td = torch = None
    json_out = sys.stdout
    if world_size > 1:
        # stdout carries exactly one line, the JSON record: RCCL prints a version banner there (at communicator
        # teardown), so everything else this process writes to fd 1 is sent to stderr
        sys.stdout.flush()
        json_out = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    if world_size > 1:
        # torch first: its bundled HIP runtime must be the one the process loads.  gloo only:
        # no second RCCL instance, no torch CUDA context needed for the bootstrap.
        import torch
        import torch.distributed as td
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        td.init_process_group(backend="gloo", rank=rank, world_size=world_size)
        if args.transport == "host" or args.one_device:
            local_rank = 0

    from pykrylov_amd import _lib, dist
    from pykrylov_amd.generic import DeviceRun
    lib = _lib.init(local_rank)
    world = dist.World(rank, world_size, td)
    transport_used, transport_note = args.transport, None
    if world_size > 1:
        # RCCL, checked with one small all-reduce; if any rank cannot get a working communicator, ALL ranks fall
        # back to the host-staged transport (gloo on pinned buffers): slow, but a measured line with the reason in
        # it is worth more than no line
        ok, err = True, ""
        try:
            world.init_device_comm(transport=args.transport)
            probe = (ctypes.c_double * 1)(float(rank + 1))
            _lib.check(lib.mk_comm_allreduce_host(probe, 1))
            if probe[0] != world_size * (world_size + 1) / 2.0:
                raise RuntimeError("all-reduce self-test returned %r" % probe[0])
        except Exception as e:                               # noqa: BLE001 - any failure triggers the fallback
            ok, err = False, "rank %d: %r" % (rank, e)
        states = world.allgather_object((ok, err))
        if not all(o for o, _ in states):
            first = [e for o, e in states if not o][0][:300]
            if args.transport != "rccl":
                raise SystemExit("communicator setup failed: " + first)
            lib.mk_comm_destroy()
            world.init_device_comm(transport="host")
            transport_used, transport_note = "host", "RCCL communicator unavailable (%s): host-staged gloo fallback" % first
            if rank == 0:
                sys.stderr.write("bench.py: " + transport_note + "\n")

    name = args.workload
    if name == "auto":
        name = "poisson3d-512"

    def device_sync():
        _lib.check(lib.mk_sync())
        if torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized():
            torch.cuda.synchronize()

    def barrier():
        if td is not None:
            td.barrier()

    def run_cg(workload, steps, warmup, stride, exchange="halo", spmv_launches=None, comm_probe=False):
        op, n_global, meta = build_workload(workload, world, exchange)
        n_local = getattr(op, "local_size", None) or op.shape[1]
        ones = _lib.DeviceArray.from_numpy(np.ones(op.shape[1]))
        rhs = _lib.DeviceArray(n_local)
        op.spmv_device(ones.ptr, rhs.ptr)                     # rhs = A * 1 (test_diagdom.py:78-79 convention)
        # tolerances 0: never converges, so exactly `steps` passes run inside the timed region
        run = DeviceRun(op, _lib.MK_CG, rhs, None, abstol=0.0, reltol=0.0, matvec_max=1 << 60,
                        check_curvature=1, spmv_event_stride=stride)
        run.setup()
        done_w = run.iterate(warmup)
        device_sync()
        barrier()
        t0 = time.perf_counter()
        done = run.iterate(steps)
        device_sync()
        elapsed = time.perf_counter() - t0
        if td is not None:
            t = torch.tensor([elapsed], dtype=torch.float64)
            td.all_reduce(t, op=td.ReduceOp.MAX)              # gloo, CPU tensor
            elapsed = float(t.item())
            barrier()
        timing = run.timing()
        res = run.finish()
        comm = None
        if td is not None:
            last = ctypes.c_double()
            _lib.check(lib.mk_csr_comm_last_us(op.handle, ctypes.byref(last)))
            comm = {"last_overlapped_halo_group_us": last.value}
            if comm_probe:
                # the collectives alone, back to back on the library stream (every rank takes part)
                ex_us, ar_us = ctypes.c_double(), ctypes.c_double()
                xbuf = _lib.DeviceArray(op.shape[1])
                _lib.check(lib.mk_comm_time_exchange(op.handle, xbuf.ptr, 20, ctypes.byref(ex_us)))
                _lib.check(lib.mk_comm_time_allreduce(2048, 50, ctypes.byref(ar_us)))
                xbuf.free()
                comm.update({"exchange_alone_us": ex_us.value, "allreduce_2048_doubles_us": ar_us.value,
                             "allreduces_per_step": 2})
            vals = [None] * world_size
            td.all_gather_object(vals, comm)
            comm = {"per_rank": vals}
            ni, nb = ctypes.c_int64(), ctypes.c_int64()
            _lib.check(lib.mk_csr_overlap_info(op.handle, ctypes.byref(ni), ctypes.byref(nb)))
            comm["tiles_interior_boundary_rank0"] = [ni.value, nb.value]
        # dominant kernel: the loop's fused SpMV, launched back to back with one HIP-event pair around the
        # whole train (on the solver's stream), after the timed region so that it does not disturb `value`
        timing["spmv_b2b_us"] = None
        nl = args.spmv_launches if spmv_launches is None else spmv_launches
        if stride >= 0 and nl > 0 and td is None:
            avg = ctypes.c_double()
            _lib.check(lib.mk_solver_time_spmv(run.handle, nl, ctypes.byref(avg)))
            timing["spmv_b2b_us"] = avg.value
        assert done == steps and done_w == warmup, (done, steps, done_w, warmup)
        assert np.isfinite(res.residNorm), "CG diverged"
        hist = run.history()
        info = dict(op_shape=op.shape, nnz=op.nnz, n_local=n_local, n_global=n_global, meta=meta, elapsed=elapsed,
                    timing=timing, resid_first=float(hist[0]), resid_last=float(hist[-1]), comm=comm,
                    fmt=format_info(lib, op), steps=steps, launches=nl)
        run.close()
        op.free()
        return info

    def roofline_of(info, workload):
        steps = info["steps"]
        n_g, n_l = info["n_global"], info["n_local"]
        b_spmv = spmv_bytes(n_l, info["op_shape"][1], info["nnz"])
        tm = info["timing"]
        spmv_us = tm["spmv_b2b_us"]
        achieved = b_spmv / (spmv_us * 1e-6) / 1e9 if spmv_us else None
        inloop_us = 1e3 * tm["spmv_ms"] / tm["spmv_launches"] if tm["spmv_launches"] else None
        stencil = info["meta"]["stencil"]
        nnz_global = stencil * n_g - 2 * sum(n_g // g for g in info["meta"]["grid"])
        its = steps / info["elapsed"]
        fmt = info["fmt"]
        # bytes this kernel actually has to stream: the matrix in its storage format + x once + y (+ nothing for
        # the fused dot): what an HBM counter would show with perfect reuse of x
        actual = fmt["matrix_bytes_per_product"] + 8 * info["op_shape"][1] + 8 * n_l
        traffic, tnote = None, "no PMC profile for this kernel build under profiles/spmv_traffic.json"
        tpath = os.path.join(ROOT, "profiles", "spmv_traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            ent = tj.get("%s@%d" % (workload, world_size))
            if ent and tj.get("kernel_source_sha") == kernel_source_sha() and ent.get("format") == fmt["format"]:
                traffic, tnote = ent, "measured with rocprofv3 PMC at this kernel build (%s)" % tj.get("measured", "?")
            elif ent:
                tnote = "profiles/spmv_traffic.json was measured at another kernel build or format: not quoted"
        roof = {"bound": "hbm", "kernel": "mk_spmv_kernel<CgSpmvEpi> (CSR SpMV + fused <p,Ap>), " + fmt["format_name"],
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic, "traffic_note": tnote,
                "bytes_per_launch": b_spmv, "avg_launch_us": spmv_us, "launches_timed": info["launches"],
                "method": "one hipEvent pair around back-to-back launches on the solver stream",
                "inloop_event_pair_us": inloop_us,
                "format_bytes_per_launch": actual,
                "frac_of_format_bytes": (actual / (spmv_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if spmv_us else None,
                "note": "achieved/frac price the launch at the ALGORITHMIC CSR bytes (12 nnz + 4(n+1) + 8 ncols + 8 nrows); "
                        "format_bytes_per_launch is what the storage format in use streams, so frac > frac_of_format_bytes "
                        "(and possibly > 1) when the format is more compact than CSR"}
        # whole-iteration rooflines: the reference's op count (B_spmv + 104 n per pass) and the bytes
        # the fused kernels of this implementation move (B_spmv + 64 n; with the format's matrix bytes)
        n_sum = n_g
        iter_bytes = spmv_bytes(n_g, n_g, nnz_global) + 104 * n_sum
        fused = spmv_bytes(n_g, n_g, nnz_global) + 64 * n_sum
        fused_fmt = (fmt["matrix_bytes_per_product"] * (n_g / float(n_l)) + 16 * n_g) + 64 * n_sum
        agg = HBM_PEAK_GBS * world_size
        it_roof = {"algorithmic_bytes_per_iter": iter_bytes, "achieved_GBs": iter_bytes * its / 1e9,
                   "frac_of_aggregate_hbm": iter_bytes * its / 1e9 / agg,
                   "fused_bytes_per_iter": fused, "frac_fused": fused * its / 1e9 / agg,
                   "format_fused_bytes_per_iter": int(fused_fmt), "frac_format_fused": fused_fmt * its / 1e9 / agg}
        return its, nnz_global, roof, it_roof

    multi = world_size > 1
    first_mode = "halo" if (not multi or args.exchange in ("both", "halo")) else "allgather"
    info = run_cg(name, args.steps, args.warmup, args.event_stride, exchange=first_mode, comm_probe=multi)
    elapsed = info["elapsed"]
    tm = info["timing"]
    n_g = info["n_global"]
    its, nnz_global, roof, it_roof = roofline_of(info, name)

    line = {
        "metric": METRIC,
        "value": its, "unit": "iterations/s", "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "CG %s (%d rows, %d nnz), rhs=A*1, x0=0, tolerances 0" % (name, n_g, nnz_global),
                   "baseline_config": ("configs[4]: CG on 3-D 7-point Poisson 512^3, the configuration the target is "
                                       "quoted on (fits one GPU: same problem at every N, strong scaling); configs[1] "
                                       "is reported under extra") if name == "poisson3d-512" else name,
                   "solver": "cg", "rows": n_g, "nnz": nnz_global, "storage_format": info["fmt"],
                   "parallelism": "1 GPU" if not multi else "row-partition x%d, %s exchange + allreduce(dots), %s"
                                  % (world_size, first_mode, "RCCL (gloo bootstrap)" if transport_used == "rccl"
                                     else "host-staged gloo (%s)" % (transport_note or "smoke test"))},
        "roofline": roof,
        "iteration_roofline": it_roof,
        "device_loop_ms": tm["iterate_ms"],
        "residual": {"first": info["resid_first"], "last": info["resid_last"]},
    }
    if multi:
        ex = {first_mode: {"value": its, "ms_per_step": 1e3 * elapsed / args.steps, "steps": args.steps,
                           "comm": info["comm"]}}
        if args.exchange == "both":
            s2 = max(10, args.steps // 5)
            i2 = run_cg(name, s2, max(5, args.warmup // 5), 0, exchange="allgather", comm_probe=True)
            ex["allgather"] = {"value": s2 / i2["elapsed"], "ms_per_step": 1e3 * i2["elapsed"] / s2, "steps": s2,
                               "comm": i2["comm"],
                               "note": "full-iterate all-gather (%.0f MB per product): north_star's general path; "
                                       "the halo exchange is the one `value` is quoted on" % (8e-6 * n_g)}
        line["exchange"] = ex
        # SpMV roofline of rank 0's share, timed without collectives is not possible per launch: quote the
        # per-step budget instead
        line["roofline"]["note_multi"] = ("N > 1: the SpMV kernel is not timed alone (each launch is preceded by an "
                                          "exchange); see iteration_roofline and exchange.*.comm")
    if rank == 0 and not multi and not args.no_cpu and not os.environ.get("BENCH_CHILD"):
        line["cpu_baseline"] = cpu_baseline(name)
        line["cpu_baseline_all_cores"] = cpu_baseline_all_cores(name)
    if not multi and name == "poisson3d-512" and not args.no_extra:
        # BASELINE configs[1] (CG, 2-D Poisson n = 1e6, one GPU): same measurement, reported beside the headline
        exi = run_cg("poisson2d-1000", 2000, 200, 0)
        e_its, e_nnz, e_roof, e_it = roofline_of(exi, "poisson2d-1000")
        line["extra"] = {"poisson2d-1000@1": {"value": e_its, "unit": "iterations/s", "steps": 2000, "warmup": 200,
                                              "ms_per_step": 1e3 * exi["elapsed"] / 2000, "roofline": e_roof,
                                              "iteration_roofline": e_it, "storage_format": exi["fmt"]}}
    if not multi and args.all_configs:
        line.setdefault("extra", {}).update(other_configs(lib))
    if rank == 0:
        json_out.write(json.dumps(line) + "\n")
        json_out.flush()
    if td is not None:
        barrier()
        lib.mk_comm_destroy()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
