#!/usr/bin/env python3
"""bench.py -- Krylov iterations/s of the device-resident CG + roofline of its SpMV kernel.

    python bench.py --gpus N --steps K --warmup W           # N = 1: in-process; N > 1: spawns N ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W   # same thing under an external launcher

A "step" is one pass of the CG loop body (reference pykrylov/cg/cg.py:113-158: 1 SpMV, 2 dots,
3 vector updates) on synthetic data resident in HBM.  Tolerances are set to zero so that exactly
K passes run inside the timed region.

OUTPUT.  Rank 0 prints ONE compact JSON line (<= 4096 bytes, `compact_line`): the contract keys, `config`, `roofline`
(the SpMV kernel: physical bytes / HIP-event time), `iteration_frac`, `residual`, `parity_vs_n1`, `placement_draws`,
`cpu_baseline`, the second workload as a block of the same shape (`second_workload`), and every other measured loop as
[iterations/s, physical fraction] (`cg_other_workloads`, `solver_loops`).  Everything else -- per-product tables, notes,
format descriptions, per-rank comm timings -- goes to bench_detail.json next to this file (BENCH_DETAIL overrides the path).

Workloads (BASELINE.json `configs`):
  poisson3d-512          configs[4]  CG, 3-D 7-point Poisson 512^3 (1.34e8 rows, 9.4e8 nnz; diagonal 6, off-diagonals -1),
                               row-partitioned over N GPUs.  THE DEFAULT: `value`, `roofline` and `cpu_baseline` are
                               quoted on it.  It fits one GPU, so the 1/2/4/8 series is one strong-scaling series over
                               a fixed problem.  Rows and values compress to one byte per row (storage formats 4 / 9).
  poisson3d-512-varcoef  configs[4]'s grid with a VARIABLE coefficient field (-div(k grad u), harmonic-mean face
                               coefficients, SPD, every stored value distinct): no constant-coefficient compression applies,
                               the product has to stream 8 bytes per nonzero.  `second_workload` at N = 1.
  poisson2d-1000         configs[1]  CG, 2-D 5-point Poisson, n = 1e6, one GPU; `cg_other_workloads`.
  bicgstab-rand1m        configs[2]  BiCGSTAB, random nonsymmetric n = 1e6, ~5 nnz/row, one GPU; `solver_loops`.
  minres-shifted2d-2000  configs[3]  MINRES, shifted 2-D Laplacian n = 4e6, one GPU; `solver_loops`.

Roofline convention (all figures PHYSICAL): `roofline.achieved` = bytes the kernel has to move in the storage format
in use (matrix data of the format + x once + y once) / its average duration; `frac` = achieved / 8 TB/s, never above
1 by construction.  The CSR-priced figure (12 nnz + 4 (n+1) + 8 ncols + 8 nrows over the same time) is
in bench_detail.json as `csr_equivalent_GBs` -- a throughput in CSR units, not a fraction of anything.

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
VARCOEF_SEED = 7
FMT_NAMES = {0: "csr (int32 columns + fp64 values, x gathered)",
             1: "windowed tiles (x windows in LDS, uint16 slots + fp64 values)",
             2: "windowed tiles + value dictionary (one packed 32-bit word per nonzero: LDS slot + value code)",
             3: "csr, tile resident in LDS, gathers ordered by column block (x longer than an L2)",
             4: "windowed tiles + value dictionary + row patterns (one byte per ROW: the number of its pattern of "
                "{LDS slot - lane, value code} words)",
             5: "windowed tiles + row patterns for the x positions (one byte per row) + the fp64 values streamed in "
                "tile-sliced ELL order (8 B per nonzero; no column indices, no row pointers)",
             6: "wide windowed tiles (rows <= 32 entries, 32 window chunks): uint16 LDS slots + fp64 values, both streamed in "
                "tile-sliced ELL order (10 B per nonzero)",
             7: "wide windowed tiles + row patterns (one byte per row) + fp64 values streamed in tile-sliced ELL order",
             8: "wide windowed tiles + value dictionary + row patterns (one byte per ROW, entries {offset, value} through "
                "the scalar cache)",
             9: "z-marching bricks for 7-point-class matrices (one byte per ROW + value dictionary; every x entry loaded "
                "once, the planes z-1, z, z+1 of a workgroup's rows ride in registers)",
             10: "z-marching bricks for 7-point-class matrices without a value dictionary (mask byte per row + seven fp64 values "
                 "per row streamed position-major: 56 B per row; every x entry loaded once)",
             11: "z-marching bricks for SYMMETRIC 7-point-class matrices without a value dictionary (mask byte per row + the diagonal "
                 "and the three upper fp64 values per row streamed position-major: 32 B per row; the lower values are the "
                 "neighbouring rows' upper ones, taken from registers / an LDS image of the plane's values)"}


SHORT_FMT = {0: "fmt0 csr, x gathered", 1: "fmt1 windowed tiles: u16 slots + f64 values",
             2: "fmt2 windowed tiles + value dictionary (4 B/nnz)", 3: "fmt3 csr, LDS-resident tile, column phases",
             4: "fmt4 pattern byte/row + value dictionary (0 B/nnz)", 5: "fmt5 pattern byte/row + streamed f64 values (8 B/nnz)",
             6: "fmt6 wide tiles: u16 slots + f64 values (10 B/nnz)", 7: "fmt7 wide tiles: pattern byte/row + f64 values",
             8: "fmt8 wide tiles: pattern byte/row + value dictionary", 9: "fmt9 z-marching bricks: pattern byte/row + value dictionary (0 B/nnz)",
             10: "fmt10 z-marching bricks: mask byte/row + streamed f64 values (56 B/row)",
             11: "fmt11 z-marching bricks, symmetric: mask byte/row + diagonal and upper f64 values (32 B/row)"}
BASELINE_CONFIG = {
    "poisson3d-512": "configs[4]: CG, 3-D 7-point Poisson 512^3 (diagonal 6, off-diagonals -1); fits one GPU: 1/2/4/8 = strong scaling",
    "poisson3d-512-varcoef": "configs[4]'s grid with a variable coefficient field (no constant-coefficient compression "
                             "applies: the product streams 8 B per nonzero)",
    "poisson2d-1000": "configs[1]", "bicgstab": "configs[2]", "minres": "configs[3]"}
LINE_LIMIT = 4096              # bytes of the ONE JSON line on stdout; everything else goes to bench_detail.json


def sig(v, n=6):
    """Floats cut to n significant digits (the compact line has a byte budget); non-finite -> None."""
    if isinstance(v, (bool, int, str)) or v is None:
        return v
    v = float(v)
    return float("%.*g" % (n, v)) if np.isfinite(v) else None


def _compact_roofline(r):
    tb = r.get("traffic_bytes")
    out = {"bound": r.get("bound", "hbm"), "kernel": str(r.get("kernel", ""))[:80], "achieved": sig(r.get("achieved")),
           "peak": r.get("peak", HBM_PEAK_GBS), "unit": r.get("unit", "GB/s"), "frac": sig(r.get("frac"), 4),
           "bytes_per_launch": r.get("bytes_per_launch"), "avg_launch_us": sig(r.get("avg_launch_us")),
           "traffic": tb, "traffic_bytes": tb}
    if tb and r.get("bytes_per_launch"):
        out["traffic_ratio"] = sig(tb / float(r["bytes_per_launch"]), 4)
    if tb:                                                   # a counter figure replayed from a committed profile, not collected now
        out["traffic_source"] = str(r.get("traffic_source", "profiles/spmv_traffic.json"))[:72]
    if r.get("fused_pass"):
        out["fused_pass"] = True
    return out


def _compact_cpu(c):
    if not c:
        return None
    out = {k: sig(c.get(k)) for k in ("value", "unit", "cores", "kind", "extrapolated", "sample_rows")}
    out["sample"] = str(c.get("sample_short") or c.get("sample", ""))[:64]
    return out


def _compact_cg(b):
    """The compact form of one CG workload block (same shape for the headline and the second workload)."""
    res, par = b.get("residual") or {}, b.get("parity_vs_n1")
    out = {"workload": "CG " + b["workload"], "value": sig(b["value"]), "unit": "iterations/s", "steps": b["steps"],
           "ms_per_step": sig(b["ms_per_step"]), "rows": b["rows"], "nnz": b["nnz"],
           "storage_format": SHORT_FMT.get(b["storage_format"]["format"], str(b["storage_format"]["format"])),
           "roofline": _compact_roofline(b["roofline"]),
           "iteration_frac": sig(b["iteration_roofline"]["frac_of_aggregate_hbm"], 4),
           "residual": {k: sig(res.get(k)) for k in ("true", "recurrence", "rel_gap", "ok")},
           "parity_vs_n1": ({"ok": par.get("ok"), "rel_hist_err": sig(par.get("rel_hist_err"), 3), "passes": par.get("passes"),
                             "fixture": "device-generated"} if par else None),
           "placement_draws": {"count": (b.get("placement_draws") or {}).get("count", 1),
                               "probe_s": sig((b.get("placement_draws") or {}).get("probe_seconds", 0.0), 3)}}
    if b.get("cpu_baseline"):
        out["cpu_baseline"] = _compact_cpu(b["cpu_baseline"])
    if b.get("cpu_baseline_all_cores"):
        out["cpu_baseline_all_cores"] = _compact_cpu(b["cpu_baseline_all_cores"])
    return out


def compact_line(detail):
    """The ONE JSON object bench.py prints (<= LINE_LIMIT bytes): the driver's contract keys, the headline workload's
    roofline / residual / parity / CPU baseline, the second workload as a block of the same shape, and every other
    measured loop as [iterations/s, physical fraction of the HBM roofline].  Everything else is in bench_detail.json.
    Pure function of the detail dict (tests/test_host.py runs it on canned details without a GPU)."""
    name = detail["headline"]
    head = _compact_cg(detail["workloads"][name])
    line = {"metric": detail["metric"], "value": head["value"], "unit": "iterations/s", "n_gpus": detail["n_gpus"],
            "steps": detail["steps"], "warmup": detail["warmup"], "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": detail["scaling"], "vs_baseline": None, "dtype": detail["dtype"],
            "data": detail["data"],
            "config": {"workload": head["workload"], "baseline_config": str(detail.get("baseline_config", ""))[:160],
                       "solver": "cg", "rows": head["rows"], "nnz": head["nnz"], "rhs": "A*1, x0=0, tolerances 0",
                       "parallelism": str(detail["parallelism"])[:200], "storage_format": head["storage_format"]}}
    for k in ("roofline", "iteration_frac", "residual", "parity_vs_n1", "placement_draws", "cpu_baseline",
              "cpu_baseline_all_cores"):
        if k in head:
            line[k] = head[k]
    for wname, b in detail["workloads"].items():
        if wname == name:
            continue
        if wname.startswith("poisson3d-512"):                 # the second workload: a block of the headline's shape
            sec = _compact_cg(b)                              # (less what the byte budget cannot afford twice)
            sec.pop("cpu_baseline_all_cores", None)
            sec["roofline"].pop("traffic_bytes", None)
            line["second_workload"] = sec
        else:                                                 # [iterations/s, SpMV physical frac, iteration physical frac]
            line.setdefault("cg_other_workloads", {})[wname] = [sig(b["value"], 5), sig(b["roofline"]["frac"], 3),
                                                                sig(b["iteration_roofline"]["frac_of_aggregate_hbm"], 3)]
    if detail.get("solver_loops"):
        # [iterations/s, physical HBM fraction of the pass] and, for loops whose products are bound by the per-entry cost of
        # divergent gathers (not by HBM), a third number: gather floor of the pass's products / their measured time
        loops = {"cg": [head["value"], head["iteration_frac"]]}
        for k, v in detail["solver_loops"].items():
            e = [sig(v["value"], 5), sig(v["iteration_roofline"]["frac"], 3)]
            if v.get("gather_bound"):
                e.append(sig(v["gather_bound"]["frac_of_gather_floor"], 3))
            loops[k.split("@")[0]] = e
        line["solver_loops"] = loops
        line["solver_loops_cols"] = "it/s, hbm frac[, gather-floor frac of the products: bound=gather]"
    if detail.get("csr_plain"):
        c = detail["csr_plain"]
        line["csr_plain"] = {"fmt": c["format"], "us": sig(c["avg_product_us"], 5), "bytes": c["bytes_per_launch"],
                             "frac": sig(c["frac"], 4), "traffic_ratio": sig(c.get("traffic_ratio"), 4),
                             "what": "512^3 product forced to fmt 0, priced at SURVEY 8(d) B_spmv"}
    if detail.get("build_sha"):
        line["build_sha"] = detail["build_sha"]
    if detail.get("transport"):
        line["transport"] = detail["transport"]
    if detail.get("exchange"):
        ex = {}
        for mode, e in detail["exchange"].items():
            per = (e.get("comm") or {}).get("per_rank") or []
            worst = {}
            for key in ("product_alone_us", "exchange_alone_us", "allreduce_2048_doubles_us",
                        "last_overlapped_halo_group_us", "device_loop_ms_per_step"):
                vals = [p[key] for p in per if p and p.get(key) is not None]
                if vals:
                    worst[key] = sig(max(vals), 5)
            ex[mode] = {"value": sig(e["value"]), "ms_per_step": sig(e["ms_per_step"]), "steps": e["steps"],
                        "max_over_ranks": worst}
        line["exchange"] = ex
    if detail.get("per_rank_budget"):
        line["per_rank_budget_us"] = detail["per_rank_budget"].get("kernels_per_pass_us")
    line["detail"] = "bench_detail.json"
    return line


def spmv_bytes(nrows, ncols, nnz):
    """Algorithmic bytes of one CSR SpMV launch: 12 nnz + 4 (n+1) + 8 ncols + 8 nrows."""
    return 12 * nnz + 4 * (nrows + 1) + 8 * ncols + 8 * nrows


def kernel_source_sha():
    """Fingerprint of the SpMV kernel sources: PMC traffic figures under profiles/ are only quoted while it matches."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "pykrylov_amd", "csrc")
    # every header a product kernel is compiled from: the per-format headers (all of them, whatever is added later), the
    # shared device code, the epilogues' stores (mk_solver.h) and the format builder
    files = sorted(f for f in os.listdir(csrc) if f.startswith("mk_spmv_fmt") and f.endswith(".h"))
    for f in ["mk_device.h", "mk_format.hip", "mk_internal.h", "mk_solver.h"] + files:
        with open(os.path.join(csrc, f), "rb") as fh:
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
        if name.startswith("stencil27-"):
            ms = min(m, 80)
            A = csr_ref.stencil27(ms, seed=VARCOEF_SEED if name.endswith("-varcoef") else 0)
        else:
            A = csr_ref.poisson3d_varcoef(ms, seed=VARCOEF_SEED) if name.endswith("-varcoef") else csr_ref.poisson3d(ms)
        scale = float(ms ** 3) / float(m ** 3)
        sample = ("first %%d CG iterations on %d^3 (%d rows, same operator family), measured iterations/s "
                  "EXTRAPOLATED by the rows ratio %.4g to %s" % (ms, ms ** 3, scale, name))
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
            "sample_short": "%d CG passes on %d rows%s" % (out["nMatvec"], n, ", scaled by the rows ratio" if scale != 1.0 else ""),
            "extrapolated": scale != 1.0, "measured_on_sample": out["nMatvec"] / dt, "sample_rows": int(n),
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


def host_mem_available_gb():
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                return int(ln.split()[1]) / 1048576.0
    except OSError:
        pass
    return None


def cpu_baseline_full_size(names, passes=3, threads=(1,)):
    """The CPU oracle MEASURED on the quoted workloads themselves (poisson3d-512 / poisson3d-512-varcoef: 134 217 728 rows,
    9.4e8 nonzeros, 11.8 GB of CSR arrays on the host + five 1 GiB vectors): per workload and thread count `passes` passes of
    the reference's CG loop (oracle/krylov_ref.cg: NumPy element-wise updates and np.dot on one thread, the C CSR product
    on OpenMP threads) after one untimed pass.  The matrices are written by the C generator twins (oracle/csr_ref.c,
    pinned bit for bit against the NumPy twins in tests/test_oracle_golden.py).  One child process per thread count
    (libgomp reads its thread count when it loads; the 17 GB are returned with the child).
    Returns {workload: {str(threads): {...}}}."""
    if os.environ.get("BENCH_CHILD") != "full":
        need = 24.0
        have = host_mem_available_gb()
        if have is not None and have < need:
            return {"value": None, "extrapolated": None,
                    "skipped": "host MemAvailable %.1f GB < %.0f GB needed for the 512^3 CSR arrays and vectors" % (have, need)}
        out = {}
        for th in threads:
            env = dict(os.environ, OMP_NUM_THREADS=str(th), OPENBLAS_NUM_THREADS="1", BENCH_CHILD="full")
            code = ("import json,sys; sys.path.insert(0, %r); import bench; "
                    "print(json.dumps(bench.cpu_baseline_full_size(%r, %d, (%d,))))" % (ROOT, list(names), passes, th))
            try:
                res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
                got = json.loads(res.stdout.strip().splitlines()[-1])
            except Exception as e:                           # a baseline must never take the bench line down
                return {"value": None, "error": repr(e)[:300]}
            for wname, per in got.items():
                out.setdefault(wname, {}).update(per)
        return out
    from oracle import csr_ref, krylov_ref
    th = int(threads[0])
    out = {}
    for name in names:
        m = int(name.split("-")[1])
        t0 = time.perf_counter()
        A = csr_ref.poisson3d_varcoef_c(m, seed=VARCOEF_SEED) if name.endswith("-varcoef") else csr_ref.poisson3d_c(m)
        t_gen = time.perf_counter() - t0
        n = A.shape[0]
        rhs = A.matvec(np.ones(n))
        krylov_ref.cg(A, rhs, abstol=0.0, reltol=0.0, matvec_max=1)          # untimed: first touch of every page
        t0 = time.perf_counter()
        res = krylov_ref.cg(A, rhs, abstol=0.0, reltol=0.0, matvec_max=passes)
        dt = time.perf_counter() - t0
        assert res["nMatvec"] == passes and np.isfinite(res["residHistory"][-1])
        out[name] = {str(th): {
            "value": passes / dt, "unit": "iterations/s", "cores": th, "kind": "port", "extrapolated": False,
            "sample_rows": int(n), "sample_nnz": int(A.nnz), "seconds_per_pass": dt / passes,
            "matrix_generation_seconds": t_gen,
            "sample_short": "%d CG passes of the workload itself, 1 untimed; %d thread(s)" % (passes, th),
            "sample": "%d CG passes of %s itself (%d rows, %d nnz) after 1 untimed pass; NumPy updates and np.dot on one "
                      "thread, C CSR product on %d OpenMP thread(s)" % (passes, name, n, A.nnz, th),
            "host_cpus": os.cpu_count(), "residual_after": float(res["residHistory"][-1])}}
        del A, rhs, res
    return out


# ======================================================================================
# N > 1 validates itself against a committed N = 1 device run (SURVEY.md 8e "P > 1 vs P = 1")
# ======================================================================================
PARITY_PASSES = 60
DEV_HIST = os.path.join(ROOT, "tests", "golden", "dev_hist_512.npz")   # device-generated (tools/make_dev_hist.py)


def rel_hist_err(h, href):
    """Parity metric of SURVEY.md 7.4-4: max |h - href| / max(href, 1e-4 href[0])."""
    h, href = np.asarray(h, dtype=float), np.asarray(href, dtype=float)
    if h.shape != href.shape:
        return float("inf")
    return float(np.max(np.abs(h - href) / np.maximum(href, 1e-4 * href[0])))


def n1_history(workload):
    """Residual history of the first PARITY_PASSES CG passes of `workload` as ONE MI355X produced it (rhs = A 1, x0 = 0);
    None when the fixture holds no run of this workload."""
    if not os.path.exists(DEV_HIST):
        return None
    z = np.load(DEV_HIST, allow_pickle=False)
    key = "hist_" + workload.replace("-", "_")
    return np.array(z[key]) if key in z.files else None


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
        seed = VARCOEF_SEED if name.endswith("-varcoef") else None
        meta = {"grid": [m, m, m], "stencil": 7, "coefficients": "variable (seed %d)" % seed if seed is not None
                else "constant"}
        if world.nranks == 1:
            return (gallery.poisson3d_varcoef(m, seed=seed) if seed is not None else gallery.poisson3d(m)), n, meta
        op, _ = dist.partition_poisson3d(world, m, m, m, mode=exchange, varcoef_seed=seed)
        return op, n, meta
    if name.startswith("stencil27-"):                        # (27-point box stencil, HPCG's sparsity; single GPU only)
        m = int(name.split("-")[1])
        seed = VARCOEF_SEED if name.endswith("-varcoef") else 0
        if world.nranks != 1:
            raise SystemExit("workload %r runs on one GPU" % name)
        return gallery.stencil27(m, seed=seed), m ** 3, {"grid": [m, m, m], "stencil": 27, "coefficients":
                                                          "variable (seed %d)" % seed if seed else "constant (-1 / 26)"}
    raise SystemExit("unknown workload %r" % name)


def format_info(lib, op):
    from pykrylov_amd import _lib
    fmt, chunks, nd = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    tiles, mbytes = ctypes.c_int64(), ctypes.c_int64()
    _lib.check(lib.mk_csr_format_info(op.handle, ctypes.byref(fmt), ctypes.byref(tiles), ctypes.byref(chunks),
                                      ctypes.byref(nd), ctypes.byref(mbytes)))
    grid, tmap = ctypes.c_int32(), ctypes.c_int32()
    _lib.check(lib.mk_csr_launch_info(op.handle, ctypes.byref(grid), ctypes.byref(tmap)))
    march = (ctypes.c_int64 * 12)()
    _lib.check(lib.mk_csr_march_info(op.handle, march, 12))
    return {"march_general": bool(march[9]), "format": fmt.value, "format_name": FMT_NAMES.get(fmt.value, "format %d" % fmt.value), "tiles_windowed": tiles.value,
            "lds_window_chunks": chunks.value if fmt.value != 3 else 0,
            "column_phases": chunks.value if fmt.value == 3 else 0, "dictionary_size": nd.value,
            "matrix_bytes_per_product": mbytes.value, "grid": grid.value, "tile_order": tmap.value}


def colblocks(lib, op):
    from pykrylov_amd import _lib
    k = ctypes.c_int32()
    _lib.check(lib.mk_csr_colblocks(op.handle, ctypes.byref(k)))
    return k.value


GATHER_CLK = 2.9               # clocks per gathered entry and CU of the scattered products (measured)
CLOCK_GHZ = 2.4                # MI355X engine clock

LOOP_BYTES = {
    # bytes per pass BEYOND the products (each product priced at matrix data of its storage format + input once + output
    # once), counted from the fused kernels of csrc/mk_*.hip; m = rows, n = columns of A
    "bicgstab": lambda m, n: 136 * n, "cgs": lambda m, n: 128 * n, "tfqmr": lambda m, n: 256 * n,
    "minres": lambda m, n: 96 * n, "symmlq": lambda m, n: 88 * n,
    "lsqr": lambda m, n: 24 * m + 56 * n, "lsmr": lambda m, n: 24 * m + 72 * n,
    "craig": lambda m, n: 64 * m + 72 * n, "craigmr": lambda m, n: 80 * m + 24 * n,
}
REF_LOOP_BYTES = {   # the reference's own op count per pass beyond the products, square solvers
    "bicgstab": 224, "cgs": 232, "tfqmr": 368, "minres": 176, "symmlq": 168}


def random_tall_csr(m, n, k=5, seed=11):
    """Seeded m x n matrix with k entries per row for the least-squares loops: entry j of a row lies at a uniformly random
    column of the j-th of k equal column ranges (so columns ascend and never repeat: canonical CSR by construction),
    values standard normal.  Tall and random: well conditioned."""
    rng = np.random.default_rng(seed)
    w = n // k
    cols = (rng.integers(0, w, size=(m, k)) + np.arange(k, dtype=np.int64)[None, :] * w).astype(np.int32)
    data = rng.standard_normal((m, k))
    indptr = (np.arange(m + 1, dtype=np.int64) * k).astype(np.int32)
    return indptr, cols.reshape(-1), data.reshape(-1)


def other_configs(lib, passes=400, warm=20, only=None):
    """Every solver loop north_star names that is not CG, on one GPU: BASELINE configs[2] (BiCGSTAB) and configs[3]
    (MINRES), CGS and TFQMR on the configs[2] matrix, SYMMLQ on the configs[3] matrix, LSQR / LSMR / CRAIG / CRAIG-MR on
    a seeded 4e6 x 1e6 matrix with 5 nonzeros per row.  Per loop: passes per second with the tolerances at zero, each
    product kernel of the pass timed alone (back-to-back launches, one HIP event pair), and PHYSICAL rooflines -- bytes the
    kernels have to move in the storage format in use.  Loops that converge to the last bit within a few dozen passes
    (everything but MINRES / SYMMLQ here) are timed on FINITE data: the run is re-set-up every `seg` passes, the first
    `head` passes of a segment are untimed, the rest is timed with HIP events on the solver stream, and the residual is
    asserted finite and positive at the end of every segment."""
    from pykrylov_amd import _lib, gallery
    from pykrylov_amd.linop import CsrOperator
    from pykrylov_amd.generic import DeviceRun
    out = {}

    def spmv_entry(us, b_fmt, b_csr, note=None):
        d = {"avg_product_us": us, "bytes_per_launch": int(b_fmt), "achieved_GBs": b_fmt / us / 1e3,
             "frac": b_fmt / us / 1e3 / HBM_PEAK_GBS, "csr_equivalent_GBs": b_csr / us / 1e3}
        if note:
            d["note"] = note
        return d

    def segmented(run, passes, head=2, seg_try=(16, 12, 10, 8, 6, 4), lls=False):
        """(passes timed, seconds, smallest residual seen, seg) on finite data; `run` is re-set-up per segment.  A segment
        is good when the loop is still running at its end (with zero tolerances a halt means the machine-precision or
        breakdown tests fired) and the residual it reports is finite (and positive, where the solver reports one: LSMR
        reports ||r|| in aux[1], CRAIG-MR none)."""
        def state():
            r = run.finish()
            vals = (float(r.residNorm), float(r.aux[0]), float(r.aux[1])) if lls else (float(r.residNorm),)
            rn = max(vals) if lls else vals[0]
            good = (not r.halted) and all(np.isfinite(v) for v in vals) and (lls or rn > 0.0)
            return good, rn
        seg = None
        for cand in seg_try:
            run.setup()
            done = run.iterate(cand)
            good, _ = state()
            if done == cand and good:
                seg = cand
                break
        assert seg is not None, "no segment length keeps the data finite"
        total_ms, done_all, resid_min = 0.0, 0, float("inf")
        while done_all < passes:
            run.setup()
            assert run.iterate(head) == head
            assert run.iterate(seg - head) == seg - head
            total_ms += run.timing()["iterate_ms"]
            done_all += seg - head
            good, rn = state()
            assert good, "segment ended on non-finite data or a halted loop: %r" % rn
            resid_min = min(resid_min, rn)
        return done_all, total_ms * 1e-3, resid_min, seg

    def gather_bound(prods, entries):
        """Products that gather x entry by entry (formats 0 / 3 on scattered columns) are bound by the per-entry cost of
        divergent gathers -- 2.9 clocks per gathered entry and CU (profiles/r04_*phase_stamps*) -- before
        HBM: floor = entries x 2.9 clk / 256 CUs at 2.4 GHz per product."""
        floor_us = [e * GATHER_CLK / 256.0 / CLOCK_GHZ * 1e-3 for e in entries]
        meas = [p[1] for p in prods]
        if not all(meas):
            return None
        return {"bound": "gather", "clocks_per_entry_and_cu": GATHER_CLK, "clock_GHz": CLOCK_GHZ,
                "floor_us_per_product": floor_us, "measured_us_per_product": meas,
                "frac_of_gather_floor": sum(floor_us) / sum(meas)}

    def entry(name, key, done, dt, m, n, nnz, fmt_a, prods, extra):
        """prods: list of (label, avg_us, physical bytes, csr bytes) of the pass's product kernels."""
        b_prod = sum(p[2] for p in prods)
        b_csr = sum(p[3] for p in prods)
        b_loop = LOOP_BYTES[key](m, n)
        e = {"value": done / dt, "unit": "iterations/s", "ms_per_step": 1e3 * dt / done, "steps": done,
             "rows": m, "cols": n, "nnz": int(nnz), "matvecs_per_iteration": len(prods), "format": fmt_a,
             "products": {p[0]: spmv_entry(p[1], p[2], p[3]) for p in prods if p[1]},
             "iteration_roofline": {"bytes_per_iter": int(b_prod + b_loop),
                                    "frac": (b_prod + b_loop) * done / dt / 1e9 / HBM_PEAK_GBS,
                                    "achieved_GBs": (b_prod + b_loop) * done / dt / 1e9,
                                    "note": "%d product(s) in the format in use + %d bytes of the fused update kernels"
                                            % (len(prods), b_loop)}}
        if key in REF_LOOP_BYTES:
            ref = b_csr + REF_LOOP_BYTES[key] * n
            e["iteration_roofline"]["reference_op_count_bytes_per_iter"] = int(ref)
            e["iteration_roofline"]["reference_op_count_GBs"] = ref * done / dt / 1e9
        e.update(extra)
        out[name] = e

    want = lambda k: only is None or k in only                # noqa: E731

    # ---- configs[2] matrix: random nonsymmetric diagonally dominant CSR, n = 1e6, ~5 nnz/row -- BiCGSTAB, CGS, TFQMR
    if want("bicgstab") or want("cgs") or want("tfqmr"):
        n = 1000000
        op = gallery.random_diagdom(n, seed=1)
        ones = _lib.DeviceArray.from_numpy(np.ones(n))
        rhs = _lib.DeviceArray(n)
        op.spmv_device(ones.ptr, rhs.ptr)
        for key, kind, label in (("bicgstab", _lib.MK_BICGSTAB, "bicgstab-rand1m@1"), ("cgs", _lib.MK_CGS, "cgs-rand1m@1"),
                                 ("tfqmr", _lib.MK_TFQMR, "tfqmr-rand1m@1")):
            if not want(key):
                continue
            run = DeviceRun(op, kind, rhs, None, abstol=0.0, reltol=0.0, matvec_max=1 << 60)
            run.setup()
            run.iterate(min(warm, 8))
            done, dt, rmin, seg = segmented(run, passes)
            us = [run.time_product(w) for w in (0, 1)]
            run.close()
            fmt = format_info(lib, op)
            b_csr = spmv_bytes(n, n, op.nnz)
            b_fmt = fmt["matrix_bytes_per_product"] + 16 * n
            entry(label, key, done, dt, n, n, op.nnz, fmt,
                  [("first (A p / A y)", us[0], b_fmt, b_csr), ("second (A z)", us[1], b_fmt, b_csr)],
                  {"column_blocks": colblocks(lib, op),
                   "gather_bound": gather_bound([("", us[0]), ("", us[1])], [op.nnz, op.nnz]) if fmt["format"] in (0, 3) else None,
                   "spmv": spmv_entry(us[0], b_fmt, b_csr),
                   "data": "finite: re-set-up every %d passes, first 2 of each segment untimed; smallest residual norm seen "
                           "%.3e" % (seg, rmin)})
        for b in (ones, rhs):
            b.free()
        op.free()

    # ---- configs[3] matrix: 2-D Laplacian m = 2000 (n = 4e6), keyword shift 1.5 (symmetric indefinite) -- MINRES, SYMMLQ
    if want("minres") or want("symmlq"):
        mg = 2000
        n = mg * mg
        op = gallery.poisson2d(mg)
        ones = _lib.DeviceArray.from_numpy(np.ones(n))
        rhs = _lib.DeviceArray(n)
        op.spmv_device(ones.ptr, rhs.ptr)
        rhs_h = rhs.to_numpy() - 1.5
        for key, kind, label, params, lanczos in (
                ("minres", _lib.MK_MINRES, "minres-shifted2d-2000@1",
                 dict(shift=1.5, itnlim=1 << 60, rtol=0.0, etol=0.0, window=5), 24),
                ("symmlq", _lib.MK_SYMMLQ, "symmlq-shifted2d-2000@1",
                 dict(shift=1.5, has_shift=1, matvec_max=1 << 60, rtol=0.0), 24)):
            if not want(key):
                continue
            run = DeviceRun(op, kind, rhs_h, None, **params)
            run.setup()
            assert run.iterate(warm) == warm
            _lib.check(lib.mk_sync())
            t0 = time.perf_counter()
            done = run.iterate(passes)
            _lib.check(lib.mk_sync())
            dt = time.perf_counter() - t0
            assert done == passes, (key, done, passes)
            rn = float(run.finish().residNorm)
            assert np.isfinite(rn), (key, rn)
            us = run.time_product(0)
            run.close()
            fmt = format_info(lib, op)
            b_csr = spmv_bytes(n, n, op.nnz)
            b_fmt = fmt["matrix_bytes_per_product"] + 16 * n
            entry(label, key, done, dt, n, n, op.nnz, fmt, [("A y + fused Lanczos step", us, b_fmt + lanczos * n, b_csr)],
                  {"shift": 1.5, "residual_norm_after": rn,
                   "spmv": spmv_entry(us, b_fmt + lanczos * n, b_csr,
                                      "product + fused Lanczos step (reads r1, writes y and v: %d n more)" % lanczos)})
            # (the Lanczos step's bytes are inside the product entry: take them out of the loop's extra)
            ir = out[label]["iteration_roofline"]
            ir["bytes_per_iter"] = int(b_fmt + LOOP_BYTES[key](n, n))
            ir["achieved_GBs"] = ir["bytes_per_iter"] * done / dt / 1e9
            ir["frac"] = ir["achieved_GBs"] / HBM_PEAK_GBS
            ir["note"] = "product in the format in use + the %d n bytes of the fused kernels" % (LOOP_BYTES[key](1, 1))
        for b in (ones, rhs):
            b.free()
        op.free()

    # ---- least squares: seeded 4e6 x 1e6, 5 nnz/row; A.T is a second device CSR (mk_csr_transpose)
    if any(want(k) for k in ("lsqr", "lsmr", "craig", "craigmr")):
        m, n = 4000000, 1000000
        indptr, indices, data = random_tall_csr(m, n)
        op = CsrOperator(indptr, indices, data, (m, n))
        del indptr, indices, data
        At = op.T
        xs = _lib.DeviceArray.from_numpy(np.random.default_rng(12).standard_normal(n))
        rhs = _lib.DeviceArray(m)
        op.spmv_device(xs.ptr, rhs.ptr)                       # consistent system: b = A x*
        xs.free()
        for key, kind in (("lsqr", _lib.MK_LSQR), ("lsmr", _lib.MK_LSMR), ("craig", _lib.MK_CRAIG),
                          ("craigmr", _lib.MK_CRAIGMR)):
            if not want(key):
                continue
            run = DeviceRun(op, kind, rhs, None, transpose=At, itnlim=1 << 60, damp=0.0, atol=0.0, btol=0.0, conlim=0.0,
                            etol=0.0, window=5)
            done, dt, rmin, seg = segmented(run, max(120, passes // 2), lls=True)
            us = [run.time_product(w) for w in (0, 1)]
            run.close()
            fa, ft = format_info(lib, op), format_info(lib, At)
            ba = (fa["matrix_bytes_per_product"] + 8 * n + 8 * m, spmv_bytes(m, n, op.nnz))
            bt = (ft["matrix_bytes_per_product"] + 8 * m + 8 * n, spmv_bytes(n, m, op.nnz))
            entry("%s-rand4m-x-1m@1" % key, key, done, dt, m, n, op.nnz, fa,
                  [("A v (+ u update, <u,u>)", us[0], ba[0], ba[1]), ("A.T u (+ v update, <v,v>)", us[1], bt[0], bt[1])],
                  {"format_transpose": ft,
                   "gather_bound": gather_bound([("", us[0]), ("", us[1])], [op.nnz, op.nnz])
                   if fa["format"] in (0, 3) and ft["format"] in (0, 3) else None,
                   "data": "finite: re-set-up every %d passes, first 2 of each segment untimed; smallest residual norm seen "
                           "%.3e" % (seg, rmin)})
        rhs.free()
        At.free() if hasattr(At, "free") else None
        op.free()
    return out


def csr_plain_512(lib, m=512, launches=50):
    """north_star's literal kernel -- coalesced reads of indptr / indices / data, x gathered -- on the headline matrix: the
    512^3 constant-coefficient product forced to plain CSR (storage format 0), CG's product kernel (SpMV + fused <p, Ap>)
    timed back to back and priced at SURVEY.md 8(d)'s B_spmv = 12 nnz + 4 (n + 1) + 8 ncols + 8 nrows, which here IS what
    moves.  What a caller whose matrix is in no compressible class gets."""
    from pykrylov_amd import _lib, gallery
    from pykrylov_amd.generic import DeviceRun
    n = m ** 3
    op = gallery.poisson3d(m)
    _lib.check(lib.mk_csr_set_format(op.handle, 0))
    ones = _lib.DeviceArray.from_numpy(np.ones(n))
    rhs = _lib.DeviceArray(n)
    op.spmv_device(ones.ptr, rhs.ptr)
    run = DeviceRun(op, _lib.MK_CG, rhs, None, abstol=0.0, reltol=0.0, matvec_max=1 << 60, check_curvature=1, placement_draws=1)
    run.setup()
    run.iterate(3)
    us = run.time_product(0, launches)
    fmt = format_info(lib, op)
    b = spmv_bytes(n, n, op.nnz)
    out = {"workload": "poisson3d-%d" % m, "format": fmt["format"], "kernel": "mk_spmv_kernel<CgSpmvEpiT,MkNoGate,false,0> (SpMV + fused <p,Ap>)",
           "avg_product_us": us, "launches_timed": launches, "bytes_per_launch": int(b), "achieved_GBs": b / us / 1e3,
           "frac": b / us / 1e3 / HBM_PEAK_GBS, "traffic_ratio": None,
           "note": "priced at SURVEY.md 8(d)'s CSR bytes: for this storage format they are the physical bytes"}
    tpath = os.path.join(ROOT, "profiles", "spmv_traffic.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        ent = tj.get("csr_plain@1")
        if ent and tj.get("kernel_source_sha") == kernel_source_sha() and ent.get("bytes"):
            out["traffic_bytes"] = ent["bytes"]
            out["traffic_ratio"] = ent["bytes"] / float(b)
    run.close()
    for buf in (ones, rhs):
        buf.free()
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
    ap.add_argument("--no-parity", action="store_true", help="skip the 60-pass parity segment of the 512^3 workloads")
    ap.add_argument("--parity", action="store_true",
                    help="N = 1: also run the 60-pass parity segment against tests/golden/dev_hist_512.npz (always on for N > 1)")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "host"],
                    help="host: collectives staged through host memory over gloo, all ranks on GPU 0 -- a smoke test "
                         "of the N > 1 path on a single-GPU box, not a measurement")
    ap.add_argument("--all-configs", action="store_true", help="(kept for compatibility: all BASELINE configs are in the "
                                                               "default line now)")
    ap.add_argument("--force-format", type=int, default=None,
                    help="profiling aid: force this storage format on the workload's matrix (0 = plain CSR: north_star's "
                         "literal kernel)")
    ap.add_argument("--only-other-configs", action="store_true",
                    help="profiling aid: run only configs[2] and configs[3] (BiCGSTAB random, MINRES shifted) and print them")
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

    if args.only_other_configs:
        only = os.environ.get("BENCH_ONLY_LOOPS")            # (profiling aid: comma-separated solver keys)
        json_out.write(json.dumps(other_configs(lib, only=only.split(",") if only else None)) + "\n")
        return
    name = args.workload
    if name == "auto":
        name = "poisson3d-512"                               # BASELINE.json configs[4], literally

    def device_sync():
        _lib.check(lib.mk_sync())
        if torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized():
            torch.cuda.synchronize()

    def barrier():
        if td is not None:
            td.barrier()

    def run_cg(workload, steps, warmup, stride, exchange="halo", spmv_launches=None, comm_probe=False, parity=False):
        arena_vecs = int(os.environ.get("BENCH_ARENA_VECTORS", "0"))
        if arena_vecs > 0 and world.nranks == 1 and workload.startswith("poisson3d-"):
            # placement experiment: the loop's vectors come from ONE block reserved BEFORE the matrix
            mm = int(workload.split("-")[1])
            _lib.check(lib.mk_arena_reserve(arena_vecs * (8 * mm ** 3 + (4 << 20))))
        op, n_global, meta = build_workload(workload, world, exchange)
        if args.force_format is not None:
            _lib.check(lib.mk_csr_set_format(op.handle, args.force_format))
        n_local = getattr(op, "local_size", None) or op.shape[1]
        ones = _lib.DeviceArray.from_numpy(np.ones(op.shape[1]))
        rhs = _lib.DeviceArray(n_local)
        op.spmv_device(ones.ptr, rhs.ptr)                     # rhs = A * 1 (test_diagdom.py:78-79 convention)
        parity_info = None
        href = n1_history(workload) if parity else None
        if parity:
            # the same PARITY_PASSES passes ONE GPU ran when the fixture was made: partitioning changes the summation order
            # of the dots (per-rank partial sums, all-reduced) and nothing else, so the history must agree to 1e-12
            prun = DeviceRun(op, _lib.MK_CG, rhs, None, abstol=0.0, reltol=0.0, matvec_max=PARITY_PASSES, check_curvature=1)
            prun.setup()                                      # (exactly PARITY_PASSES passes are enqueued: no halted launches
            prun.iterate(PARITY_PASSES)                       #  behind them, which would dilute a profiler's per-kernel averages)
            pres = prun.finish()
            ph = prun.history()
            prun.close()
            parity_info = {"passes": int(pres.nMatvec), "fixture": os.path.relpath(DEV_HIST, ROOT),
                           "fixture_has_workload": href is not None}
            if href is not None:
                err = rel_hist_err(ph, href)
                parity_info.update({"rel_hist_err": err, "bit_equal": bool(np.array_equal(ph, href)),
                                    "tolerance": 1e-12, "ok": bool(err <= 1e-12),
                                    "resid_first": float(ph[0]), "resid_last": float(ph[-1])})
                if not err <= 1e-12:
                    raise SystemExit("bench.py: N = %d history of %s is %.2e from the committed N = 1 device history "
                                     "(> 1e-12): the partitioned run does not reproduce the single-GPU one"
                                     % (world_size, workload, err))
        # tolerances 0: never converges, so exactly `steps` passes run inside the timed region
        run = DeviceRun(op, _lib.MK_CG, rhs, None, abstol=0.0, reltol=0.0, matvec_max=1 << 60,
                        check_curvature=1, spmv_event_stride=stride)
        run.setup()
        done_w = run.iterate(warmup)
        # Fused passes (csrc/mk_cg.hip) apply the x / p update of pass k inside the product kernel of pass k + 1.  In steady
        # state the K timed passes therefore execute exactly K updates -- the one left pending by the last warm-up pass and
        # K - 1 of their own -- and the K-th is applied when the iterate is asked for (`mk_solver_x`, right after the clock
        # stops: the residual check below needs it).  Nothing is skipped and nothing is counted twice; with NO warm-up pass
        # there is no update carried in, so the flush is taken inside the timed region instead.
        px0 = ctypes.c_void_p()
        device_sync()
        barrier()
        t0 = time.perf_counter()
        done = run.iterate(steps)
        if warmup == 0:
            _lib.check(lib.mk_solver_x(run.handle, ctypes.byref(px0)))
        device_sync()
        elapsed = time.perf_counter() - t0
        if td is not None:
            t = torch.tensor([elapsed], dtype=torch.float64)
            td.all_reduce(t, op=td.ReduceOp.MAX)              # gloo, CPU tensor
            elapsed = float(t.item())
            barrier()
        timing = run.timing()
        res = run.finish()
        # the line proves its own work: ||b - A x_k|| recomputed from the iterate with the plain product
        # (one exchange + one product + one norm, all ranks), against the residual the recurrence carries (cg.py:131,146,154)
        px = ctypes.c_void_p()
        _lib.check(lib.mk_solver_x(run.handle, ctypes.byref(px)))
        xext = _lib.DeviceArray(op.shape[1])
        _lib.check(lib.mk_memcpy_d2d(xext.ptr, px.value, 8 * n_local))
        if td is not None:
            _lib.check(lib.mk_exchange(op.handle, xext.ptr))
        axb = _lib.DeviceArray(n_local)
        _lib.check(lib.mk_spmv(op.handle, xext.ptr, axb.ptr))
        _lib.check(lib.mk_axpy(n_local, -1.0, rhs.ptr, axb.ptr))
        sq = ctypes.c_double()
        _lib.check(lib.mk_dot(n_local, axb.ptr, axb.ptr, ctypes.byref(sq)))
        sqv = (ctypes.c_double * 1)(sq.value)
        _lib.check(lib.mk_comm_allreduce_host(sqv, 1))
        true_resid = float(np.sqrt(sqv[0]))
        xext.free()
        axb.free()
        comm = None
        if td is not None:
            last = ctypes.c_double()
            _lib.check(lib.mk_csr_comm_last_us(op.handle, ctypes.byref(last)))
            comm = {"last_overlapped_halo_group_us": last.value}
            comm["device_loop_ms_per_step"] = timing["iterate_ms"] / max(1, steps)
            if comm_probe:
                # this rank's product kernel alone (all tiles in one launch, no exchange in front of it), and
                # the collectives alone, back to back on the library stream (every rank takes part)
                comm["product_alone_us"] = run.time_product(0, 50)
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
        fz = ctypes.c_int32()
        _lib.check(lib.mk_solver_fused(run.handle, ctypes.byref(fz)))
        assert done == steps and done_w == warmup, (done, steps, done_w, warmup)
        assert np.isfinite(res.residNorm), "CG diverged"
        hist = run.history()
        rel_gap = abs(true_resid - float(hist[-1])) / float(hist[0])
        info = dict(op_shape=op.shape, nnz=op.nnz, n_local=n_local, n_global=n_global, meta=meta, elapsed=elapsed,
                    timing=timing, resid_first=float(hist[0]), resid_last=float(hist[-1]), comm=comm,
                    fmt=format_info(lib, op), steps=steps, launches=nl, placement=dict(run.placement), parity=parity_info,
                    fused=bool(fz.value),
                    residual={"first": float(hist[0]), "last": float(hist[-1]), "recurrence": float(hist[-1]), "true": true_resid,
                              "rel_gap": rel_gap, "passes": int(res.nMatvec),
                              "note": "true = ||b - A x_k|| recomputed from the iterate after the timed region with the "
                                      "plain product; rel_gap = |true - recurrence| / ||r_0||, the line fails above 1e-10"})
        info["residual"]["ok"] = bool(rel_gap <= 1e-10)
        if not (rel_gap <= 1e-10) and workload == name:       # (the headline fails the line; extras carry ok = false)
            raise SystemExit("bench.py: %s: recurrence residual %.6e but true residual %.6e (gap / r0 = %.2e > 1e-10): "
                             "the timed kernels did not do CG's work" % (workload, hist[-1], true_resid, rel_gap))
        run.close()
        op.free()
        return info

    def roofline_of(info, workload):
        """Physical rooflines of a CG run: the SpMV kernel priced at the bytes it has to move in the storage format in
        use, the iteration at those plus the 64 n bytes of the two fused update kernels."""
        steps = info["steps"]
        n_g, n_l = info["n_global"], info["n_local"]
        b_csr = spmv_bytes(n_l, info["op_shape"][1], info["nnz"])
        tm = info["timing"]
        spmv_us = tm["spmv_b2b_us"]
        method = "one hipEvent pair around back-to-back launches on the solver stream"
        if spmv_us is None and (info.get("comm") or {}).get("per_rank"):
            # N > 1: this rank's product kernel alone (its whole slab in one launch, no exchange in front of it), 50 launches
            # back to back between one HIP-event pair (the comm probe of run_cg) -- the same measurement on the rank's share
            spmv_us = info["comm"]["per_rank"][rank].get("product_alone_us")
            method = "rank %d's product kernel on its own slab: one hipEvent pair around 50 back-to-back launches" % rank
        inloop_us = 1e3 * tm["spmv_ms"] / tm["spmv_launches"] if tm["spmv_launches"] else None
        stencil = info["meta"]["stencil"]
        nnz_global = stencil * n_g - 2 * sum(n_g // g for g in info["meta"]["grid"])
        if stencil == 27:
            nnz_global = int(np.prod([3 * g - 2 for g in info["meta"]["grid"]]))
        its = steps / info["elapsed"]
        fmt = info["fmt"]
        # bytes this kernel has to stream: the matrix in its storage format + x once + y once (nothing for the fused
        # dot: p[r] comes from the LDS window) -- what an HBM counter would show with perfect reuse of x
        fused = bool(info.get("fused"))
        # fused CG passes (storage format 9): the product kernel also carries the previous pass's x / p update -- p_old, r, x
        # in; p, x, A p out: 48 bytes per row instead of 16 -- and the pass has no third kernel
        b_fmt = fmt["matrix_bytes_per_product"] + ((48 * n_l) if fused else (8 * info["op_shape"][1] + 8 * n_l))
        achieved = b_fmt / (spmv_us * 1e-6) / 1e9 if spmv_us else None
        traffic, tnote = None, "no PMC profile for this kernel build under profiles/spmv_traffic.json"
        tpath = os.path.join(ROOT, "profiles", "spmv_traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            ent = tj.get("%s@%d" % (workload, world_size))
            if ent and tj.get("kernel_source_sha") == kernel_source_sha() and ent.get("format") == fmt["format"]:
                traffic, tnote = ent, "measured with rocprofv3 PMC at this kernel build (%s)" % tj.get("measured", "?")
            elif ent:
                tnote = "profiles/spmv_traffic.json was measured at another kernel build or format: not quoted"
        tfmt = {9: 11, 10: 12, 11: 13}.get(fmt["format"], fmt["format"]) + (3 if fmt.get("march_general") else 0)
        kname = ("mk_spmv_kernel<CgFusedEpiT,MkNoGate,false,%d> (x,p update + SpMV + <p,Ap>)" % tfmt if fused else
                 "mk_spmv_kernel<CgSpmvEpiT,MkNoGate,false,%d> (SpMV + fused <p,Ap>)" % tfmt)
        roof = {"bound": "hbm", "kernel": kname, "fused_pass": fused,
                "kernel_format": fmt["format_name"],
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                "traffic_bytes": (traffic or {}).get("bytes"), "traffic_note": tnote,
                "traffic_source": "profiles/spmv_traffic.json (%s)" % json.load(open(tpath)).get("measured", "?") if traffic else None,
                "bytes_per_launch": b_fmt, "avg_launch_us": spmv_us, "launches_timed": info["launches"],
                "method": method,
                "inloop_event_pair_us": inloop_us,
                "csr_bytes_per_launch": b_csr,
                "csr_equivalent_GBs": (b_csr / (spmv_us * 1e-6) / 1e9) if spmv_us else None,
                "note": "achieved/frac are PHYSICAL: bytes_per_launch = matrix data of the storage format in use + 8 ncols "
                        "(x once) + 8 nrows (y once); csr_equivalent_GBs prices the same launch at the CSR bytes of "
                        "SURVEY.md 8d (12 nnz + 4 (n+1) + 8 ncols + 8 nrows) and is a throughput, not a fraction of peak"}
        # whole iteration: format bytes of the product + the 64 n bytes of the two fused update kernels (r update + dot:
        # 16 n read, 8 n written; x, p update: 24 n read, 16 n written); beside it the reference's op count in CSR units
        scale = n_g / float(n_l)
        it_fmt = (fmt["matrix_bytes_per_product"] * scale + 16 * n_g) + 64 * n_g
        if fused:                                             # K1f 48 n + K2 24 n (the scalar kernel moves nothing)
            it_fmt = (fmt["matrix_bytes_per_product"] * scale + 48 * n_g) + 24 * n_g
        it_ref = spmv_bytes(n_g, n_g, nnz_global) + 104 * n_g
        agg = HBM_PEAK_GBS * world_size
        it_roof = {"bytes_per_iter": int(it_fmt), "achieved_GBs": it_fmt * its / 1e9,
                   "frac_of_aggregate_hbm": it_fmt * its / 1e9 / agg,
                   "note": ("physical: fused pass -- product kernel with the x / p update (matrix + 48 n) + 24 n of the r update"
                            if fused else "physical: product in the storage format in use + 64 n bytes of the fused update kernels"),
                   "reference_op_count_bytes_per_iter": it_ref, "reference_op_count_GBs": it_ref * its / 1e9,
                   "reference_op_count_note": "SURVEY.md 8d: B_spmv(CSR) + 104 n per pass; a throughput in the "
                                              "reference's units (for the 60 %% target: %.0f GB/s), not a fraction"
                                              % (0.6 * agg)}
        return its, nnz_global, roof, it_roof

    multi = world_size > 1
    first_mode = "halo" if (not multi or args.exchange in ("both", "halo")) else "allgather"
    full_512 = name in ("poisson3d-512-varcoef", "poisson3d-512")

    def cg_workload(wname, steps, warmup, stride=0, headline=False):
        """One CG workload measured end to end: the detail block both the compact line and bench_detail.json are cut from."""
        info = run_cg(wname, steps, warmup, stride, exchange=first_mode, comm_probe=multi and headline,
                      parity=(multi or args.parity or (wname in ("poisson3d-512-varcoef", "poisson3d-512")
                                                        and not args.no_parity)))
        w_its, w_nnz, w_roof, w_it = roofline_of(info, wname)
        blk = {"workload": wname, "value": w_its, "unit": "iterations/s", "steps": steps, "warmup": warmup,
               "ms_per_step": 1e3 * info["elapsed"] / steps, "rows": info["n_global"], "nnz": w_nnz,
               "roofline": w_roof, "iteration_roofline": w_it, "storage_format": info["fmt"],
               "device_loop_ms": info["timing"]["iterate_ms"], "residual": info["residual"],
               "parity_vs_n1": info["parity"], "placement_draws": info["placement"], "comm": info["comm"]}
        return blk

    head = cg_workload(name, args.steps, args.warmup, args.event_stride, headline=True)
    tkind, tranks, tsplit = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _lib.check(lib.mk_comm_transport(ctypes.byref(tkind), ctypes.byref(tranks), ctypes.byref(tsplit)))
    parallelism = "1 GPU" if not multi else ("row-partition x%d, %s exchange + allreduce(dots), %s"
                                             % (world_size, first_mode, "RCCL (gloo bootstrap)" if transport_used == "rccl"
                                                else "host-staged gloo (%s)" % (transport_note or "smoke test")))
    detail = {"metric": METRIC, "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup, "scaling": "strong",
              "dtype": "f64", "data": "synthetic", "headline": name, "parallelism": parallelism,
              "baseline_config": BASELINE_CONFIG.get(name, name), "workloads": {name: head},
              "kernel_source_sha": kernel_source_sha()}
    if multi:
        # what actually carried the collectives (a silent host-staged fallback must not pass for an RCCL number)
        detail["transport"] = {"kind": {0: "none", 1: "rccl", 2: "host-staged"}[tkind.value],
                               "rccl_ranks_seen": tranks.value, "halo_communicator_split": bool(tsplit.value)}
        if transport_used == "rccl" and not (tkind.value == 1 and tranks.value == world_size):
            raise SystemExit("bench.py: asked for RCCL over %d ranks but the communicator reports kind %d with %d ranks"
                             % (world_size, tkind.value, tranks.value))
        # what one rank's kernels were budgeted at when its slab was run alone on one GPU (profiles/r05_slab_budget.txt: the
        # brick march on the slab with fused passes -- interior planes, boundary planes, r update, scalar kernel, pack):
        # only the 8-way split of the 512^3 problem has a budget
        budget = None
        bpath = os.path.join(ROOT, "profiles", "r06_slab_budget.json")
        if world_size == 8 and name.startswith("poisson3d-512") and os.path.exists(bpath):
            bj = json.load(open(bpath))
            budget = dict(bj.get("varcoef" if name.endswith("-varcoef") else "const") or {})
            if budget:
                budget["source"] = "profiles/r06_slab_budget.json (%s)" % bj.get("source", "?")
            else:
                budget = None
        detail["per_rank_budget"] = budget
        ex = {first_mode: {"value": head["value"], "ms_per_step": head["ms_per_step"], "steps": args.steps,
                           "comm": head["comm"]}}
        if args.exchange == "both":
            s2 = max(10, args.steps // 5)
            i2 = run_cg(name, s2, max(5, args.warmup // 5), 0, exchange="allgather", comm_probe=True)
            ex["allgather"] = {"value": s2 / i2["elapsed"], "ms_per_step": 1e3 * i2["elapsed"] / s2, "steps": s2,
                               "comm": i2["comm"],
                               "note": "full-iterate all-gather (%.0f MB per product): north_star's general path; "
                                       "the halo exchange is the one `value` is quoted on" % (8e-6 * head["rows"])}
        detail["exchange"] = ex
        head["roofline"]["note_multi"] = ("N > 1: the SpMV kernel is not timed alone (each launch is preceded by an "
                                          "exchange); see iteration_roofline and exchange.*.comm")
    detail["build_sha"] = _lib.build_info(lib)
    if not multi and name == "poisson3d-512" and not args.no_extra:
        detail["csr_plain"] = csr_plain_512(lib)
    second = None
    if not multi and name == "poisson3d-512" and not args.no_extra:
        # the same grid with a VARIABLE coefficient field: no constant-coefficient compression applies, the product streams
        # 8 B per nonzero -- the CSR-class product north_star names, measured beside the literal configs[4] matrix
        second = "poisson3d-512-varcoef"
        detail["workloads"][second] = cg_workload(second, args.steps, args.warmup)
    if rank == 0 and not multi and not args.no_cpu and not os.environ.get("BENCH_CHILD"):
        small = cpu_baseline(name, seconds_budget=8.0 if full_512 else 20.0)
        head["cpu_baseline"] = small
        if full_512:
            # MEASURED at the quoted size, both matrices in one child process; the 128^3 sample scaled by the rows ratio
            # stays beside it in the detail file
            ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            wl = [name] + ([second] if second else [])
            full = cpu_baseline_full_size(wl, passes=3, threads=(1, min(ncpu, 64)))
            for wname in wl:
                got = full.get(wname) if isinstance(full, dict) else None
                blk = detail["workloads"][wname]
                if got and got.get("1") and got["1"].get("value"):
                    one = got["1"]
                    one["extrapolated_from_128cubed"] = small if wname == name else None
                    blk["cpu_baseline"] = one
                    blk["cpu_baseline_all_cores"] = got.get(str(min(ncpu, 64)))
                else:
                    blk.setdefault("cpu_baseline", small if wname == name else cpu_baseline(wname, seconds_budget=8.0))
                    blk["cpu_baseline"]["full_size_attempt"] = full if not isinstance(full, dict) or "error" in full \
                        or "skipped" in full else got
    if not multi and full_512 and not args.no_extra:
        # the other BASELINE configs and every other solver loop on one GPU, same measurement (bench_detail.json; the
        # compact line carries iterations/s and the physical fraction of each)
        others = [("poisson2d-1000", 2000, 200), ("stencil27-256", 400, 40), ("stencil27-256-varcoef", 200, 20)]
        for wname, st, wu in others:
            detail["workloads"][wname] = cg_workload(wname, st, wu)
        detail["workloads"]["stencil27-256"]["note"] = detail["workloads"]["stencil27-256-varcoef"]["note"] = (
            "not a BASELINE config: the 27-point box stencil (4.5e8 nonzeros), the matrix class of storage formats 6 / 7 / 8")
        detail["solver_loops"] = other_configs(lib)
    if rank == 0:
        line = compact_line(detail)
        text = json.dumps(line, separators=(",", ":"))
        dpath = os.environ.get("BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json"))
        try:
            with open(dpath, "w") as fh:
                json.dump(dict(detail, line=line), fh, indent=1)
        except OSError as e:                                  # (a read-only checkout must not cost the line)
            sys.stderr.write("bench.py: could not write %s: %r\n" % (dpath, e))
        assert len(text) <= LINE_LIMIT, "compact line is %d bytes" % len(text)
        json_out.write(text + "\n")
        json_out.flush()
    if td is not None:
        barrier()
        lib.mk_comm_destroy()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
