#!/usr/bin/env python3
"""(round 6: + the general-geometry march kernels, template values 14 / 15 / 16, and the plain-CSR twin of the headline)
(round 5: fused CG passes / storage format 9 -- the product kernel of a CG workload is CgFusedEpiT where the passes
are fused, template value 11 = storage format 9)
Summarise tools/profile_r06.sh: per-kernel durations (kernel traces, one per workload) and per-kernel counters (one
PMC run per counter set and workload), with the derived figures DESIGN.md uses.  Writes spmv_traffic.json (fabric-side
bytes per launch of the product kernels, stamped with the fingerprint of the kernel sources) next to the summary."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from pmc_summary2 import counters, report, trace  # noqa: E402

WORKLOADS = [("varcoef", "poisson3d-512-varcoef@1", "CG, 512^3 variable coefficients (second workload of the line)"),
             ("const", "poisson3d-512@1", "CG, 512^3 constant coefficients (the line's headline: BASELINE configs[4])"),
             ("p500", "poisson3d-500@1", "CG, 500^3 constant coefficients: the general-geometry brick march (round 6)"),
             ("plain", "csr_plain@1", "CG, 512^3 constant coefficients FORCED to plain CSR (storage format 0): north_star's literal kernel"),
             ("p2d", "poisson2d-1000@1", "CG, 2-D n = 1e6"),
             ("others", None, "the nine other solver loops: BiCGSTAB / CGS / TFQMR (random n = 1e6), MINRES / SYMMLQ (shifted 2-D n = 4e6), LSQR / LSMR / CRAIG / CRAIG-MR (random 4e6 x 1e6)")]


def main():
    out = sys.argv[1]
    import bench
    traffic = {"kernel_source_sha": bench.kernel_source_sha(), "measured": os.path.basename(out.rstrip("/")),
               "unit": "bytes per launch at the L2's fabric side: FETCH_SIZE x 2048 + WRITE_SIZE x 1024 (calibrated; "
                       "Infinity-Cache hits are counted)"}
    trace(out, "trace_default", "python bench.py (the driver's command: every workload in one process)")
    for w, key, title in WORKLOADS:
        dur = trace(out, "trace_" + w, title)
        tab = counters(out, "pmc_" + w)
        want = ("CgSpmvEpi", "CgFusedEpi", "CgUpdate", "cg_beta") if key else ("mk_spmv_kernel", "mk_stream_kernel")
        report(tab, dur, title, want)
        for k, c in tab.items():
            if "mk_spmv_kernel" in k and "Partial" not in k and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                fmt = int(k.rstrip(">").split(",")[-1])
                ent = {"bytes": int(c["FETCH_SIZE"] * 2048 + c["WRITE_SIZE"] * 1024),
                       "read_bytes": int(c["FETCH_SIZE"] * 2048), "written_bytes": int(c["WRITE_SIZE"] * 1024),
                       "format": {6: 5, 11: 9, 12: 10, 13: 11, 14: 9, 15: 10, 16: 11}.get(fmt, fmt), "kernel": k, "avg_us_in_trace": dur.get(k)}
                if key and key.startswith("stencil27"):      # (template values 7 / 8 -> formats 7 / 8)
                    ent["format"] = fmt
                if key and "CgFusedEpi" in k:                 # fused passes: THE product kernel of the workload
                    traffic[key] = ent
                elif key and "CgSpmvEpi" in k and "CgFusedEpi" not in traffic.get(key, {}).get("kernel", ""):
                    traffic[key] = ent
                elif not key:
                    traffic.setdefault("other_configs", {})[k] = ent
    cal = counters(out, "cal")
    print("== calibration (1 GiB streams): counter per launch -> bytes per count")
    for k, c in cal.items():
        for name in ("FETCH_SIZE", "WRITE_SIZE"):
            if name in c and (("read" in k) == (name == "FETCH_SIZE")):
                print("   %-60s %-11s %12.1f -> %.1f B/count" % (k[:60], name, c[name], 1073741824.0 / c[name]))
    json.dump(traffic, open(os.path.join(out, "spmv_traffic.json"), "w"), indent=1)
    print("\n== spmv_traffic.json\n" + json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
