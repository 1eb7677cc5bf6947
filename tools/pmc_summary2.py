#!/usr/bin/env python3
"""Summarise tools/profile_r02.sh: per-kernel durations (kernel traces) and per-kernel counters (one PMC run per
counter set), with the derived figures the limiter discussion in DESIGN.md uses.  Writes spmv_traffic.json (HBM-side
bytes per launch of the CG SpMV kernels, stamped with the fingerprint of the kernel sources) next to the summary."""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def read_csv(pattern):
    rows = []
    for f in glob.glob(pattern, recursive=True):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    return rows


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    m = re.match(r"(?:void )?(mk_\w+_kernel<[^(]*>)\(", name)
    if m:
        return m.group(1).replace(" ", "")
    return name.split("(")[0][:60]


def trace(out, sub, title):
    tr = read_csv(os.path.join(out, sub, "**", "*kernel_trace.csv"))
    agg = collections.OrderedDict()
    for r in tr:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = agg.setdefault(short(r["Kernel_Name"]), [0, 0])
        a[0] += 1
        a[1] += d
    tot = sum(a[1] for a in agg.values()) or 1
    print("== %s: kernel trace (calls, avg us, share of GPU time)" % title)
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print("%-78s %7d %10.2f %6.2f%%" % (k[:78], a[0], a[1] / a[0] / 1e3, 100.0 * a[1] / tot))
    print()
    return {k: a[1] / a[0] / 1e3 for k, a in agg.items()}


def counters(out, prefix):
    acc = collections.OrderedDict()
    for d in sorted(glob.glob(os.path.join(out, prefix + "_*"))):
        if not os.path.isdir(d):
            continue
        for r in read_csv(os.path.join(d, "**", "*counter_collection.csv")):
            key = (short(r["Kernel_Name"]), r["Counter_Name"])
            a = acc.setdefault(key, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    tab = collections.OrderedDict()
    for (k, c), a in acc.items():
        if a[0] >= 10:
            tab.setdefault(k, {})[c] = a[1] / a[0]
            tab[k]["_n"] = a[0]
    return tab


def report(tab, durations, title, want):
    print("== %s: counters per launch (averages), derived figures" % title)
    for k, c in tab.items():
        if not any(w in k for w in want):
            continue
        print("--", k, "(launches per pass: %d)" % c.get("_n", 0))
        for name in sorted(c):
            if name != "_n":
                print("     %-34s %16.1f" % (name, c[name]))
        g = c.get("GRBM_GUI_ACTIVE")
        if g:
            cyc = g / 8.0                                   # summed over the 8 XCDs
            if "TA_TA_BUSY_sum" in c:
                print("     > texture-address unit busy             %5.1f %% of the kernel (256 CUs)" % (100 * c["TA_TA_BUSY_sum"] / 256 / cyc))
            if "TA_FLAT_READ_WAVEFRONTS_sum" in c:
                print("     > TA busy cycles per wave-level read    %5.1f" % (c["TA_TA_BUSY_sum"] / c["TA_FLAT_READ_WAVEFRONTS_sum"]))
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            print("     > L2 hit rate                           %5.1f %%" % (100 * c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])))
        if "TCP_TCC_READ_REQ_sum" in c and "TCP_TOTAL_CACHE_ACCESSES_sum" in c:
            print("     > L1 -> L2 read requests per L1 access   %5.3f" % (c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"]))
        if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
            print("     > waves waiting (s_waitcnt / barrier)    %5.1f %% of wave time; issue stalls %5.1f %%" %
                  (100 * c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 100 * c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]))
        if "SQ_LDS_BANK_CONFLICT" in c and "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"]:
            print("     > LDS bank-conflict cycles               %5.1f %% of LDS-active cycles" % (100 * c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]))
        if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
            rd, wr = c.get("FETCH_SIZE", 0) * 2048.0, c.get("WRITE_SIZE", 0) * 1024.0
            us = durations.get(k)
            print("     > fabric-side bytes per launch: read %.1f MB (FETCH_SIZE x 2048, calibrated), written %.1f MB (WRITE_SIZE x 1024)%s"
                  % (rd / 1e6, wr / 1e6, "; %.2f TB/s over the traced %.1f us" % ((rd + wr) / us / 1e6, us) if us else ""))
        print()


def main():
    out = sys.argv[1]
    d3 = trace(out, "trace_3d", "bench.py 512^3 (configs[4] at N=1)")
    d2 = trace(out, "trace_2d", "bench.py 2-D n=1e6 + --all-configs (configs[1], [2], [3])")
    t3 = counters(out, "pmc3d")
    t2 = counters(out, "pmc2d")
    report(t3, d3, "512^3", ("CgSpmvEpi", "CgUpdate"))
    report(t2, d2, "2-D n=1e6 / random n=1e6 / 2-D n=4e6", ("mk_spmv_kernel",))
    # calibration
    cal = counters(out, "cal")
    print("== calibration (1 GiB streams): counter per launch -> bytes per count")
    for k, c in cal.items():
        for name in ("FETCH_SIZE", "WRITE_SIZE"):
            if name in c and (("read" in k) == (name == "FETCH_SIZE")):
                print("   %-60s %-11s %12.1f -> %.1f B/count" % (k[:60], name, c[name], 1073741824.0 / c[name]))
    # traffic file
    import bench
    traffic = {"kernel_source_sha": bench.kernel_source_sha(), "measured": os.path.basename(out.rstrip("/")),
               "unit": "bytes per launch at the L2's fabric side: FETCH_SIZE x 2048 + WRITE_SIZE x 1024 (calibrated; "
                       "Infinity-Cache hits are counted)"}
    for tab, key, fmtsrc in ((t3, "poisson3d-512@1", "bench_trace_3d.json"), (t2, "poisson2d-1000@1", "bench_trace_2d.json")):
        for k, c in tab.items():
            if "CgSpmvEpi" in k and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                fmt = int(k.rstrip(">").split(",")[-1])
                traffic[key] = {"bytes": int(c["FETCH_SIZE"] * 2048 + c["WRITE_SIZE"] * 1024),
                                "read_bytes": int(c["FETCH_SIZE"] * 2048), "written_bytes": int(c["WRITE_SIZE"] * 1024),
                                "format": fmt, "kernel": k}
    json.dump(traffic, open(os.path.join(out, "spmv_traffic.json"), "w"), indent=1)
    print("\n== spmv_traffic.json\n" + json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
