#!/usr/bin/env python3
"""Summarise tools/profile_bench.sh output: per-kernel durations (kernel trace) and per-kernel
FETCH_SIZE / WRITE_SIZE (PMC passes) scaled by the calibration runs."""
import collections
import csv
import glob
import os
import sys


def read_csv(pattern):
    rows = []
    for f in glob.glob(pattern, recursive=True):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    return rows


def short(name):
    return name.replace("(anonymous namespace)::", "")[:70]


def main():
    out = sys.argv[1]
    tr = read_csv(os.path.join(out, "trace", "**", "*kernel_trace.csv"))
    agg = collections.OrderedDict()
    for r in tr:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = agg.setdefault(r["Kernel_Name"], [0, 0])
        a[0] += 1
        a[1] += d
    tot = sum(a[1] for a in agg.values()) or 1
    print("== kernel trace: calls, avg us, share")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
        print("%-72s %7d %10.2f %6.2f%%" % (short(k), a[0], a[1] / a[0] / 1e3, 100.0 * a[1] / tot))

    # idle time between consecutive dispatches of the timed loop (launch-bound or not?)
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in tr)
    gaps = collections.defaultdict(list)
    for (s0, e0, n0), (s1, e1, n1) in zip(ev[:-1], ev[1:]):
        gaps[(short(n0)[:44], short(n1)[:44])].append(s1 - e0)
    print("\n== idle gap between consecutive dispatches (pairs seen > 50 times): median / mean us")
    for (a, b), v in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
        if len(v) > 50:
            v.sort()
            print("%-46s -> %-46s n=%6d  %6.2f / %6.2f" % (a, b, len(v), v[len(v) // 2] / 1e3, sum(v) / len(v) / 1e3))

    def counters(sub, prefix):
        rows = read_csv(os.path.join(out, sub, "**", "*counter_collection.csv"))
        acc = collections.OrderedDict()
        for r in rows:
            key = (r["Kernel_Name"], r["Counter_Name"])
            a = acc.setdefault(key, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        return acc

    cal = {}
    for sub, cname in (("cal_fetch", "FETCH_SIZE"), ("cal_write", "WRITE_SIZE")):
        for (k, c), a in counters(sub, "cal").items():
            if "calib" in k and c == cname:
                cal[(short(k), c)] = a[1] / a[0]
    print("\n== calibration: counter value per launch of a 1 GiB (1073741824 B) stream")
    scale = {}
    for (k, c), v in cal.items():
        is_write = "write" in k
        if (c == "FETCH_SIZE") == (not is_write):
            f = 1073741824.0 / v if v else float("nan")
            print("%-72s %-11s %14.1f  -> bytes per count %.2f" % (k, c, v, f))
            width = "int" if "<int>" in k else ("double2" if "double2" in k or "HIP_vector" in k else "double")
            scale[(c, width)] = f
    print("\n== bench kernels: counter per launch (and bytes with the 16 B / 8 B / 4 B calibration factors)")
    for sub, cname in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        for (k, c), a in counters(sub, "bench").items():
            if c != cname or a[0] < 50:
                continue
            v = a[1] / a[0]
            facs = ", ".join("%s: %.1f MB" % (w, v * scale[(c, w)] / 1e6) for w in ("double2", "double", "int")
                             if (c, w) in scale)
            print("%-72s %-11s n=%5d  %14.1f   [%s]" % (short(k), c, a[0], v, facs))


if __name__ == "__main__":
    main()
