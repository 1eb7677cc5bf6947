#!/usr/bin/env python3
"""Summarise a rocprofv3 results database (kernel trace) as text: per-kernel calls / total / avg / share,
plus the average idle gap between consecutive kernels.  Usage: prof_summary.py results.db [title]"""
import collections
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
    rows = list(cur.execute("select name, start, end from kernels order by start"))
    agg = collections.OrderedDict()
    for name, s, e in rows:
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += e - s
        a[2] = min(a[2], e - s)
        a[3] = max(a[3], e - s)
    total = sum(a[1] for a in agg.values())
    print("# rocprofv3 --kernel-trace summary: %s" % title)
    print("# %d dispatches, %.3f ms of kernel time" % (len(rows), total / 1e6))
    print("%-96s %8s %12s %10s %10s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "share"))
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-96s %8d %12.1f %10.2f %10.2f %10.2f %6.2f%%" % (name[:96], a[0], a[1] / 1e3, a[1] / a[0] / 1e3,
                                                                a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / total))
    gaps = collections.defaultdict(list)
    for (n0, s0, e0), (n1, s1, e1) in zip(rows[:-1], rows[1:]):
        gaps[(n0[:60], n1[:60])].append(s1 - e0)
    print("\n# idle gap between consecutive dispatches (pairs seen > 20 times)")
    for (a, b), v in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
        if len(v) > 20:
            v.sort()
            print("%-62s -> %-62s n=%6d  median %.2f us  mean %.2f us" % (a, b, len(v), v[len(v) // 2] / 1e3,
                                                                          sum(v) / len(v) / 1e3))
    try:
        pm = list(cur.execute("select * from pmc_events limit 1"))
    except Exception:
        pm = []
    if pm:
        print("\n# PMC counters (sum over dispatches / per dispatch)")
        q = ("select k.name, p.counter_name, count(*), sum(p.value) from pmc_events p join kernels k "
             "on p.dispatch_id = k.dispatch_id group by k.name, p.counter_name")
        try:
            for name, cname, cnt, val in cur.execute(q):
                print("%-80s %-24s n=%6d per_dispatch=%.1f" % (name[:80], cname, cnt, val / cnt))
        except Exception as e:
            print("(could not join pmc_events: %s)" % e)


if __name__ == "__main__":
    main()
