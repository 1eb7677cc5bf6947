#!/bin/bash
# per-kernel times of rank 3's 512^3 / 8 slab (tools/slab_budget.py) under rocprofv3 --kernel-trace
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/slab_budget; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kind in varcoef const; do
  rocprofv3 --kernel-trace --stats -f csv -d $OUT/$kind -o s -- python $R/tools/slab_budget.py $kind 200 > $OUT/$kind.json 2> $OUT/$kind.err
done
cd $R
python - "$OUT" <<'PY'
import csv, glob, json, sys, re, collections
out = sys.argv[1]
lines = []
def short(name):
    name = name.replace("(anonymous namespace)::", "")
    m = re.match(r"(?:void )?(mk_\w+_kernel<[^(]*>)\(", name)
    return (m.group(1).replace(" ", "") if m else name.split("(")[0])[:64]
for kind in ("varcoef", "const"):
    try:
        info = json.loads(open("%s/%s.json" % (out, kind)).read().strip().splitlines()[-1])
    except Exception as e:
        lines.append("%s: FAILED %r" % (kind, e)); continue
    lines.append("== rank 3 of 8, 512^3 %s: %s" % (kind, json.dumps(info)))
    rows = []
    for f in glob.glob("%s/%s/**/*kernel_trace.csv" % (out, kind), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last 200 passes only: per-kernel average + launches per pass
    agg = collections.OrderedDict()
    names = [short(r["Kernel_Name"]) + " grid=" + str(int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])) for r in rows]
    loop = [i for i, nm in enumerate(names) if "CgUpdateXP" in nm]
    start = loop[-200] if len(loop) >= 200 else 0
    for r, nm in list(zip(rows, names))[start:]:
        a = agg.setdefault(nm, [0, 0]); a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot = 0.0
    for nm, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        per_pass = a[1] / 200.0 / 1e3
        if a[0] < 100: continue
        tot += per_pass
        lines.append("   %-86s %6.2f launches/pass  avg %8.1f us  -> %8.1f us per pass" % (nm, a[0] / 200.0, a[1] / a[0] / 1e3, per_pass))
    lines.append("   kernels per pass: %.1f us  (+ 2 all-reduces of 16 KiB and the halo messages, not measured here)" % tot)
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
