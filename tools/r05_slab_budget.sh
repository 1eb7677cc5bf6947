#!/bin/bash
# round 5: per-kernel times of rank 3's 512^3 / 8 slab (tools/slab_budget.py: production partitioning path, loopback
# transport) under rocprofv3 --kernel-trace, for the three ways a rank can run a CG pass now:
#   windowed  MK_SPMV_FORMAT=5   formats 4 / 5, three kernels per pass (rounds 3-4: what every rank ran)
#   march     MK_CG_FUSE=0       formats 9 / 10 on the slab (neighbours' planes from the received entries), three kernels
#   fused     (default)          the march with fused passes: r's boundary planes travel, p's are formed on the spot
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/slab_budget5; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kind in const varcoef; do
  for mode in windowed march fused; do
    case $mode in
      windowed) E="MK_SPMV_FORMAT=5";;
      march)    E="MK_CG_FUSE=0";;
      fused)    E="MK_DUMMY=1";;
    esac
    env $E rocprofv3 --kernel-trace --stats -f csv -d $OUT/$kind.$mode -o s -- python $R/tools/slab_budget.py $kind 200 > $OUT/$kind.$mode.json 2> $OUT/$kind.$mode.err
  done
done
cd $R
python - "$OUT" <<'PY'
import csv, glob, json, sys, re, collections
out = sys.argv[1]
lines = []
def short(name):
    name = name.replace("(anonymous namespace)::", "")
    m = re.match(r"(?:void )?(mk_\w+_kernel<[^(]*>)\(", name)
    return (m.group(1).replace(" ", "") if m else name.split("(")[0])[:72]
for kind in ("const", "varcoef"):
    for mode in ("windowed", "march", "fused"):
        tag = "%s.%s" % (kind, mode)
        try:
            info = json.loads(open("%s/%s.json" % (out, tag)).read().strip().splitlines()[-1])
        except Exception as e:
            lines.append("%s: FAILED %r" % (tag, e)); continue
        lines.append("== rank 3 of 8, 512^3 %s, %s: format %d, wall %.1f us per pass through the host-staged loopback" %
                     (kind, mode, info["format"], 1e3 * info["wall_ms_per_pass_with_host_staged_loopback"]))
        rows = []
        for f in glob.glob("%s/%s/**/*kernel_trace.csv" % (out, tag), recursive=True):
            rows += list(csv.DictReader(open(f)))
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        names = [short(r["Kernel_Name"]) + " grid=" + str(int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])) for r in rows]
        loop = [i for i, nm in enumerate(names) if "CgUpdateR" in nm]         # once per pass in every mode
        start = loop[-200] if len(loop) >= 200 else 0
        agg = collections.OrderedDict()
        for r, nm in list(zip(rows, names))[start:]:
            a = agg.setdefault(nm, [0, 0]); a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        tot = 0.0
        for nm, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            if a[0] < 100: continue
            per_pass = a[1] / 200.0 / 1e3
            tot += per_pass
            lines.append("   %-92s %5.2f launches/pass  avg %8.1f us  -> %8.1f us per pass" % (nm, a[0] / 200.0, a[1] / a[0] / 1e3, per_pass))
        lines.append("   kernels per pass: %.1f us  (+ 2 all-reduces of 16 KiB and the halo messages, not measured here)" % tot)
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
