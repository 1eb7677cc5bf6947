#!/bin/bash
# round 6: per-kernel times of rank 3's 512^3 / 8 slab (tools/slab_budget.py: production partitioning path, loopback transport)
# under rocprofv3 --kernel-trace, fused CG passes on the brick march:
#   fused_r5   MK_PEN_TAIL_GEN=0   the slab's last 4 planes one unpipelined plane after the other (round 5)
#   fused      (default)           ... as one masked round of the general-geometry kernel (round 6)
# Writes gpurun_out/slab_budget6/summary.txt and budget.json (copied to profiles/r06_slab_budget.{txt,json}; bench.py reads the
# json for the N = 8 line's per-rank budget).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/slab_budget6; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kind in const varcoef; do
  for mode in fused_r5 fused; do
    case $mode in
      fused_r5) E="MK_PEN_TAIL_GEN=0";;
      fused)    E="MK_DUMMY=1";;
    esac
    env $E rocprofv3 --kernel-trace --stats -f csv -d $OUT/$kind.$mode -o s -- python $R/tools/slab_budget.py $kind 200 > $OUT/$kind.$mode.json 2> $OUT/$kind.$mode.err
  done
done
cd $R
python - "$OUT" <<'PY'
import csv, glob, json, sys, re, collections
out = sys.argv[1]
lines, budget = [], {}
def short(name):
    name = name.replace("(anonymous namespace)::", "")
    m = re.match(r"(?:void )?(mk_\w+_kernel<[^(]*>)\(", name)
    return (m.group(1).replace(" ", "") if m else name.split("(")[0])[:72]
for kind in ("const", "varcoef"):
    for mode in ("fused_r5", "fused"):
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
        b = {}
        fusedk = sorted(((nm, a) for nm, a in agg.items() if "CgFusedEpiT" in nm and a[0] >= 100), key=lambda kv: -kv[1][1])
        for nm, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            if a[0] < 100: continue
            per_pass = a[1] / 200.0 / 1e3
            tot += per_pass
            lines.append("   %-92s %5.2f launches/pass  avg %8.1f us  -> %8.1f us per pass" % (nm, a[0] / 200.0, a[1] / a[0] / 1e3, per_pass))
            if "CgUpdateR" in nm: b["update_r_us"] = round(per_pass, 1)
            if "cg_beta" in nm: b["scalar_us"] = round(per_pass, 1)
            if "pack_kernel" in nm: b["pack_us"] = round(per_pass, 1)
        if len(fusedk) >= 2:
            b["fused_interior_us"] = round(fusedk[0][1][1] / 200.0 / 1e3, 1)
            b["fused_boundary_us"] = round(fusedk[1][1][1] / 200.0 / 1e3, 1)
            b["boundary_kernel"] = fusedk[1][0]
        b["kernels_per_pass_us"] = round(tot, 1)
        lines.append("   kernels per pass: %.1f us  (+ 2 all-reduces of 16 KiB and the halo messages, not measured here)" % tot)
        if mode == "fused":
            budget[kind] = b
        else:
            budget[kind + "_round5_tail"] = b
budget["source"] = "tools/r06_slab_budget.sh: rank 3's slab of 512^3 / 8 alone on one MI355X, rocprofv3 kernel trace, 200 passes"
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
json.dump(budget, open(out + "/budget.json", "w"), indent=1)
print("\n".join(lines))
PY
