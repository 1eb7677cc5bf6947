#!/bin/bash
# PMC counters (own runs, no tracing) of the CG kernels for a few launch variants.
# usage: tools/pmc_quick.sh <outdir-name> <label>:<ENV=val,...>:<workload> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUTN=$1; shift
OUT=$R/gpurun_out/$OUTN; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SETS=("FETCH_SIZE" "WRITE_SIZE" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
      "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES" \
      "GRBM_GUI_ACTIVE TA_TA_BUSY_sum")
for spec in "$@"; do
  label=${spec%%:*}; rest=${spec#*:}; envs=${rest%%:*}; wl=${rest#*:}
  IFS=',' read -ra EV <<< "$envs"
  i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    env "${EV[@]}" rocprofv3 --pmc $set -f csv -d $OUT/${label}_p$i -o b -- python $R/bench.py --workload $wl --steps 20 --warmup 5 --no-cpu --no-extra --spmv-launches 10 > /dev/null 2> $OUT/${label}_p$i.err
  done
done
cd $R
python - "$OUT" <<'PY'
import csv, glob, collections, os, re, sys
out = sys.argv[1]
def short(name):
    name = name.replace("(anonymous namespace)::", "")
    m = re.match(r"(?:void )?(mk_\w+_kernel<[^(]*>)\(", name)
    return (m.group(1).replace(" ", "") if m else name.split("(")[0])[:70]
labels = sorted({os.path.basename(d).rsplit('_p', 1)[0] for d in glob.glob(out + '/*_p*') if os.path.isdir(d)})
lines = []
for lab in labels:
    acc = collections.OrderedDict()
    for f in glob.glob(out + '/' + lab + '_p*/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            if not ('CgSpmvEpi' in k or 'CgUpdate' in k): continue
            if 'MkPartialOf' in k: continue
            a = acc.setdefault((k, r['Counter_Name']), [0, 0.0]); a[0] += 1; a[1] += float(r['Counter_Value'])
    lines.append('==== ' + lab)
    kern = collections.OrderedDict()
    for (k, c), a in acc.items(): kern.setdefault(k, {})[c] = a[1] / a[0]
    for k, cs in kern.items():
        lines.append('  ' + k)
        f2, w2 = cs.get('FETCH_SIZE'), cs.get('WRITE_SIZE')
        if f2 is not None: lines.append('     FETCH_SIZE %.0f KB -> x2048 B = %.3f GB (gfx950: 2 x FETCH_SIZE KB)' % (f2, f2 * 2048 / 1e9))
        if w2 is not None: lines.append('     WRITE_SIZE %.0f KB -> x1024 B = %.3f GB' % (w2, w2 * 1024 / 1e9))
        if 'TCC_HIT_sum' in cs: lines.append('     L2 hit %.1f %%  (req %.3g, EA0_RDREQ %.3g)' % (100 * cs['TCC_HIT_sum'] / max(1, cs['TCC_HIT_sum'] + cs['TCC_MISS_sum']), cs['TCC_REQ_sum'], cs['TCC_EA0_RDREQ_sum']))
        if 'SQ_WAVE_CYCLES' in cs: lines.append('     waves waiting %.1f %%, issue-stalled %.1f %%; VMEM_RD %.3g VALU %.3g SALU %.3g LDS %.3g insts' % (100 * cs['SQ_WAIT_ANY'] / cs['SQ_WAVE_CYCLES'], 100 * cs['SQ_WAIT_INST_ANY'] / cs['SQ_WAVE_CYCLES'], cs['SQ_INSTS_VMEM_RD'], cs['SQ_INSTS_VALU'], cs['SQ_INSTS_SALU'], cs['SQ_INSTS_LDS']))
        if 'GRBM_GUI_ACTIVE' in cs: lines.append('     GUI_ACTIVE %.3g cycles, TA busy (sum) %.3g' % (cs['GRBM_GUI_ACTIVE'], cs.get('TA_TA_BUSY_sum', 0)))
open(out + '/summary.txt', 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
PY
