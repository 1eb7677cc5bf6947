#!/bin/bash
# kernel-trace of a long CG run: distribution and time series of the per-kernel durations (is the run-to-run spread a
# clock / power state that changes over time inside one process?)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/trace_series; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -f csv -d $OUT/t -o b -- python $R/bench.py --steps 3000 --warmup 50 --no-cpu --no-extra --spmv-launches 0 > $OUT/bench.json 2> $OUT/err.txt
cd $R
python - "$OUT" <<'PY'
import csv, glob, sys, collections
import numpy as np
out = sys.argv[1]
rows = []
for f in glob.glob(out + '/t/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
series = collections.OrderedDict()
for r in rows:
    k = r['Kernel_Name']
    name = 'spmv' if 'CgSpmvEpi' in k and 'Partial' not in k else ('updR' if 'CgUpdateR' in k else ('updXP' if 'CgUpdateXP' in k else None))
    if not name: continue
    series.setdefault(name, []).append(((int(r['Start_Timestamp']) - t0) / 1e9, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
lines = []
for name, s in series.items():
    t = np.array([a for a, _ in s]); d = np.array([b for _, b in s])
    lines.append('%s: n %d  mean %.1f  p5 %.1f  p25 %.1f  p50 %.1f  p75 %.1f  p95 %.1f us' % ((name, len(d), d.mean()) + tuple(np.percentile(d, [5, 25, 50, 75, 95]))))
    # time series in 20 bins
    edges = np.linspace(t.min(), t.max() + 1e-9, 21)
    bins = ['%.0f' % d[(t >= edges[i]) & (t < edges[i + 1])].mean() if ((t >= edges[i]) & (t < edges[i + 1])).any() else '-' for i in range(20)]
    lines.append('   mean per 1/20 of the run (%.1f s): %s' % (t.max() - t.min(), ' '.join(bins)))
open(out + '/summary.txt', 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
PY
