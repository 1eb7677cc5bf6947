#!/bin/bash
# instruction / wait counters of the ubench w7 kernel and the production fmt 2 kernel on the same box
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_ab; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "FETCH_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set -f csv -d $OUT/u$i -o s -- $R/tools/ubench/spmv_win2 512 512 256 2 > /dev/null 2>$OUT/u$i.err
  MK_GRID_SPMV=1024 rocprofv3 --pmc $set -f csv -d $OUT/p$i -o s -- python $R/bench.py --workload poisson3d-512 --steps 20 --warmup 5 --no-cpu --no-extra --spmv-launches 5 > /dev/null 2>$OUT/p$i.err
done
python3 - <<'PY'
import csv,glob,collections,os
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc_ab'
acc=collections.OrderedDict()
for f in sorted(glob.glob(out+'/*/**/*counter_collection.csv',recursive=True)):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if not (('spmv_w7' in k and r['Grid_Size']=='262144') or ('CgSpmvEpi' in k)): continue
        key=(k[:40], r['Counter_Name'])
        a=acc.setdefault(key,[0,0.0]); a[0]+=1; a[1]+=float(r['Counter_Value'])
names=sorted({k for k,_ in acc})
cs=sorted({c for _,c in acc})
print('%-30s'%''+''.join('%22s'%n[:21] for n in names))
for c in cs:
    print('%-30s'%c+''.join('%22.4g'%(acc[(n,c)][1]/acc[(n,c)][0]) if (n,c) in acc else '%22s'%'-' for n in names))
PY
