#!/bin/bash
# PMC passes over the windowed-SpMV micro-benchmark (one rocprofv3 run per counter set, no tracing).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_win2; rm -rf $OUT; mkdir -p $OUT
ARGS=${ARGS:-"512 512 256 1"}
cd /tmp; export TMPDIR=/tmp
i=0
for set in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set -f csv -d $OUT/p$i -o s -- $R/tools/ubench/spmv_win2 $ARGS > $OUT/p$i.out 2>$OUT/p$i.err
done
python3 - <<'PY'
import csv,glob,collections,os
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc_win2'
acc=collections.OrderedDict()
for f in sorted(glob.glob(out+'/p*/**/*counter_collection.csv',recursive=True)):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'spmv' not in k: continue
        key=(k[:48], r['Grid_Size'], r['Counter_Name'])
        a=acc.setdefault(key,[0,0.0]); a[0]+=1; a[1]+=float(r['Counter_Value'])
last=None
with open(out+'/summary.txt','w') as fo:
    for (k,g,c),a in acc.items():
        if (k,g)!=last: print('\n==',k,'grid',g,file=fo); last=(k,g)
        print('   %-40s %16.0f'%(c,a[1]/a[0]),file=fo)
print(open(out+'/summary.txt').read())
PY
