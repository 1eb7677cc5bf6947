#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_spmv; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for set in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set -f csv -d $OUT/p$i -o s -- $R/tools/ubench/spmv_bench 512 512 128 1 > /dev/null 2>$OUT/p$i.err
done
python - <<'PY'
import csv,glob,collections,os
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc_spmv'
acc=collections.OrderedDict()
for f in sorted(glob.glob(out+'/p*/**/*counter_collection.csv',recursive=True)):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'spmv' not in k: continue
        key=(k[:60], r['Grid_Size'], r['Counter_Name'])
        a=acc.setdefault(key,[0,0.0]); a[0]+=1; a[1]+=float(r['Counter_Value'])
last=None
for (k,g,c),a in acc.items():
    if (k,g)!=last: print('\n==',k,'grid',g); last=(k,g)
    print('   %-40s %16.0f'%(c,a[1]/a[0]))
PY
