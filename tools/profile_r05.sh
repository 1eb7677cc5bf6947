#!/bin/bash
# Round-5 profile of the production kernels on the GPU box (rocprofv3; counters in their own runs, no tracing):
#   trace_<w>   kernel trace + stats of bench.py per workload (the default line's primary workload and every extra)
#   pmc_<w>_i   one run per counter set on the same commands
#   cal_*       FETCH_SIZE / WRITE_SIZE calibration on 1 GiB streams
# usage: tools/profile_r05.sh [tag]        (results: gpurun_out/prof_<tag>/, summary.txt + spmv_traffic.json)
set -u
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/prof_$TAG          # (raw rocprofv3 output stays on the box: gpurun_out/ is capped at 64 MiB)
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
declare -A CMD
CMD[varcoef]="--workload poisson3d-512-varcoef --no-extra --no-cpu"
CMD[const]="--workload poisson3d-512 --no-extra --no-cpu"
CMD[p2d]="--workload poisson2d-1000 --no-extra --no-cpu"
CMD[others]="--only-other-configs"
CMD[s27c]="--workload stencil27-256 --no-extra --no-cpu"
CMD[s27v]="--workload stencil27-256-varcoef --no-extra --no-cpu"
for w in varcoef const p2d s27c s27v others; do
  steps="--steps 300 --warmup 20"; [ $w = p2d ] && steps="--steps 2000 --warmup 100"
  rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace_$w -o b -- python $R/bench.py ${CMD[$w]} $steps > $OUT/bench_trace_$w.json 2> $OUT/trace_$w.err
done
# the driver's command itself (one process, every workload): the judged kernel-stats file
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace_default -o b -- python $R/bench.py --no-cpu > $OUT/bench_trace_default.json 2> $OUT/trace_default.err
for w in varcoef const p2d s27c s27v others; do
  steps="--steps 40 --warmup 5 --spmv-launches 10"; [ $w = p2d ] && steps="--steps 300 --warmup 20 --spmv-launches 20"
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
             "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES" \
             "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES"; do
    i=$((i+1))
    rocprofv3 --pmc $set -f csv -d $OUT/pmc_${w}_$i -o b -- python $R/bench.py ${CMD[$w]} $steps > /dev/null 2> $OUT/pmc_${w}_$i.err
  done
done
rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/cal_1 -o cal -- python $R/tools/pmc_calibrate.py > $OUT/cal_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/cal_2 -o cal -- python $R/tools/pmc_calibrate.py > $OUT/cal_write.log 2>&1
cd $R
python tools/pmc_summary5.py $OUT > $OUT/summary.txt 2>&1
KEEP=$R/gpurun_out/prof_$TAG
rm -rf $KEEP; mkdir -p $KEEP
cp $OUT/summary.txt $OUT/spmv_traffic.json $OUT/bench_trace_*.json $KEEP/ 2>/dev/null
for d in $OUT/trace_*; do [ -d "$d" ] && for f in $(find $d -name "*kernel_stats.csv"); do cp $f $KEEP/$(basename $d)_kernel_stats.csv; done; done
cp $OUT/*.err $KEEP/ 2>/dev/null
tail -n 60 $OUT/summary.txt
