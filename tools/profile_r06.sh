#!/bin/bash
# Round-6 profile of the production kernels on the GPU box (rocprofv3; counters in their own runs, no tracing):
#   trace_<w>   kernel trace + stats of bench.py per workload
#   pmc_<w>_i   one run per counter set on the same commands
#   cal_*       FETCH_SIZE / WRITE_SIZE calibration on 1 GiB streams
# workloads: const / varcoef (512^3, the line's two CG workloads), p500 (500^3: the general-geometry march), plain (512^3
# forced to storage format 0: north_star's literal CSR kernel), p2d (configs[1]), others (the nine other loops)
# usage: tools/profile_r06.sh [tag]        (results: gpurun_out/prof_<tag>/, summary.txt + spmv_traffic.json)
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/prof_$TAG          # (raw rocprofv3 output stays on the box: gpurun_out/ is capped at 64 MiB)
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
declare -A CMD
CMD[varcoef]="--workload poisson3d-512-varcoef --no-extra --no-cpu"
CMD[const]="--workload poisson3d-512 --no-extra --no-cpu"
CMD[p500]="--workload poisson3d-500 --no-extra --no-cpu --no-parity"
CMD[plain]="--workload poisson3d-512 --force-format 0 --no-extra --no-cpu --no-parity"
CMD[p2d]="--workload poisson2d-1000 --no-extra --no-cpu"
CMD[others]="--only-other-configs"
WL="varcoef const p500 plain p2d others"
for w in $WL; do
  steps="--steps 300 --warmup 20"; [ $w = p2d ] && steps="--steps 2000 --warmup 100"; [ $w = plain ] && steps="--steps 100 --warmup 10"
  rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace_$w -o b -- python $R/bench.py ${CMD[$w]} $steps > $OUT/bench_trace_$w.json 2> $OUT/trace_$w.err
done
# the driver's command itself (one process, every workload): the judged kernel-stats file
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace_default -o b -- python $R/bench.py --no-cpu > $OUT/bench_trace_default.json 2> $OUT/trace_default.err
for w in $WL; do
  steps="--steps 40 --warmup 5 --spmv-launches 10"; [ $w = p2d ] && steps="--steps 300 --warmup 20 --spmv-launches 20"
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
             "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES"; do
    i=$((i+1))
    rocprofv3 --pmc $set -f csv -d $OUT/pmc_${w}_$i -o b -- python $R/bench.py ${CMD[$w]} $steps > /dev/null 2> $OUT/pmc_${w}_$i.err
  done
done
rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/cal_1 -o cal -- python $R/tools/pmc_calibrate.py > $OUT/cal_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/cal_2 -o cal -- python $R/tools/pmc_calibrate.py > $OUT/cal_write.log 2>&1
cd $R
python tools/pmc_summary6.py $OUT > $OUT/summary.txt 2>&1
KEEP=$R/gpurun_out/prof_$TAG
rm -rf $KEEP; mkdir -p $KEEP
cp $OUT/summary.txt $OUT/spmv_traffic.json $OUT/bench_trace_*.json $KEEP/ 2>/dev/null
for d in $OUT/trace_*; do [ -d "$d" ] && for f in $(find $d -name "*kernel_stats.csv"); do cp $f $KEEP/$(basename $d)_kernel_stats.csv; done; done
cp $OUT/*.err $KEEP/ 2>/dev/null
tail -n 60 $OUT/summary.txt
