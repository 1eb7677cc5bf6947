#!/bin/bash
# Round-2 profile of the production kernels on the GPU box (rocprofv3; counters in their own runs, no tracing):
#   trace_*   kernel trace + stats of `bench.py` (512^3 + 2-D n=1e6) and of `--all-configs` (BiCGSTAB random, MINRES)
#   pmc_*     one run per counter set on the same commands
#   cal_*     FETCH_SIZE / WRITE_SIZE calibration on 1 GiB streams
# usage: tools/profile_r02.sh [tag]        (results: gpurun_out/prof_<tag>/, summary.txt + spmv_traffic.json)
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
A3D="--steps ${STEPS3D:-100} --warmup 10 --no-cpu --no-extra --spmv-launches 20"
A2D="--workload poisson2d-1000 --steps 400 --warmup 20 --no-cpu --all-configs --spmv-launches 20"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace_3d -o b -- python $R/bench.py --steps 300 --warmup 20 --no-cpu --no-extra > $OUT/bench_trace_3d.json 2> $OUT/trace_3d.err
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace_2d -o b -- python $R/bench.py $A2D > $OUT/bench_trace_2d.json 2> $OUT/trace_2d.err
i=0
for set in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set -f csv -d $OUT/pmc3d_$i -o b -- python $R/bench.py $A3D > /dev/null 2> $OUT/pmc3d_$i.err
  rocprofv3 --pmc $set -f csv -d $OUT/pmc2d_$i -o b -- python $R/bench.py $A2D > /dev/null 2> $OUT/pmc2d_$i.err
done
rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/cal_fetch -o cal -- python $R/tools/pmc_calibrate.py > $OUT/cal_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/cal_write -o cal -- python $R/tools/pmc_calibrate.py > $OUT/cal_write.log 2>&1
cd $R
python tools/pmc_summary2.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
