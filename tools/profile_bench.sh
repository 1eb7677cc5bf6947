#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + two PMC passes + counter calibration.
# usage: tools/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps ${STEPS:-300} --warmup 20 --no-cpu --no-extra $*"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o bench -- python $R/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/bench_trace.err
rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch -o bench -- python $R/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/pmc_write -o bench -- python $R/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/bench_write.err
rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/cal_fetch -o cal -- python $R/tools/pmc_calibrate.py > $OUT/cal_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/cal_write -o cal -- python $R/tools/pmc_calibrate.py > $OUT/cal_write.log 2>&1
cd $R
find $OUT -name "*.csv" | head -40
python tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
