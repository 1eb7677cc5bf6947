cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -m gpu -q ) > gpurun_out/r06d_gputests.log 2>&1
tail -8 gpurun_out/r06d_gputests.log
timeout 900 bash tools/r06_slab_budget.sh > gpurun_out/r06d_slab.log 2>&1
grep -E "^==|CgFused|kernels per pass" gpurun_out/r06d_slab.log
