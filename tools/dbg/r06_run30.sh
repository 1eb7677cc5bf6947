cd $GRAFT_REPO_ROOT
( time timeout 2400 python -m pytest tests -m gpu -q --durations=12 ) > gpurun_out/r06zz_gputests.log 2>&1
tail -16 gpurun_out/r06zz_gputests.log
