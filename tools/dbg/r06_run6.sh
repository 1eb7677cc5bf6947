cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 600 ./tools/ubench/xcdpipe ) > gpurun_out/r06f_xcdpipe.txt 2>&1
cat gpurun_out/r06f_xcdpipe.txt
