cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python tools/r06_march_sizes.py > gpurun_out/r06b_sizes.log 2>&1
MK_PEN_GEN=1 timeout 600 python tools/r06_march_sizes.py 512,512,512 v:512,512,512 > gpurun_out/r06b_sizes_gen512.log 2>&1
timeout 600 python tools/r06_march_sizes.py 512,512,512 v:512,512,512 > gpurun_out/r06b_sizes_aligned512.log 2>&1
