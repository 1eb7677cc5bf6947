cd $GRAFT_REPO_ROOT
( time timeout 1500 python tools/r06_placement_offsets.py 512 big ) > gpurun_out/r06s_placement_big.txt 2>&1
cat gpurun_out/r06s_placement_big.txt
