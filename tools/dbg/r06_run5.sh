cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 2400 python tools/r06_placement_offsets.py 512 ) > gpurun_out/r06e_placement_offsets.txt 2>&1
tail -50 gpurun_out/r06e_placement_offsets.txt
( time timeout 3000 bash tools/profile_r06.sh r06a ) > gpurun_out/r06e_profile.log 2>&1
tail -20 gpurun_out/r06e_profile.log
