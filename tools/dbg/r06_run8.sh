cd $GRAFT_REPO_ROOT
STEPS=2000 timeout 300 python tools/r06_march_sizes.py 1000,0,0 1448,0,0 100,100,100 128,128,128 200,200,200 2>&1 | cut -c1-260 > gpurun_out/r06h_small.txt
cat gpurun_out/r06h_small.txt
