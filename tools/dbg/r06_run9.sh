cd $GRAFT_REPO_ROOT
STEPS=600 timeout 600 python tools/r06_march_sizes.py 256,256,128 200,200,200 256,256,256 250,250,250 300,300,300 384,384,384 400,400,400 v:200,200,200 v:250,250,250 v:300,300,300 2>&1 | cut -c1-260 > gpurun_out/r06i_mid.txt
echo "== MK_PEN_GEN=1" >> gpurun_out/r06i_mid.txt
MK_PEN_GEN=1 STEPS=600 timeout 600 python tools/r06_march_sizes.py 256,256,128 256,256,256 2>&1 | cut -c1-260 >> gpurun_out/r06i_mid.txt
cat gpurun_out/r06i_mid.txt
