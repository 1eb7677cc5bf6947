cd $GRAFT_REPO_ROOT
SZ="200,200,200 250,250,250 300,300,300 400,400,400 v:200,200,200 v:300,300,300 v:400,400,400 2000,0,0 4000,0,0 64,64,2048 500,500,500"
echo "== linear bricks where the rule takes them (default)" > gpurun_out/r06n_lin.txt
STEPS=400 timeout 900 python tools/r06_march_sizes.py $SZ 2>&1 | cut -c1-260 >> gpurun_out/r06n_lin.txt
echo "== MK_PEN_LIN=0 (line bricks only)" >> gpurun_out/r06n_lin.txt
MK_PEN_LIN=0 STEPS=400 timeout 900 python tools/r06_march_sizes.py $SZ 2>&1 | cut -c1-260 | sed 's/.*|  fmt/   | fmt/' >> gpurun_out/r06n_lin.txt
echo "== MK_PEN_LIN=1 (linear bricks wherever L <= 512)" >> gpurun_out/r06n_lin.txt
MK_PEN_LIN=1 STEPS=400 timeout 900 python tools/r06_march_sizes.py 500,500,500 512,512,512 v:512,512,512 256,256,256 2>&1 | cut -c1-260 >> gpurun_out/r06n_lin.txt
cat gpurun_out/r06n_lin.txt
