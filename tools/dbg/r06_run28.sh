cd $GRAFT_REPO_ROOT
SZ="200,200,200 300,300,300 400,400,400 528,528,528 v:300,300,300 v:400,400,400"
for w in auto 128 64 32; do
  echo "== MK_PEN_W=$w"
  if [ $w = auto ]; then E="MK_X=1"; else E="MK_PEN_W=$w"; fi
  env $E STEPS=400 timeout 900 python tools/r06_march_sizes.py $SZ 2>&1 | cut -c1-260 | sed 's/.*|  fmt/   | fmt/'
done > gpurun_out/r06z_shapes.txt 2>&1
cat gpurun_out/r06z_shapes.txt
