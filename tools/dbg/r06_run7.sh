cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for zc in 0 504 168 126 84; do
  echo "== MK_PENCIL_ZC=$zc"
  MK_PENCIL_ZC=$zc timeout 300 python tools/r06_march_sizes.py 500,500,500 v:500,500,500 2>&1 | sed 's/|.*fmt  *9/| fmt 9/; s/|.*fmt 11/| fmt 11/' | cut -c1-230
done > gpurun_out/r06g_zc.txt 2>&1
cat gpurun_out/r06g_zc.txt
