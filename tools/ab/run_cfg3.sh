run() { tag=$1; shift; env "$@" python bench.py --workload poisson2d-1000 --all-configs --steps 300 --warmup 30 --no-cpu 2>/dev/null > gpurun_out/c3_$tag.json; }
rm -f gpurun_out/c3_*.json
for u in 1 2; do for k in 4 6 8; do for g in 1536 2048; do run u${u}k${k}g$g MK_RT_U=$u MK_RT_PHASES=$k MK_GRID_SPMV=$g; done; done; done
