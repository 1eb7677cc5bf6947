#!/bin/bash
# fmt 9 variants at 512^3: register budget / ring depth (variant libraries) and planes per chunk (MK_PENCIL_ZC)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5e; mkdir -p $O
L=$PWD/pykrylov_amd
for t in base occ3 occ4 r12 r12h4; do
  lib=$L/libmikrylov_$t.so; [ $t = base ] && lib=$L/libmikrylov.so
  echo "== $t" >> $O/variants.txt
  MIKRYLOV_LIB=$lib timeout 200 python tools/pencil_sizes.py 512,512,512 256,256,256 >> $O/variants.txt 2>&1
done
for zc in 24 48 66 96 132 258 516; do
  echo "== zc $zc" >> $O/variants.txt
  MK_PENCIL_ZC=$zc timeout 200 python tools/pencil_sizes.py 512,512,512 >> $O/variants.txt 2>&1
done
for g in 64,64,64 128,64,64 128,128,64 128,128,32 256,64,32; do
  timeout 100 python tools/pencil_sizes.py $g >> $O/small.txt 2>&1
done
cat $O/variants.txt $O/small.txt
