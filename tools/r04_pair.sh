#!/bin/bash
O=gpurun_out/r4g; mkdir -p $O
cat > /tmp/chk.py <<'PY'
import sys, hashlib, numpy as np
sys.path.insert(0, '.')
from pykrylov_amd import _lib, gallery
import bench
from pykrylov_amd.linop import CsrOperator
_lib.init(0)
op = gallery.random_diagdom(1000000, seed=1)
x = np.random.default_rng(5).standard_normal(1000000)
y = op * x
print("cfg3", hashlib.sha1(y.tobytes()).hexdigest(), bench.format_info(_lib.init(), op)["format"])
ip, ix, dv = bench.random_tall_csr(700001, 100000*11, 5, 11)
A = CsrOperator(ip, ix, dv, (700001, 1100000))
x = np.random.default_rng(6).standard_normal(1100000)
print("tall", hashlib.sha1((A * x).tobytes()).hexdigest(), bench.format_info(_lib.init(), A)["format"])
PY
MK_SPMV_FORMAT=0 python /tmp/chk.py > $O/chk_fmt0.txt 2>&1
MK_RT_REG=0 python /tmp/chk.py > $O/chk_lds.txt 2>&1
python /tmp/chk.py > $O/chk_pair.txt 2>&1
tail -n 3 $O/chk_*.txt
export BENCH_ONLY_LOOPS=bicgstab,cgs,tfqmr,lsqr
for v in lds pair lds pair; do
  case $v in
    lds) export MK_RT_REG=0;;
    pair) unset MK_RT_REG;;
  esac
  python bench.py --only-other-configs > $O/oc_$v.json 2> $O/oc_$v.err
  python - <<PY
import json
d=json.loads(open('$O/oc_$v.json').read().strip().splitlines()[-1])
for k,e in d.items():
    print('$v', k, round(e['value'],1), {a: round(b['avg_product_us'],1) for a,b in e['products'].items()}, e['format']['grid'])
PY
done
