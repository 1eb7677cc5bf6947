"""Registers / scratch / occupancy of the SpMV kernel instantiations of one translation unit.
   python tools/kernel_resources.py mk_cg.hip [template-FMT-value] [-DMACRO ...]"""
import re, subprocess, sys, os
src = sys.argv[1]
fmt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-D") else None
defs = [a for a in sys.argv[2:] if a.startswith("-D")]
here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pykrylov_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", *defs, "-c",
       os.path.join(here, src), "-o", "/tmp/_kr.o", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark: .*?\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).split()[0]] = int(m.group(2))
for name, r in rows.items():
    if "mk_spmv_kernel" not in name:
        continue
    m = re.search(r"Lb([01])ELi(\d+)EEv", name)
    if not m or (fmt and m.group(2) != fmt):
        continue
    dem = subprocess.run(["c++filt", name], stdout=subprocess.PIPE, text=True).stdout.strip()
    dem = re.sub(r"^void mk_spmv_kernel<|>\(MkCsrView.*$", "", dem).replace("(anonymous namespace)::", "")
    print("%-70s VGPR %3d  scratch %4d  occ %d  LDS %5d" % (dem[:70], r.get("VGPRs", -1), r.get("ScratchSize", -1), r.get("Occupancy", -1), r.get("LDS", -1)))
