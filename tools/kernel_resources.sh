#!/bin/bash
# VGPR / SGPR / LDS / occupancy of every kernel of one translation unit (gfx950), from the compiler's own remarks.
#   tools/kernel_resources.sh nms_layer.hip [filter]
src=groomed_nms_amd/csrc/$1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -c "$src" -o /dev/null \
    -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys, re
cur = None; rows = {}
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark: .*?\]?\s*(VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur: rows[cur][m.group(1).split()[0]] = int(m.group(2))
import subprocess
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\((anonymous namespace)\)::", "", name)
    name = re.sub(r"gnms::", "", name)
    name = re.sub(r"\((const |float|int|char|long|unsigned|gnms_).*", "", name)
    print("%-90s vgpr %3d agpr %3d sgpr %3d scratch %4d occ %2d lds %6d" % (name[:90], v.get("VGPRs",-1), v.get("AGPRs",-1), v.get("SGPRs",-1), v.get("ScratchSize",-1), v.get("Occupancy",-1), v.get("LDS",-1)))
' | grep -i "${2:-.}"
