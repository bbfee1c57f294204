#!/bin/bash
# usage: tools/kres.sh <object file> <substring>   -- VGPRs / spills / LDS / scratch of the gfx950 kernels of a built object, from its metadata notes
cd $(dirname $1) && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading $(basename $1) > /dev/null 2>&1
f=$(basename $1).0.hipv4-amdgcn-amd-amdhsa--gfx950
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $f | python3 -c "
import sys,re
txt=sys.stdin.read()
for blk in txt.split('  - .agpr_count')[1:]:
    m=re.search(r'\.name:\s+(\S+)',blk)
    if m and '$2' in m.group(1):
        g=lambda k: re.search(k+r':\s+(\S+)',blk).group(1)
        print(m.group(1)[:70], 'vgpr',g('.vgpr_count'),'spill',g('.vgpr_spill_count'),'lds',g('.group_segment_fixed_size'),'scratch',g('.private_segment_fixed_size'))
"
rm -f $(basename $1).0.*
