#!/bin/bash
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/bw_variants.hip -o /tmp/bw 2>&1 | grep -E "error" -A5 | head
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/bw_variants.hip -c -o /tmp/bw.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|Occupancy" | paste - - - | sed 's/remark: [^ ]* //g; s/\[-Rpass[^]]*\]//g' | awk '{$1=$1};1' | cut -c1-200 | grep bitmask_r
for a in "$@"; do /tmp/bw $a; done
