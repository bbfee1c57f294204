#!/bin/bash
FLAGS=${MB_FLAGS:--DGNMS_TIMING}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off $FLAGS -I groomed_nms_amd/csrc tools/microbench.hip -o /tmp/microbench -L groomed_nms_amd -lgroomed_nms_hip -Wl,-rpath,$PWD/groomed_nms_amd 2>&1 | grep -E "error" -A5 | head -20
for args in "$@"; do /tmp/microbench $args 2>&1 | grep -v "warning\|RAND_MAX\|rand()"; done
