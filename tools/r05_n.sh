#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/profiles_r05f; mkdir -p $O; T=r05f
: > $O/${T}_phase_ticks.txt
for cfg in "1 4096 uniform" "8 4096 uniform" "8 1024 uniform" "8 512 uniform"; do set -- $cfg; echo "== B=$1 N=$2 $3 (lists)" >> $O/${T}_phase_ticks.txt; GNMS_LIB_PATH=build/timing/libgroomed_nms_hip.so GNMS_BINDING=ctypes timeout 300 python tools/phase_ticks.py --batch $1 --boxes $2 --kind $3 --lists 2>&1 | grep -v amdgpu.ids >> $O/${T}_phase_ticks.txt; done
head -20 $O/${T}_phase_ticks.txt
