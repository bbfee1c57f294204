#!/bin/bash
# Developer build with the s_memtime phase markers (GNMS_TIMING) in a library of its own: build/timing/libgroomed_nms_hip.so = nms_layer.hip
# recompiled with -DGNMS_TIMING + the production objects of the other translation units.  Use: GNMS_LIB_PATH=build/timing/libgroomed_nms_hip.so
# GNMS_BINDING=ctypes python tools/phase_ticks.py ...
set -e
cd "$(dirname "$0")/.."
mkdir -p build/timing
C=groomed_nms_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -DGNMS_TIMING $EXTRA -c $C/nms_layer.hip -o build/timing/nms_layer.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/timing/libgroomed_nms_hip.so build/timing/nms_layer.o $C/iou_kernels.o $C/soft_sort.o $C/classic_nms.o $C/nms_others.o $C/aploss.o $C/proposals.o $C/host_mailbox.o
ls -la build/timing/libgroomed_nms_hip.so
