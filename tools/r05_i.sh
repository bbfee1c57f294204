#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "from_boxes or adversarial or n4096 or fuzz_layer or scale or one_call" 2>&1 | tail -2
GNMS_LIB_PATH=build/timing/libgroomed_nms_hip.so GNMS_BINDING=ctypes python tools/bits_ticks.py 2>&1 | grep -v amdgpu
bash tools/prof.sh r05l_b8 --steps 200 --warmup 20 --no-cpu-baseline --no-other-kind 2>&1 | head -6
bash tools/prof.sh r05l_b4 --batch 4 --steps 200 --warmup 20 --no-cpu-baseline --no-other-kind 2>&1 | head -4
