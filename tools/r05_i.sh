#!/bin/bash
export TMPDIR=/tmp
for v in varA varB; do
GNMS_BINDING=ctypes GNMS_LIB_PATH=build/$v/libgroomed_nms_hip.so timeout 300 python bench.py --two-calls --steps 100 --warmup 10 --no-cpu-baseline --no-other-kind 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v two_calls', d['ms_per_step'], r['kernel'], r['kernel_ms'], r['frac'])"
done
