#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "n4096 or scale or leader_scan or recycled or empty or fuzz_layer" 2>&1 | tail -2
for cs in 1 0; do for a in "--batch 1 --graph" "--batch 2 --graph" "--batch 1"; do
GNMS_COUNT_SORT=$cs timeout 300 python bench.py $a --steps 200 --warmup 20 --no-cpu-baseline --no-other-kind 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('count_sort=$cs', d['config']['workload'][:40], 'graph' if d['config'].get('hip_graph_replay') else 'eager', d['ms_per_step'])"
done; done
