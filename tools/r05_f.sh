#!/bin/bash
export TMPDIR=/tmp
T=${1:-r05h}
O=gpurun_out/$T
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
tail -3 $O/pytest.txt
: > $O/shapes.jsonl
for cs in 1 0; do for a in "--boxes 256 --graph" "--boxes 512 --graph" "--boxes 1024 --graph" "--boxes 2048 --graph" "--boxes 1024" "--boxes 512"; do
  GNMS_COUNT_SORT=$cs timeout 300 python bench.py $a --steps 200 --warmup 20 --no-cpu-baseline --no-other-kind 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('count_sort=$cs', d['config']['workload'][:44], 'graph' if d['config'].get('hip_graph_replay') else 'eager', d['ms_per_step'])" | tee -a $O/shapes.txt
done; done
bash tools/prof.sh ${T}_n1024 --boxes 1024 --steps 100 --warmup 10 --no-cpu-baseline --no-other-kind 2>&1 | head -6
