#!/bin/bash
# round-5 session script A: GPU tests, phase ticks of the chain (timing build), a few bench shapes
export TMPDIR=/tmp
O=gpurun_out/r05b
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
tail -5 $O/pytest.txt
for cfg in "1 4096 uniform" "8 4096 uniform" "8 4096 clustered" "8 1024 uniform" "8 512 uniform" "4 4096 uniform"; do
  set -- $cfg
  echo "== B=$1 N=$2 $3 (lists)" >> $O/ticks.txt
  GNMS_LIB_PATH=build/timing/libgroomed_nms_hip.so GNMS_BINDING=ctypes timeout 300 python tools/phase_ticks.py --batch $1 --boxes $2 --kind $3 --lists >> $O/ticks.txt 2>&1
done
echo "== two-calls B=8 N=4096 uniform (lists)" >> $O/ticks.txt
GNMS_LIB_PATH=build/timing/libgroomed_nms_hip.so GNMS_BINDING=ctypes timeout 300 python tools/phase_ticks.py --batch 8 --boxes 4096 --kind uniform --lists --two-calls >> $O/ticks.txt 2>&1
cat $O/ticks.txt
: > $O/shapes.jsonl
for a in "--batch 1" "--batch 4" "--batch 8" "--boxes 1024" "--boxes 256" "--boxes 512" "--two-calls" "--batch 1 --graph" "--boxes 1024 --graph" "--boxes 512 --graph"; do
  timeout 300 python bench.py $a --steps 100 --warmup 10 --no-cpu-baseline --no-other-kind 2>/dev/null | tail -1 >> $O/shapes.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r05b/shapes.jsonl'):
    if l.strip():
        d=json.loads(l); r=d.get('roofline') or {}
        print(d['config']['workload'][:50], 'graph' if d['config'].get('hip_graph_replay') else '', d['ms_per_step'], r.get('kernel'), r.get('kernel_ms'), r.get('frac'))
PY
