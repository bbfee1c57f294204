#!/bin/bash
# round-5 session script B: GPU tests, phase ticks (timing build), kernel stats of B=1 / B=8 / N=1024, bench shapes.  $1 = output tag
export TMPDIR=/tmp
T=${1:-r05c}
O=gpurun_out/$T
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
tail -3 $O/pytest.txt
for cfg in "1 4096 uniform" "8 4096 uniform" "8 1024 uniform" "8 512 uniform"; do
  set -- $cfg
  echo "== B=$1 N=$2 $3 (lists)" >> $O/ticks.txt
  GNMS_LIB_PATH=build/timing/libgroomed_nms_hip.so GNMS_BINDING=ctypes timeout 300 python tools/phase_ticks.py --batch $1 --boxes $2 --kind $3 --lists 2>&1 | grep -v amdgpu.ids >> $O/ticks.txt
done
cat $O/ticks.txt
for cfg in "b1:--batch 1" "b8:--batch 8" "n1024:--boxes 1024" "tc:--two-calls"; do
  tag=${cfg%%:*}; a=${cfg#*:}
  bash tools/prof.sh ${T}_$tag $a --steps 100 --warmup 10 --no-cpu-baseline --no-other-kind > $O/prof_$tag.txt 2>&1
  cp gpurun_out/prof_${T}_$tag/bench_kernel_stats.csv $O/${tag}_kernel_stats.csv
  python - $O/${tag}_kernel_stats.csv <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
print(sys.argv[1])
for r in rows[:9]:
    n=re.sub(r"\(anonymous namespace\)::|gnms::|void ","",r["Name"])[:60]
    print("  %-60s calls %5s avg %8.1f us  %5s%%"%(n,r["Calls"],float(r["AverageNs"])/1e3,r["Percentage"]))
PY
done
: > $O/shapes.jsonl
for a in "--batch 1" "--batch 4" "--batch 8" "--boxes 1024" "--boxes 256" "--boxes 512" "--two-calls" "--batch 1 --graph" "--boxes 1024 --graph" "--boxes 512 --graph"; do
  timeout 300 python bench.py $a --steps 100 --warmup 10 --no-cpu-baseline --no-other-kind 2>/dev/null | tail -1 >> $O/shapes.jsonl
done
python - $O/shapes.jsonl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.strip():
        d=json.loads(l); r=d.get('roofline') or {}
        print(d['config']['workload'][:50], 'graph' if d['config'].get('hip_graph_replay') else '', d['ms_per_step'], r.get('kernel'), r.get('kernel_ms'), r.get('frac'))
PY
python tools/tail_time.py > $O/tail_times.jsonl 2>/dev/null; cat $O/tail_times.jsonl
python tools/aploss_time.py > $O/aploss_times.jsonl 2>/dev/null; cat $O/aploss_times.jsonl
