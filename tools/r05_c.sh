#!/bin/bash
# round-5 session script C: the 3D split (GNMS_3D_SPLIT_PCT sweep), 3D tests
export TMPDIR=/tmp
T=${1:-r05e}
O=gpurun_out/$T
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q -k "iou3d or switches or symmetric or 3d or empty" > $O/pytest3d.txt 2>&1; echo "pytest rc $?" >> $O/pytest3d.txt
tail -3 $O/pytest3d.txt
: > $O/split.jsonl
for n in 16384 8192; do for pct in 0 8 15 22 30; do for kind in uniform; do
  GNMS_3D_SPLIT_PCT=$pct GNMS_BENCH_PREWARM=10 timeout 600 python bench.py --dim 3 --boxes $n --kind $kind --steps 30 --warmup 5 --no-cpu-baseline --no-other-kind 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print(json.dumps({'N':$n,'pct':$pct,'kind':'$kind','ms':d['ms_per_step'],'kernel_ms':r.get('kernel_ms'),'launches':r.get('launches_per_step')}))" | tee -a $O/split.jsonl
done; done; done
timeout 300 python bench.py --two-calls --steps 100 --warmup 10 --no-cpu-baseline --no-other-kind 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('two_calls', d['ms_per_step'])"
