#!/bin/bash
export TMPDIR=/tmp
T=${1:-r05k}
O=gpurun_out/$T
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
tail -3 $O/pytest.txt
for pipe in 1 0; do for k in uniform clustered; do
GNMS_PIPE=$pipe timeout 300 python bench.py --two-calls --kind $k --steps 100 --warmup 10 --no-cpu-baseline --no-other-kind 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('pipe=$pipe $k two_calls', d['ms_per_step'], r['kernel'], r['kernel_ms'], r['frac'])"
done; done
for a in "--two-calls --boxes 1024" "--two-calls --boxes 512" "--two-calls --batch 1"; do for pipe in 1 0; do
GNMS_PIPE=$pipe timeout 300 python bench.py $a --graph --steps 100 --warmup 10 --no-cpu-baseline --no-other-kind 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pipe=$pipe', d['config']['workload'][:40], 'graph', d['ms_per_step'])"
done; done
bash tools/prof.sh ${T}_tc --two-calls --steps 100 --warmup 10 --no-cpu-baseline --no-other-kind 2>&1 | head -7
