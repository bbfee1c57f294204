#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/sweep_fused.txt
: > $O
run() { echo "== $* $EXTRA" >> $O; env "$@" python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-other-kind $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline'] or {}
print(d['config']['workload'][:40], 'ms/step', d['ms_per_step'], 'boxes/s %.3e' % d['value'], '|', (r.get('kernel') or '')[-20:], r.get('kernel_ms'), 'frac', r.get('frac'))" >> $O; }
EXTRA="" run A=1
EXTRA="--dim 3" run A=1
EXTRA="--dim 3 --kind uniform" run A=1
EXTRA="--dim 3 --boxes 16384 --steps 30" run A=1
EXTRA="--boxes 16384 --steps 30" run A=1
cat $O
