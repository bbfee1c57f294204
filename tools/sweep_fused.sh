#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/sweep_fused.txt
: > $O
run() { echo "== $* $EXTRA" >> $O; env "$@" python bench.py --warmup 5 --no-cpu-baseline --no-other-kind $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline'] or {}
print(d['config']['workload'][:40], 'ms/step', d['ms_per_step'], 'boxes/s %.3e' % d['value'], '|', (r.get('kernel') or '')[-20:], r.get('kernel_ms'), 'frac', r.get('frac'))" >> $O; }
for o in 0 1; do
EXTRA="--dim 3 --steps 100" run GNMS_3D_BITS_IN_WRITE=$o
EXTRA="--dim 3 --steps 100 --kind uniform" run GNMS_3D_BITS_IN_WRITE=$o
EXTRA="--dim 3 --boxes 1024 --steps 100" run GNMS_3D_BITS_IN_WRITE=$o
EXTRA="--dim 3 --boxes 2048 --steps 100" run GNMS_3D_BITS_IN_WRITE=$o
EXTRA="--dim 3 --boxes 8192 --steps 50" run GNMS_3D_BITS_IN_WRITE=$o
EXTRA="--dim 3 --boxes 16384 --steps 30" run GNMS_3D_BITS_IN_WRITE=$o
EXTRA="--dim 3 --boxes 16384 --steps 30 --kind uniform" run GNMS_3D_BITS_IN_WRITE=$o
done
cat $O
