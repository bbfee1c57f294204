#!/bin/bash
# usage: tools/ab.sh ENVVAR "v1 v2 ..." [bench args...]   -- bench.py under each value of one environment switch (ms/step, dominant kernel)
var=$1; vals=$2; shift 2
for v in $vals; do
  env $var=$v python bench.py --no-cpu-baseline --no-other-kind "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline'] or {}
print('$var=$v', '$*', 'ms/step', d['ms_per_step'], '|', r.get('kernel'), r.get('kernel_ms'), 'GB/s', r.get('achieved'), 'ceiling', (r.get('ceiling') or {}).get('GB/s'))"
done
