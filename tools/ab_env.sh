#!/bin/bash
# A/B on one box: headline step for values of one environment variable.  usage: tools/ab_env.sh VAR "v1 v2 ..." [bench args]
VAR=$1; VALS=$2; shift 2
for rep in 1 2; do for v in $VALS; do
  for kind in ${KINDS:-clustered uniform}; do
  echo -n "$VAR=$v $kind "
  env $VAR=$v python bench.py --steps 100 --warmup 5 --kind $kind --no-other-kind --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(d['ms_per_step'], 'launch GB/s', r['achieved'], r['frac'])
"
  done
done; done
