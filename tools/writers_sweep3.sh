#!/bin/bash
for w in ${@:-128 192 224 248}; do
  echo -n "3D writers=$w "
  GNMS_TAIL_WRITERS=$w python bench.py --steps 60 --warmup 5 --dim 3 --no-other-kind --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(d['ms_per_step'], 'launch GB/s', r['achieved'], r['frac'])
"
done
