#!/bin/bash
# A/B on one box: variants of the plain row body (GNMS_FAST_ROWS: 0 round 2's body, 1 default, 1x vmcnt throttle x, 20/22 ordinary stores) by writer count
for fr in ${FR:-0 1}; do for w in ${@:-0 232 216 200 184 168}; do
  for kind in ${KINDS:-clustered uniform}; do
  echo -n "fast_rows=$fr writers=$w $kind "
  GNMS_FAST_ROWS=$fr GNMS_TAIL_WRITERS=$w python bench.py --steps 100 --warmup 5 --kind $kind --no-other-kind --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(d['ms_per_step'], 'launch GB/s', r['achieved'], r['frac'])
"
  done
done; done
