#!/bin/bash
# headline step against the number of writer workgroups in tail_write_kernel (GNMS_TAIL_WRITERS)
for w in ${@:-64 96 128 160 192 224 240 248 256}; do
  for kind in clustered uniform; do
  echo -n "writers=$w $kind "
  GNMS_TAIL_WRITERS=$w python bench.py --steps 100 --warmup 5 --kind $kind --no-other-kind --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(d['ms_per_step'], 'launch GB/s', r['achieved'], r['frac'])
"
  done
done
