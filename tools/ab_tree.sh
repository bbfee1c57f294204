#!/bin/bash
# A/B on one box: this tree against the copy of an earlier commit under ab_old/ (built there; not tracked)
for rep in 1 2; do for t in . ab_old; do for kind in clustered uniform; do
  echo -n "tree=$t $kind "
  (cd $t && python bench.py --steps 100 --warmup 5 --kind $kind --no-other-kind --no-extras --no-cpu-baseline "$@" 2>/dev/null) | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(d['ms_per_step'], 'launch GB/s', r['achieved'], r['frac'])
"
done; done; done
