#!/bin/bash
run() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['value'])"; }
for d in 2 3; do for n in 8192 16384; do for p in 0 10 20 30 40 50 60; do
GNMS_SPLIT_PCT=$p run "dim=$d N=$n pct=$p" --boxes $n --dim $d
done; done; done
