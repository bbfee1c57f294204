#!/bin/bash
run() { python bench.py --steps 200 --warmup 20 --no-cpu-baseline "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['value'])"; }
GNMS_TWO_STREAMS=0 run "N=4096 one stream"
for p in 15 25 35 45; do GNMS_NO_FENCE=1 GNMS_SPLIT_PCT=$p run "N=4096 nofence pct=$p"; GNMS_SPLIT_PCT=$p run "N=4096 fence pct=$p"; done
GNMS_NO_FENCE=1 run "N=16384 nofence" --boxes 16384 --steps 30;  run "N=16384 fence" --boxes 16384 --steps 30
GNMS_NO_FENCE=1 run "N=8192 nofence" --boxes 8192 --steps 30;  run "N=8192 fence" --boxes 8192 --steps 30
