#!/bin/bash
run() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['value'])"; }
for d in 2 3; do for n in 8192 16384; do
GNMS_TWO_STREAMS=0 run "one dim=$d N=$n" --boxes $n --dim $d
run "default dim=$d N=$n" --boxes $n --dim $d
done; done
run "N=4096 2D" --steps 200
run "N=4096 3D" --steps 200 --dim 3
