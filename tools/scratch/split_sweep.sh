#!/bin/bash
# matrix write on a CU-masked stream beside the per-image chain: ms/step of the default bench per (reserved CUs, mode)
run() { python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['value'])"; }
run base
for k in 8 16 32 64; do for m in 0 1 2 3; do GNMS_SPLIT_CUS=$k GNMS_SPLIT_MODE=$m run "k=$k mode=$m"; done; done
