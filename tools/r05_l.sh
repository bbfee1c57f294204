#!/bin/bash
export TMPDIR=/tmp
for n in 256 512; do
echo "== N=$n"; python tools/host_overhead.py --boxes $n --steps 1500 2>&1 | head -3
PYTHONPATH=$PWD bash tools/prof_cmd.sh r05l_n$n python $PWD/bench.py --boxes $n --steps 200 --warmup 20 --no-cpu-baseline --no-other-kind 2>&1 | head -8
done
