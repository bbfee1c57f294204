#!/bin/bash
# which (B, N) the one launch of the one-call entry wins: kernel times under the tracer, both paths
R=$PWD
for cfg in "8 512" "8 768" "4 1024" "2 1024" "1 1024" "1 500" "2 500" "16 256"; do
set -- $cfg
for mode in 1 0; do
GNMS_ONE_LAUNCH=$mode timeout 300 tools/prof_cmd.sh r06g_b$1_n$2_m$mode python $R/bench.py --batch $1 --boxes $2 --steps 200 --no-extras --no-cpu-baseline --no-other-kind > gpurun_out/r06g_b$1_n$2_m$mode.txt 2>&1
echo "== B=$1 N=$2 one_launch=$mode"; grep "one_launch\|tail_write\|bitmask_boxes\|sort_count\|sort_runs\|sort_merge" gpurun_out/r06g_b$1_n$2_m$mode.txt | cut -c1-40,100-160
done
done
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bench_line" 2>&1 | tail -30 > gpurun_out/r06g_tests.txt
tail -30 gpurun_out/r06g_tests.txt
