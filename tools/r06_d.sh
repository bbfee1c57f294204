#!/bin/bash
set -x
R=$PWD
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r06f_tests.txt
tail -8 gpurun_out/r06f_tests.txt
for n in 1024 256; do
timeout 300 tools/prof_cmd.sh r06f_b8_n$n python $R/bench.py --boxes $n --steps 300 --no-extras --no-cpu-baseline --no-other-kind > gpurun_out/r06f_b8_n${n}_stats.txt 2>&1
GNMS_ONE_LAUNCH=0 timeout 300 tools/prof_cmd.sh r06f_b8_n${n}_three python $R/bench.py --boxes $n --steps 300 --no-extras --no-cpu-baseline --no-other-kind > gpurun_out/r06f_b8_n${n}_three_stats.txt 2>&1
cat gpurun_out/r06f_b8_n${n}_stats.txt gpurun_out/r06f_b8_n${n}_three_stats.txt
done
timeout 300 tools/prof_cmd.sh r06f_single python $R/tools/single_n500.py > gpurun_out/r06f_single_n500_stats.txt 2>&1
cat gpurun_out/r06f_single_n500_stats.txt
for i in 1 2; do
timeout 300 python tools/small_n.py > gpurun_out/r06f_small_n_$i.jsonl 2>&1
GNMS_ONE_LAUNCH=0 timeout 300 python tools/small_n.py > gpurun_out/r06f_small_n_three_$i.jsonl 2>&1
done
tail -n 7 gpurun_out/r06f_small_n_1.jsonl gpurun_out/r06f_small_n_three_1.jsonl gpurun_out/r06f_small_n_2.jsonl gpurun_out/r06f_small_n_three_2.jsonl
