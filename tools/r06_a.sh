#!/bin/bash
# round 6, session a: the one launch of a small image -- parity A/B, kernel stats, the reference call's time
set -x
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_launch or recycled or capturable or detects_symmetry or random_vs_oracle or fuzz_layer or batched_ragged or lazy_index or counts" 2>&1 | tail -8 > gpurun_out/r06b_tests.txt
tail -5 gpurun_out/r06b_tests.txt
timeout 300 tools/prof_cmd.sh r06b_single python tools/single_n500.py > gpurun_out/r06b_single_n500_stats.txt 2>&1
GNMS_ONE_LAUNCH=0 timeout 300 tools/prof_cmd.sh r06b_single3 python tools/single_n500.py > gpurun_out/r06b_single_n500_three_stats.txt 2>&1
cat gpurun_out/r06b_single_n500_stats.txt gpurun_out/r06b_single_n500_three_stats.txt
for i in 1 2; do
timeout 300 python tools/small_n.py > gpurun_out/r06b_small_n_$i.jsonl 2>&1
GNMS_ONE_LAUNCH=0 timeout 300 python tools/small_n.py > gpurun_out/r06b_small_n_three_$i.jsonl 2>&1
done
tail -n 8 gpurun_out/r06b_small_n_1.jsonl gpurun_out/r06b_small_n_three_1.jsonl gpurun_out/r06b_small_n_2.jsonl gpurun_out/r06b_small_n_three_2.jsonl
