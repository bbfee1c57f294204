#!/bin/bash
R=$PWD
timeout 900 python -m pytest tests -x -q -m gpu -k "from_boxes or n4096 or at_scale or one_call or full_size or fuzz_layer or adversarial or library_switches or recycled or golden or smoke or batched" 2>&1 | tail -5 > gpurun_out/r06q_tests.txt; cat gpurun_out/r06q_tests.txt
timeout 300 tools/prof_cmd.sh r06q_bench python $R/bench.py --no-extras --no-cpu-baseline --no-other-kind > gpurun_out/r06q_bench_stats.txt 2>&1
head -7 gpurun_out/r06q_bench_stats.txt
for i in 1 2 3; do python bench.py --no-extras --no-cpu-baseline --no-other-kind 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; done
python bench.py --kind clustered --no-extras --no-cpu-baseline --no-other-kind 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1
timeout 300 tools/prof_cmd.sh r06q_clustered python $R/bench.py --kind clustered --no-extras --no-cpu-baseline --no-other-kind > gpurun_out/r06q_bench_clustered_stats.txt 2>&1
head -7 gpurun_out/r06q_bench_clustered_stats.txt
rm -f gpurun_out/prof_r06q_*/run_kernel_trace.csv
