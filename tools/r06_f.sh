#!/bin/bash
# round 6, session f: bitmask_boxes_kernel with rankof[] in LDS; eager vs graph replay gaps; the sanitizer runs
R=$PWD
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r06i_tests.txt; cat gpurun_out/r06i_tests.txt
timeout 300 tools/prof_cmd.sh r06i_bench python $R/bench.py --no-extras --no-cpu-baseline --no-other-kind > gpurun_out/r06i_bench_stats.txt 2>&1
head -8 gpurun_out/r06i_bench_stats.txt; grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_r06i_bench/stdout.txt | head -2
timeout 300 tools/prof_cmd.sh r06i_graph python $R/bench.py --graph --no-extras --no-cpu-baseline --no-other-kind > gpurun_out/r06i_graph_stats.txt 2>&1
grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_r06i_graph/stdout.txt | head -2
python tools/graph_gaps.py gpurun_out/prof_r06i_bench/run_kernel_trace.csv > gpurun_out/r06i_gaps_eager.txt 2>&1
python tools/graph_gaps.py gpurun_out/prof_r06i_graph/run_kernel_trace.csv > gpurun_out/r06i_gaps_graph.txt 2>&1
cat gpurun_out/r06i_gaps_eager.txt gpurun_out/r06i_gaps_graph.txt
rm -f gpurun_out/prof_r06i_*/run_kernel_trace.csv
if [ -f build/san_address/libgroomed_nms_hip.so ]; then bash tools/asan.sh run address 2>&1 | tail -40; fi
