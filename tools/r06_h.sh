#!/bin/bash
R=$PWD
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r06l_tests.txt; cat gpurun_out/r06l_tests.txt
GNMS_LIB_PATH=build/timing/libgroomed_nms_hip.so GNMS_BINDING=ctypes timeout 300 python tools/bits_ticks.py > gpurun_out/r06l_bits_timeline.txt 2>&1; cat gpurun_out/r06l_bits_timeline.txt
timeout 300 tools/prof_cmd.sh r06l_bench python $R/bench.py --no-extras --no-cpu-baseline --no-other-kind > gpurun_out/r06l_bench_stats.txt 2>&1
head -6 gpurun_out/r06l_bench_stats.txt
for i in 1 2 3; do python bench.py --no-extras --no-cpu-baseline --no-other-kind 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; done
python bench.py --kind clustered --no-extras --no-cpu-baseline --no-other-kind 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1
rm -f gpurun_out/prof_r06l_*/run_kernel_trace.csv
timeout 600 python tools/sgemm_variants.py > gpurun_out/r06l_sgemm_variants.txt 2>&1; cat gpurun_out/r06l_sgemm_variants.txt
timeout 300 python tools/sgemm_time.py 4096 2>&1 | tail -2
timeout 600 python tools/nms_host.py > gpurun_out/r06l_nms_host.jsonl 2>&1; tail -12 gpurun_out/r06l_nms_host.jsonl
