#!/bin/bash
R=$PWD
timeout 600 python -m pytest tests -x -q -m gpu -k "from_boxes or n4096 or at_scale or one_call or full_size or fuzz_layer or adversarial or library_switches or recycled" 2>&1 | tail -5 > gpurun_out/r06k_tests.txt; cat gpurun_out/r06k_tests.txt
GNMS_LIB_PATH=build/timing/libgroomed_nms_hip.so GNMS_BINDING=ctypes timeout 300 python tools/bits_ticks.py > gpurun_out/r06k_bits_timeline.txt 2>&1; cat gpurun_out/r06k_bits_timeline.txt
timeout 300 tools/prof_cmd.sh r06k_bench python $R/bench.py --no-extras --no-cpu-baseline --no-other-kind > gpurun_out/r06k_bench_stats.txt 2>&1
head -6 gpurun_out/r06k_bench_stats.txt; grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_r06k_bench/stdout.txt | head -1
for i in 1 2 3; do python bench.py --no-extras --no-cpu-baseline --no-other-kind 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; done
python bench.py --kind clustered --no-extras --no-cpu-baseline --no-other-kind 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1
rm -f gpurun_out/prof_r06k_*/run_kernel_trace.csv
bash tools/asan.sh run binding 2>&1 | tail -12
bash tools/asan.sh run thread 2>&1 | tail -30
