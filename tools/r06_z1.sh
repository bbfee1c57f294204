#!/bin/bash
# round 6, final evidence, part 1: both test configurations, the plain bench line, the small tools, the sanitizers (last: they swap the libraries of this scratch copy)
R=$PWD
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -4 > gpurun_out/r06z_gputest.txt; cat gpurun_out/r06z_gputest.txt
GNMS_ONE_LAUNCH=0 timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -4 > gpurun_out/r06z_gputest_three_launches.txt; cat gpurun_out/r06z_gputest_three_launches.txt
( time python bench.py > gpurun_out/r06z_bench_default_run.json 2> gpurun_out/r06z_bench_default_run.err ) 2> gpurun_out/r06z_bench_default_run_time.txt; cat gpurun_out/r06z_bench_default_run_time.txt
timeout 300 tools/prof_cmd.sh r06z_single python $R/tools/single_n500.py > gpurun_out/r06z_single_n500_stats.txt 2>&1; cp gpurun_out/prof_r06z_single/run_kernel_stats.csv gpurun_out/r06z_single_n500_kernel_stats.csv; head -3 gpurun_out/r06z_single_n500_stats.txt
rm -f gpurun_out/prof_r06z_single/run_kernel_trace.csv
timeout 300 python tools/small_n.py 2>/dev/null | grep "^{" > gpurun_out/r06z_small_n.jsonl; tail -3 gpurun_out/r06z_small_n.jsonl
GNMS_ONE_LAUNCH=0 timeout 300 python tools/small_n.py 2>/dev/null | grep "^{" > gpurun_out/r06z_small_n_three_launches.jsonl; tail -3 gpurun_out/r06z_small_n_three_launches.jsonl
timeout 600 python tools/nms_host.py 2>/dev/null | grep "^{" > gpurun_out/r06z_nms_host.jsonl; cat gpurun_out/r06z_nms_host.jsonl
timeout 600 python tools/sgemm_variants.py 2>/dev/null | grep "^n=" > gpurun_out/r06z_sgemm_variants.txt; cat gpurun_out/r06z_sgemm_variants.txt
for seed in 611 612 613; do timeout 900 python tools/deep_fuzz.py $seed 300 2>&1 | tail -1 | sed "s/^/seed $seed: /"; done > gpurun_out/r06z_deep_fuzz.txt 2>&1; cat gpurun_out/r06z_deep_fuzz.txt
timeout 600 python tools/topk_stress.py 2>&1 | tail -1 > gpurun_out/r06z_topk_stress.txt; cat gpurun_out/r06z_topk_stress.txt
bash tools/asan.sh run binding 2>&1 | tail -4
cp gpurun_out/r06_asan_binding.txt gpurun_out/r06z_asan_binding.txt; cp gpurun_out/r06_asan_binding_pytest.txt gpurun_out/r06z_asan_binding_pytest.txt
bash tools/asan.sh run thread 2>&1 | tail -6
cp gpurun_out/r06_tsan.txt gpurun_out/r06z_tsan.txt; cp gpurun_out/r06_thread_pytest.txt gpurun_out/r06z_tsan_pytest.txt
