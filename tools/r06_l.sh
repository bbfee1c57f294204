#!/bin/bash
R=$PWD
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r06p_tests.txt; cat gpurun_out/r06p_tests.txt
timeout 600 python tools/nms_host.py > gpurun_out/r06p_nms_host.jsonl 2>&1; tail -5 gpurun_out/r06p_nms_host.jsonl
for n in 16384 126720; do
timeout 300 tools/prof_cmd.sh r06p_nms_n$n python $R/tools/nms_host_prof.py $n > gpurun_out/r06p_nms_host_n${n}_stats.txt 2>&1; head -5 gpurun_out/r06p_nms_host_n${n}_stats.txt
rm -f gpurun_out/prof_r06p_nms_n$n/run_kernel_trace.csv
done
