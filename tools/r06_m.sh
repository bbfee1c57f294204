#!/bin/bash
R=$PWD
timeout 600 python -m pytest tests -x -q -m gpu -k "classic or nms_others or kitti or n_rank or smoke" 2>&1 | tail -5 > gpurun_out/r06s_tests.txt; cat gpurun_out/r06s_tests.txt
timeout 600 python tools/nms_host.py > gpurun_out/r06s_nms_host.jsonl 2>&1; tail -5 gpurun_out/r06s_nms_host.jsonl
timeout 300 tools/prof_cmd.sh r06s_nms_n126720 python $R/tools/nms_host_prof.py 126720 > gpurun_out/r06s_nms_host_n126720_stats.txt 2>&1; head -8 gpurun_out/r06s_nms_host_n126720_stats.txt
rm -f gpurun_out/prof_r06s_nms_n126720/run_kernel_trace.csv
