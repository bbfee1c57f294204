#!/bin/bash
R=$PWD
timeout 600 python tools/sgemm_variants.py > gpurun_out/r06m_sgemm_variants.txt 2>&1; cat gpurun_out/r06m_sgemm_variants.txt
timeout 300 python tools/sgemm_time.py 4096 2>&1 | tail -2
timeout 600 python -m pytest tests -x -q -m gpu -k "soft_sort or sgemm" 2>&1 | tail -3
GNMS_LIB_PATH=build/timing/libgroomed_nms_hip.so GNMS_BINDING=ctypes timeout 300 python tools/bits_ticks.py > gpurun_out/r06m_bits_timeline.txt 2>&1; cat gpurun_out/r06m_bits_timeline.txt
for n in 4096 16384; do
timeout 300 tools/prof_cmd.sh r06m_nms_n$n python $R/tools/nms_host_prof.py $n > gpurun_out/r06m_nms_host_n${n}_stats.txt 2>&1; head -6 gpurun_out/r06m_nms_host_n${n}_stats.txt
rm -f gpurun_out/prof_r06m_nms_n$n/run_kernel_trace.csv
done
