#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/profiles_r05f; mkdir -p $O; T=r05f
for m in ungrouped unmasked; do
  PYTHONPATH=$PWD bash tools/prof_cmd.sh ${T}_$m python $PWD/tools/mode_prof.py $m 2>&1 | head -8
  cp gpurun_out/prof_${T}_$m/run_kernel_stats.csv $O/${T}_${m}_uniform_kernel_stats.csv
done
timeout 60 ./build/usolve_ticks > $O/${T}_usolve_ticks.txt 2>&1; tail -14 $O/${T}_usolve_ticks.txt
