#!/bin/bash
# usage: tools/profile_round.sh <round tag, e.g. r02>   -- the evidence set kept under profiles/ (run on the GPU box, then copy):
#   <tag>_bench.json + <tag>_bench_kernel_stats.csv      the default bench line and the rocprofv3 --kernel-trace --stats of the same command
#   <tag>_pmc_summary.json                               HBM traffic per launch (tools/pmc.sh), fed back into the bench line's roofline.traffic
#   <tag>_grid.jsonl                                     N in {256,1024,4096,16384} x {clustered,uniform} (+ 3D at 4096/16384)
#   <tag>_two_calls_{uniform,clustered}_kernel_stats.csv, _b1_n4096_, _n512_, _n1024_   kernel stats of the reference's call sequence, of one image, of the small-N regime
#   <tag>_clustered4096_kernel_stats.csv                 kernel stats of the clustered N=4096 run (the headline runs on uniform boxes since round 3)
#   <tag>_dim3_n16384_kernel_stats.csv                   kernel stats of the 3D N=16384 run
#   <tag>_store_geometry.jsonl, _small_n.jsonl, _sgemm_mfma.txt   store-pattern ceilings, host cost of the small-N regime, the fp32 MFMA GEMM
#   <tag>_sgemm_kernel_stats.csv, _mode_times.jsonl, _host_overhead_n512.txt   GEMM kernel times at 1024^3 / 2048^3, every mode of the layer, cProfile of the eager step
#   <tag>_{ungrouped,unmasked}_uniform_kernel_stats.csv   kernel stats of the two non-default modes (tools/mode_prof.py)
export TMPDIR=/tmp
T=$1
O=gpurun_out/profiles_$T
mkdir -p $O
bash tools/pmc.sh $T > $O/pmc_stdout.txt 2>&1
cp gpurun_out/pmc_${T}_summary.json $O/${T}_pmc_summary.json
python bench.py --steps 200 --warmup 20 --pmc-summary $O/${T}_pmc_summary.json > $O/${T}_bench.json 2> $O/bench.err
bash tools/prof.sh ${T}_main --steps 200 --warmup 20 --no-cpu-baseline --no-other-kind > $O/prof_main.txt 2>&1
cp gpurun_out/prof_${T}_main/bench_kernel_stats.csv $O/${T}_bench_kernel_stats.csv
tail -1 gpurun_out/prof_${T}_main/bench_stdout.txt > $O/${T}_bench_under_rocprof.json
bash tools/prof.sh ${T}_d3 --steps 30 --warmup 5 --no-cpu-baseline --no-other-kind --kind clustered --dim 3 --boxes 16384 > $O/prof_d3.txt 2>&1
cp gpurun_out/prof_${T}_d3/bench_kernel_stats.csv $O/${T}_dim3_n16384_kernel_stats.csv
bash tools/prof.sh ${T}_d34k --steps 100 --warmup 5 --no-cpu-baseline --no-other-kind --kind clustered --dim 3 > $O/prof_d34k.txt 2>&1
cp gpurun_out/prof_${T}_d34k/bench_kernel_stats.csv $O/${T}_dim3_n4096_kernel_stats.csv
for k in uniform clustered; do
  bash tools/prof.sh ${T}_tc_$k --two-calls --kind $k --steps 100 --warmup 10 --no-cpu-baseline --no-other-kind > $O/prof_tc_$k.txt 2>&1
  cp gpurun_out/prof_${T}_tc_$k/bench_kernel_stats.csv $O/${T}_two_calls_${k}_kernel_stats.csv
done
bash tools/prof.sh ${T}_b1 --batch 1 --steps 200 --warmup 20 --no-cpu-baseline --no-other-kind > $O/prof_b1.txt 2>&1
cp gpurun_out/prof_${T}_b1/bench_kernel_stats.csv $O/${T}_b1_n4096_kernel_stats.csv
for n in 512 1024; do
  bash tools/prof.sh ${T}_n$n --boxes $n --steps 200 --warmup 20 --no-cpu-baseline --no-other-kind > $O/prof_n$n.txt 2>&1
  cp gpurun_out/prof_${T}_n$n/bench_kernel_stats.csv $O/${T}_n${n}_kernel_stats.csv
done
bash tools/prof.sh ${T}_clu --steps 200 --warmup 20 --no-cpu-baseline --no-other-kind --kind clustered > $O/prof_clu.txt 2>&1
cp gpurun_out/prof_${T}_clu/bench_kernel_stats.csv $O/${T}_clustered4096_kernel_stats.csv
tail -1 gpurun_out/prof_${T}_clu/bench_stdout.txt > $O/${T}_clustered4096_bench_under_rocprof.json
: > $O/${T}_grid.jsonl
for n in 256 1024 4096 16384; do for k in clustered uniform; do
  st=100; [ $n -ge 16384 ] && st=30
  python bench.py --boxes $n --kind $k --steps $st --warmup 10 --no-cpu-baseline --no-other-kind 2>/dev/null | tail -1 >> $O/${T}_grid.jsonl
done; done
for n in 4096 16384; do for k in clustered uniform; do
  st=100; [ $n -ge 16384 ] && st=30
  python bench.py --dim 3 --boxes $n --kind $k --steps $st --warmup 10 --no-cpu-baseline --no-other-kind 2>/dev/null | tail -1 >> $O/${T}_grid.jsonl
done; done
python bench.py --two-calls --kind uniform --steps 100 --warmup 10 --no-other-kind --cpu-seconds 3 2>/dev/null | tail -1 > $O/${T}_two_calls_bench.json
for n in 512 1024 2048; do python bench.py --graph --kind clustered --boxes $n --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 >> $O/${T}_graph.jsonl; done
python tools/kernel_times.py > $O/${T}_kernel_times.txt 2>/dev/null
python tools/kernel_times.py --boxes 16384 --reps 10 >> $O/${T}_kernel_times.txt 2>/dev/null
python tools/small_n.py 2>/dev/null | grep "^{" > $O/${T}_small_n.jsonl
GNMS_BINDING=ctypes python tools/small_n.py 2>/dev/null | grep "^{" >> $O/${T}_small_n.jsonl
python tools/sgemm_time.py 1024 2048 4096 8192 2>/dev/null | grep "sgemm\|soft_sort" > $O/${T}_sgemm_mfma.txt
bash tools/sgemm_pmc.sh 4096 >> $O/${T}_sgemm_mfma.txt 2>/dev/null
# kernel times of the GEMM at the sizes whose ten back-to-back calls are host-bound in sgemm_time.py (the panel scratch's hipMallocAsync / hipFreeAsync)
PYTHONPATH=$PWD bash tools/prof_cmd.sh ${T}_sgemm python $PWD/tools/sgemm_time.py 1024 2048 > $O/prof_sgemm.txt 2>&1
cp gpurun_out/prof_${T}_sgemm/run_kernel_stats.csv $O/${T}_sgemm_kernel_stats.csv
python tools/mode_times.py > $O/${T}_mode_times.jsonl 2>/dev/null
for m in ungrouped unmasked; do
  PYTHONPATH=$PWD bash tools/prof_cmd.sh ${T}_$m python $PWD/tools/mode_prof.py $m > $O/prof_$m.txt 2>&1
  cp gpurun_out/prof_${T}_$m/run_kernel_stats.csv $O/${T}_${m}_uniform_kernel_stats.csv
done
python tools/host_overhead.py --boxes 512 --steps 1500 2>/dev/null | head -12 > $O/${T}_host_overhead_n512.txt
python tools/aploss_time.py > $O/${T}_aploss_times.jsonl 2>/dev/null
python tools/e2e_bench.py --mode infer --steps 10 2>/dev/null | tail -1 > $O/${T}_e2e.jsonl
python tools/e2e_bench.py --mode train --steps 10 2>/dev/null | tail -1 >> $O/${T}_e2e.jsonl
python tools/proposals_time.py 2>/dev/null | grep "^{" > $O/${T}_proposals_times.jsonl
PYTHONPATH=$PWD bash tools/prof_cmd.sh ${T}_prop python $PWD/tools/proposals_time.py > $O/prof_prop.txt 2>&1
cp gpurun_out/prof_${T}_prop/run_kernel_stats.csv $O/${T}_proposals_kernel_stats.csv
PYTHONPATH=$PWD bash tools/prof_cmd.sh ${T}_ap python $PWD/tools/aploss_time.py > $O/prof_ap.txt 2>&1
cp gpurun_out/prof_${T}_ap/run_kernel_stats.csv $O/${T}_aploss_kernel_stats.csv
python tools/tail_time.py 2>/dev/null | grep "^{" > $O/${T}_training_tail.jsonl
for cfg in "1 4096 uniform" "8 4096 uniform" "8 1024 uniform" "8 512 uniform"; do set -- $cfg; echo "== B=$1 N=$2 $3 (lists)" >> $O/${T}_phase_ticks.txt; GNMS_LIB_PATH=build/timing/libgroomed_nms_hip.so GNMS_BINDING=ctypes timeout 300 python tools/phase_ticks.py --batch $1 --boxes $2 --kind $3 --lists 2>&1 | grep -v amdgpu.ids >> $O/${T}_phase_ticks.txt; done
cat $O/${T}_bench.json; echo; cat $O/${T}_grid.jsonl | python -c "
import json,sys
for l in sys.stdin:
    if l.strip():
        d=json.loads(l); r=d['roofline'] or {}
        print(d['config']['workload'][:46], d['ms_per_step'], '%.3e'%d['value'], r.get('kernel'), r.get('kernel_ms'), r.get('frac'))"
