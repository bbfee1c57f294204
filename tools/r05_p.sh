#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "topk or proposals or two_host_threads or tail or e2e or launcher" 2>&1 | tail -3
for s in 4 5; do timeout 280 python tools/topk_stress.py $s 150 2>&1 | tail -2; done
O=gpurun_out/profiles_r05g; mkdir -p $O
python tools/proposals_time.py 2>/dev/null | grep "^{" > $O/r05g_proposals_times.jsonl; grep topk $O/r05g_proposals_times.jsonl
PYTHONPATH=$PWD bash tools/prof_cmd.sh r05g_prop python $PWD/tools/proposals_time.py 2>&1 | grep -i "topk\|fill"
cp gpurun_out/prof_r05g_prop/run_kernel_stats.csv $O/r05g_proposals_kernel_stats.csv
python tools/e2e_bench.py --mode infer --steps 10 2>/dev/null | tail -1 > $O/r05g_e2e.jsonl
python tools/e2e_bench.py --mode train --steps 10 2>/dev/null | tail -1 >> $O/r05g_e2e.jsonl
cut -c1-60,380-480 $O/r05g_e2e.jsonl
