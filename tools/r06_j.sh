#!/bin/bash
# the final build under the fuzz: other seeds, sizes on both sides of the one launch, both entries
for seed in 601 602 603 604 605 606; do
timeout 900 python tools/deep_fuzz.py $seed 300 2>&1 | tail -3 | sed "s/^/seed $seed: /"
done > gpurun_out/r06n_deep_fuzz.txt 2>&1
cat gpurun_out/r06n_deep_fuzz.txt
timeout 600 python tools/topk_stress.py > gpurun_out/r06n_topk_stress.txt 2>&1; tail -3 gpurun_out/r06n_topk_stress.txt
