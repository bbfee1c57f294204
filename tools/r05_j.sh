#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "ungrouped or random_vs_oracle or golden_small or golden_box or parameter_extremes or non_default or batched_ragged or tight_workspace" 2>&1 | tail -5
PYTHONPATH=$PWD timeout 300 bash tools/prof_cmd.sh r05j_unm python $PWD/tools/mode_prof.py ungrouped --boxes 4096 --matrix-in 2>&1 | grep -i "ungrouped\|iou2d"
timeout 200 python tools/mode_times.py --only ungrouped --kind uniform 2>&1 | grep "^{"
timeout 300 python tools/deep_fuzz.py 201 150 2>&1 | tail -3
