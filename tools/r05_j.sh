#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "ungrouped or random_vs_oracle or golden_small or golden_box or parameter_extremes or non_default or batched_ragged" 2>&1 | tail -5
PYTHONPATH=$PWD timeout 300 bash tools/prof_cmd.sh r05j_un4096 python $PWD/tools/mode_prof.py ungrouped --boxes 4096 2>&1 | grep ungrouped
timeout 200 python tools/mode_times.py --only ungrouped --kind uniform 2>&1 | grep "^{"
