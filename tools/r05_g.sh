#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r05i
timeout 900 python -m pytest tests -m gpu -x -q -k "aploss or training" 2>&1 | tail -2
python tools/aploss_time.py 2>/dev/null | tee gpurun_out/r05i/aploss_times.jsonl
PYTHONPATH=$PWD bash tools/prof_cmd.sh r05i_ap python $PWD/tools/aploss_time.py 2>&1 | head -8
