#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "topk or proposals or two_host_threads or tail or e2e or launcher" 2>&1 | tail -4
python tools/proposals_time.py 2>/dev/null | grep "^{" | grep topk
PYTHONPATH=$PWD bash tools/prof_cmd.sh r05p_prop python $PWD/tools/proposals_time.py 2>&1 | grep -i "topk\|fill"
python tools/e2e_bench.py --mode infer --steps 10 2>/dev/null | tail -1 | cut -c1-60,400-470
python tools/e2e_bench.py --mode train --steps 10 2>/dev/null | tail -1 | cut -c1-60,400-480
