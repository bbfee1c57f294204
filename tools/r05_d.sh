#!/bin/bash
export TMPDIR=/tmp
T=${1:-r05f}
O=gpurun_out/$T
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "topk or proposals or training" > $O/pytest_sel.txt 2>&1; echo "pytest rc $?" >> $O/pytest_sel.txt
tail -5 $O/pytest_sel.txt
timeout 300 python tools/proposals_time.py 2>&1 | grep "^{" | grep anchors
