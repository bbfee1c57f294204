#!/bin/bash
export TMPDIR=/tmp
T=${1:-r05f}
O=gpurun_out/$T
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q -k "topk or proposals or training or switches or matrix_in" > $O/pytest_sel.txt 2>&1; echo "pytest rc $?" >> $O/pytest_sel.txt
tail -5 $O/pytest_sel.txt
python tools/proposals_time.py 2>&1 | grep "^{" | tee $O/proposals_times.jsonl
python tools/e2e_bench.py --mode infer --steps 10 2>/dev/null | tail -1 | tee $O/e2e.jsonl
