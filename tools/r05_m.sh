#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "golden or random_vs_oracle or ragged or n4096 or ties or from_boxes or fuzz_layer or empty or one_call or scale" 2>&1 | tail -3
bash tools/prof.sh r05m_b1 --batch 1 --steps 200 --warmup 20 --no-cpu-baseline --no-other-kind 2>&1 | head -7
bash tools/prof.sh r05m_n500 --batch 2 --boxes 500 --steps 200 --warmup 20 --no-cpu-baseline --no-other-kind 2>&1 | head -6
