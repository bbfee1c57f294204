#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r05g
timeout 900 python -m pytest tests -m gpu -x -q -k "fast_tail or fuzz_layer or switches or ties or empty or random_vs_oracle" 2>&1 | tail -3
GNMS_E2E_DUMP=gpurun_out/r05g/e2e_proposals.npz python tools/e2e_bench.py --mode infer --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['split_ms'])"
for ft in 1 0; do GNMS_FAST_TAIL=$ft python - <<'PY'
import time, torch, numpy as np, sys, os
sys.path.insert(0,'.')
import groomed_nms_amd as G
d=np.load('gpurun_out/r05g/e2e_proposals.npz')
s=torch.from_numpy(d['scores']).cuda(); b=torch.from_numpy(d['boxes']).cuda(); num=torch.from_numpy(d['num']).cuda()
def t(fn,k=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/k*1e6
print("FAST_TAIL", os.environ.get("GNMS_FAST_TAIL"), "nms on the harness's proposals, wall us:", t(lambda: G.differentiable_nms_with_iou2d_batched(s,b,counts=num,index_lists=True)))
PY
done
python tools/e2e_bench.py --mode train --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('train', d['ms_per_step'], d['split_ms'])"
