"""Developer: a random sequence of gnms_select_topk calls in one process (the cooperative kernel keeps ONE scratch per device between the calls and
leaves its header zeroed: any residue of a call would show in a later one) against the oracle's stable descending sort.
usage: python tools/topk_stress.py [SEED] [CALLS]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from groomed_nms_amd import proposals as PR  # noqa: E402
from oracle import proposals_oracle as PO  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(seed)
fails = 0
for it in range(calls):
    B = int(rng.integers(1, 9))
    A = int(rng.choice([4097, 5000, 8192, 8193, 20000, 40000, 126720]))
    K = int(rng.choice([1, 7, 500, 1024, 1025, 3000, 4096, 6000]))
    sc = rng.random((B, A), dtype=np.float32)
    mode = int(rng.integers(0, 4))
    if mode == 1:
        sc = np.round(sc * 20) / 20
    elif mode == 2:
        sc[0] = 0.5
    elif mode == 3:
        sc = (0.76 + 0.003 * sc).astype(np.float32)
    counts = rng.integers(1, A + 1, size=B).astype(np.int32) if rng.uniform() < 0.5 else None
    st = torch.from_numpy(sc).cuda()
    ct = torch.from_numpy(counts).cuda() if counts is not None else None
    cand = None
    if counts is not None:
        cand_np = np.stack([rng.permutation(A) for _ in range(B)]).astype(np.int32)
        cand = torch.from_numpy(cand_np).cuda()
    idx, num, ssel, _ = PR.select_topk(st, K, cand, ct)
    for b in range(B):
        want = PO.select_topk(sc[b], None if counts is None else cand_np[b, :counts[b]], K)
        ok = int(num[b]) == len(want) and idx[b, :len(want)].cpu().tolist() == want.tolist() and bool((idx[b, len(want):] == -1).all())
        if not ok:
            fails += 1
            print("FAIL", it, (B, A, K, mode, None if counts is None else counts.tolist()), "image", b, flush=True)
            break
print("seed %d: %d calls, %d failures" % (seed, calls, fails))
sys.exit(1 if fails else 0)
