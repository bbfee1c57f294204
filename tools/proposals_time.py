"""Times the front-end kernels of SURVEY 8-f2 (decode, projected boxes, score top-K) with HIP events.
usage: python tools/proposals_time.py  -> one JSON line per kernel."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from groomed_nms_amd import proposals as PR  # noqa: E402


def timed(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3          # us


def main():
    rng = np.random.default_rng(0)
    B, A = 8, 32 * 110 * 36                            # anchors of a 512 x 1760 image at stride 16, 36 anchors per cell
    anchors = torch.from_numpy(rng.uniform(0, 1000, (A, 4)).astype(np.float32)).cuda()
    deltas = torch.from_numpy((rng.standard_normal((B, A, 4)) * 0.3).astype(np.float32)).cuda()
    t = timed(lambda: PR.bbox_transform_inv(anchors, deltas, means=[0, 0, 0, 0], stds=[0.1, 0.1, 0.2, 0.2]))
    nbytes = B * A * 32 + A * 16
    print(json.dumps({"kernel": "bbox_transform_inv", "B": B, "A": A, "us": round(t, 1), "GB_per_s": round(nbytes / t / 1e3, 1)}))
    scores = torch.from_numpy(rng.uniform(0, 1, (B, A)).astype(np.float32)).cuda()
    boxes = torch.from_numpy(rng.uniform(0, 1000, (B, A, 4)).astype(np.float32)).cuda()
    for F, K in ((4096, 500), (1024, 500), (16384, 4096)):
        cand = torch.from_numpy(np.stack([rng.choice(A, F, replace=False) for _ in range(B)]).astype(np.int32)).cuda()
        t = timed(lambda: PR.select_topk(scores, K, cand, None, boxes))
        print(json.dumps({"kernel": "select_topk (+ gather of scores and boxes)", "B": B, "candidates": F, "K": K, "us": round(t, 1)}))
    for K in (500, 4096):                               # among ALL anchors: radix pre-selection + the sort of the K survivors
        t = timed(lambda: PR.select_topk(scores, K, None, None, boxes))
        print(json.dumps({"kernel": "select_topk among all anchors (+ gathers)", "B": B, "candidates": A, "K": K, "us": round(t, 1)}))
    for Bq, K in ((1, 4096), (1, 500), (4, 4096)):   # C3 / C4's shapes: one (four) image(s), all anchors
        sq, bq = scores[:Bq].contiguous(), boxes[:Bq].contiguous()
        t = timed(lambda: PR.select_topk(sq, K, None, None, bq))
        print(json.dumps({"kernel": "select_topk among all anchors (+ gathers)", "B": Bq, "candidates": A, "K": K, "us": round(t, 1)}))
    N = 4096
    par = torch.from_numpy(np.stack([rng.uniform(-20, 20, (B, N)), rng.uniform(0.5, 2.5, (B, N)), rng.uniform(4, 60, (B, N)),
                                     rng.uniform(1.4, 2, (B, N)), rng.uniform(1.3, 2, (B, N)), rng.uniform(3, 5, (B, N)),
                                     rng.uniform(-3.1, 3.1, (B, N))], 2).astype(np.float32)).cuda()
    p2 = np.array([[721.5, 0, 609.6, 44.9], [0, 721.5, 172.9, 0.22], [0, 0, 1, 0.0027], [0, 0, 0, 1]], np.float32)
    p2 = torch.from_numpy(p2).cuda().unsqueeze(0).repeat(B, 1, 1)
    sc = torch.full((B,), 0.7, device="cuda")
    t = timed(lambda: PR.projected_boxes_2d(par, p2, sc))
    print(json.dumps({"kernel": "projected_boxes_2d", "B": B, "N": N, "us": round(t, 1)}))


if __name__ == "__main__":
    main()
