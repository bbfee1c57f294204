#!/usr/bin/env python3
"""Why is the eager B = 8 step SLOWER on the host at N = 256 than at N = 1024 (tools/small_n.py: 77 / 64 / 45 us)?  The host's enqueue time of the
same step when every launch meets an EMPTY queue (plain loop: the GPU finishes a small step before the host has enqueued the next) against
a queue that is kept busy (the steps enqueued behind a spin kernel), JSON lines.   python tools/idle_queue.py"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from groomed_nms_amd import groomed_nms as GN, synthetic   # noqa: E402

for N in (128, 256, 512, 1024):
    b, s = synthetic.batch_2d(1, 8, N, "clustered")
    boxes = torch.from_numpy(b).cuda()
    scores = torch.from_numpy(s).cuda().requires_grad_(True)
    w = torch.ones_like(scores)
    buf = torch.empty((8, N, N), device="cuda")

    def step():
        prob = GN.differentiable_nms_with_iou2d_batched(scores, boxes, iou_out=buf, index_lists=False)[0]
        scores.grad = None
        torch.autograd.backward(prob, w)

    def fwd_only():
        with torch.no_grad():
            GN.differentiable_nms_with_iou2d_batched(scores, boxes, iou_out=buf, index_lists=False)

    for name, fn in (("fwd + bwd", step), ("fwd only (no autograd)", fwd_only)):
        for _ in range(200):
            fn()
        torch.cuda.synchronize()
        res = {}
        for mode in ("empty queue", "busy queue"):
            best = 1e9
            for rep in range(5):
                torch.cuda.synchronize()
                k = 300
                if mode == "busy queue":
                    torch.cuda._sleep(int(2.4e9 * 0.04))          # ~40 ms of spinning in front: the host runs ahead of the GPU
                t0 = time.perf_counter()
                for _ in range(k):
                    fn()
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                best = min(best, (t1 - t0) / k * 1e6)
            res[mode] = round(best, 1)
        if N >= 256:
            print(json.dumps({"N": N, "B": 8, "what": name, "host_us_per_step": res}))
