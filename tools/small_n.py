#!/usr/bin/env python3
"""The reference's own regime (N <= 500 boxes per image, lib/rpn_util.py:1293, lib/loss/rpn_3d.py:732): what one eager step and one
single call cost, through the C++ binding (default) and through ctypes (GNMS_BINDING=ctypes), JSON lines.
    eager step  = differentiable_nms_with_iou2d_batched + backward, B = 8, N in {256, 512, 1024}
    single call = lib.core.iou(boxes, boxes) then differentiable_nms(scores, iou) on GPU tensors, N = 500 (plain tensors: one host
                  sync per call; LAZY_INDEX_LISTS = True: none)
python tools/small_n.py"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from groomed_nms_amd import groomed_nms as GN, overlaps, synthetic   # noqa: E402

binding = "ctypes" if os.environ.get("GNMS_BINDING") == "ctypes" or not GN._binding() else "c++ (gnms_torch)"


def time_it(fn, n=2000, warm=100):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


# (an untimed pass first: whichever configuration ran first in the process read 15-20 us high -- allocator growth, first launches, the host's
# clocks -- which showed up as "N = 256 slower than N = 512"; tools/smalln_order.py: 37.9 / 45.7 us for N = 256 in two places of one run)
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

    us_step = time_it(step)
    if N >= 256:
        print(json.dumps({"what": "eager step, B=8, one-call entry + backward", "N": N, "binding": binding, "us_per_step": round(us_step, 1)}))

b, s = synthetic.batch_2d(1, 1, 500, "clustered", per=25)
boxes = torch.from_numpy(b[0]).cuda()
scores = torch.from_numpy(s[0]).cuda().requires_grad_(True)
iou = overlaps.iou(boxes, boxes)
for lazy in (False, True):
    GN.LAZY_INDEX_LISTS = lazy
    us = time_it(lambda: GN.differentiable_nms(scores, iou))

    def fwdbwd():
        out = GN.differentiable_nms(scores, iou)
        scores.grad = None
        out[2].sum().backward()

    print(json.dumps({"what": "single call differentiable_nms(scores, iou), GPU tensors, N=500", "binding": binding, "lazy_index_lists": lazy,
                      "counts_to_host": "torch copy (GNMS_COUNTS_MAILBOX=0)" if GN._PLAIN_COUNTS else "pinned mailbox (gnms_counts_to_host)",
                      "us_per_call": round(us, 1), "us_fwd_bwd": round(time_it(fwdbwd), 1)}))
GN.LAZY_INDEX_LISTS = False
print(json.dumps({"what": "iou(boxes, boxes) N=500", "binding": binding, "us_per_call": round(time_it(lambda: overlaps.iou(boxes, boxes)), 1)}))
