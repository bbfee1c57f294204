"""Developer tool: N steps of one mode of the layer (one-call 2D entry + backward) for a kernel trace.
python tools/mode_prof.py unmasked|ungrouped|default [--boxes 4096] [--batch 8] [--kind uniform] [--matrix-in]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import groomed_nms_amd as G  # noqa: E402
from groomed_nms_amd import overlaps, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("mode")
ap.add_argument("--boxes", type=int, default=4096)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--kind", default="uniform")
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--matrix-in", action="store_true")
a = ap.parse_args()
kw = {"default": {}, "unmasked": dict(mask_group_boxes=False), "ungrouped": dict(group_boxes=False)}[a.mode]
b, s = synthetic.batch_2d(1000, a.batch, a.boxes, a.kind)
boxes = torch.from_numpy(b).cuda()
scores = torch.from_numpy(s).cuda().requires_grad_(True)
w = torch.linspace(-1, 2, a.boxes, device="cuda").repeat(a.batch, 1).contiguous()
bufs = [torch.empty((a.batch, a.boxes, a.boxes), device="cuda") for _ in range(3)]
for i in range(a.steps + 5):
    buf = bufs[i % 3]
    if a.matrix_in:
        prob = G.differentiable_nms_batched(scores, overlaps.iou_batched(boxes, out=buf), **kw)[0]
    else:
        prob = G.differentiable_nms_with_iou2d_batched(scores, boxes, iou_out=buf, **kw)[0]
    scores.grad = None
    torch.autograd.backward(prob, w)
torch.cuda.synchronize()
