"""Host cost of one bench step (Python API: autograd Function over ctypes): the step loop on a tiny problem, where the GPU is idle
most of the time, under cProfile.  python tools/host_overhead.py [--boxes 256]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from groomed_nms_amd import groomed_nms as G, synthetic   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--boxes", type=int, default=256)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--steps", type=int, default=2000)
a = ap.parse_args()
b, s = synthetic.batch_2d(1, a.batch, a.boxes, "clustered")
boxes = torch.from_numpy(b).cuda()
scores = torch.from_numpy(s).cuda().requires_grad_(True)
w = torch.ones_like(scores)
buf = torch.empty((a.batch, a.boxes, a.boxes), device="cuda")


def step():
    prob = G.differentiable_nms_with_iou2d_batched(scores, boxes, iou_out=buf)[0]
    scores.grad = None
    torch.autograd.backward(prob, w)


for _ in range(50):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host %.1f us/step (launch loop), %.1f us/step with the final sync" % ((t1 - t0) / a.steps * 1e6, (t2 - t0) / a.steps * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(a.steps):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
