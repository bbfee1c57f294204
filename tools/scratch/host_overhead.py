import sys, os, time, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import groomed_nms_amd as G
from groomed_nms_amd import synthetic
B, N = 8, 512
boxes_np, scores_np = synthetic.batch_2d(1000, B, N, "clustered")
dev = torch.device("cuda", 0)
boxes = torch.from_numpy(boxes_np).to(dev)
scores = torch.from_numpy(scores_np).to(dev).requires_grad_(True)
w = torch.linspace(-1.0, 2.0, N, device=dev).repeat(B, 1).contiguous()
iou_buf = torch.empty((B, N, N), dtype=torch.float32, device=dev)
def one():
    prob = G.differentiable_nms_with_iou2d_batched(scores, boxes, iou_out=iou_buf)[0]
    scores.grad = None
    torch.autograd.backward(prob, w)
def fb():
    prob = G.differentiable_nms_from_boxes_batched(scores, boxes)[0]
    scores.grad = None
    torch.autograd.backward(prob, w)
def fwd_only():
    with torch.no_grad():
        G.differentiable_nms_with_iou2d_batched(scores, boxes, iou_out=iou_buf)
for name, fn in (("with_iou2d fwd+bwd", one), ("from_boxes fwd+bwd", fb), ("with_iou2d fwd only (no_grad)", fwd_only)):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-32s host %.1f us/step, incl. drain %.1f us/step" % (name, (t1 - t0) / 500 * 1e6, (t2 - t0) / 500 * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(300): one()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
