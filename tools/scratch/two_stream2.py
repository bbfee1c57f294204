"""experiment: persistent IoU writer (one 16-wave workgroup per CU) on a side stream while the from-boxes chain runs on the main stream"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import groomed_nms_amd as G
from groomed_nms_amd import overlaps, synthetic
B, N = 8, 4096
boxes_np, scores_np = synthetic.batch_2d(1000, B, N, "clustered")
dev = torch.device("cuda", 0)
boxes = torch.from_numpy(boxes_np).to(dev)
scores = torch.from_numpy(scores_np).to(dev).requires_grad_(True)
w = torch.linspace(-1.0, 2.0, N, device=dev).repeat(B, 1).contiguous()
iou_buf = torch.empty((B, N, N), dtype=torch.float32, device=dev)
side = torch.cuda.Stream()
def iou_only():
    overlaps.iou_batched(boxes, out=iou_buf)
def chain_only():
    prob = G.differentiable_nms_from_boxes_batched(scores, boxes)[0]
    scores.grad = None
    torch.autograd.backward(prob, w)
def seq():
    iou_only(); chain_only()
def two():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        iou_only()
    chain_only()
    main.wait_stream(side)
for name, fn in (("iou only", iou_only), ("chain only", chain_only), ("sequential", seq), ("two streams", two), ("sequential", seq), ("two streams", two)):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): fn()
    torch.cuda.synchronize()
    print("%-14s %.4f ms" % (name, (time.perf_counter() - t0) / 200 * 1e3))
