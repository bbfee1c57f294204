import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import groomed_nms_amd as G
from groomed_nms_amd import synthetic, overlaps
for N in (500, 2000):
    boxes_np, scores_np = synthetic.batch_2d(1, 1, N, "clustered", per=25)
    boxes = torch.from_numpy(boxes_np[0]).cuda()
    scores = torch.from_numpy(scores_np[0]).cuda().requires_grad_(True)
    iou = overlaps.iou(boxes, boxes)
    def fwd():
        return G.differentiable_nms(scores, iou)
    def fwdbwd():
        out = G.differentiable_nms(scores, iou)
        scores.grad = None
        out[2].sum().backward()
    iou_np = iou.cpu().numpy(); s_np = scores_np[0].astype(np.float64)
    def np_path():
        return G.differentiable_nms(s_np, iou_np)
    def iou_only():
        return overlaps.iou(boxes, boxes)
    for name, fn in (("iou (combinations)", iou_only), ("differentiable_nms fwd (GPU tensors, incl. its host sync)", fwd), ("fwd+bwd", fwdbwd), ("NumPy in (PCIe)", np_path)):
        for _ in range(10): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): fn()
        torch.cuda.synchronize()
        print("N=%d %-60s %.1f us/call" % (N, name, (time.perf_counter() - t0) / 200 * 1e6))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(300): fwd()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
