"""The reference call at its own size for the tracer: 1000 x differentiable_nms(scores, iou) on GPU tensors, N = 500 (rocprofv3 --kernel-trace --stats -- python tools/single_n500.py)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from groomed_nms_amd import groomed_nms as GN, overlaps, synthetic
b, s = synthetic.batch_2d(1, 1, 500, "clustered", per=25)
boxes = torch.from_numpy(b[0]).cuda(); scores = torch.from_numpy(s[0]).cuda().requires_grad_(True)
iou = overlaps.iou(boxes, boxes)
for _ in range(1000):
    GN.differentiable_nms(scores, iou)
torch.cuda.synchronize()
