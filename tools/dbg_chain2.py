import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import groomed_nms_amd as G
from groomed_nms_amd import synthetic, overlaps
from oracle import oracle as O
for (B, N, kind) in ((1, 5120, "uniform"), (1, 6144, "uniform"), (1, 7168, "uniform"), (1, 7169, "uniform"), (1, 8192, "uniform"), (1, 8192, "clustered"), (1, 12288, "uniform")):
    boxes, scores = synthetic.batch_2d(5, B, N, kind)
    bt = torch.from_numpy(boxes).cuda(); st = torch.from_numpy(scores).cuda()
    out2 = G.differentiable_nms_from_boxes_batched(st, bt) if hasattr(G, "differentiable_nms_from_boxes_batched") else None
    iou = overlaps.iou_batched(bt)
    out = G.differentiable_nms_batched(st, iou)
    torch.cuda.synchronize()
    m = O.iou2d(boxes[0], boxes[0]); ref = O.differentiable_nms(scores[0], m)
    p = out[0][0].cpu().numpy()
    bad = np.nonzero(p != ref["prob"])[0]
    print(B, N, kind, "matrix-in == oracle:", bad.size == 0, "first bad ranks", bad[:6].tolist(), "nvalid", int(out[4][0]), len(ref["valid"]), flush=True)
