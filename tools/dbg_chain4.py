import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import groomed_nms_amd as G
from groomed_nms_amd import synthetic, overlaps
from oracle import oracle as O
for N in [int(a) for a in sys.argv[1:]]:
    boxes, scores = synthetic.batch_2d(5, 1, N, "uniform")
    bt = torch.from_numpy(boxes).cuda(); st = torch.from_numpy(scores).cuda()
    print(N, "from boxes ...", flush=True)
    out = G.differentiable_nms_from_boxes_batched(st, bt); torch.cuda.synchronize()
    print(N, "iou ...", flush=True)
    iou = overlaps.iou_batched(bt); torch.cuda.synchronize()
    print(N, "matrix in ...", flush=True)
    out2 = G.differentiable_nms_batched(st, iou); torch.cuda.synchronize()
    m = O.iou2d(boxes[0], boxes[0]); ref = O.differentiable_nms(scores[0], m)
    print(N, "ok:", np.array_equal(out[0][0].cpu().numpy(), ref["prob"]), np.array_equal(out2[0][0].cpu().numpy(), ref["prob"]), flush=True)
