import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import groomed_nms_amd as G
from groomed_nms_amd import synthetic
from oracle import oracle as O
N = int(sys.argv[1]); kind = sys.argv[2] if len(sys.argv) > 2 else "uniform"
boxes, scores = synthetic.batch_2d(5, 1, N, kind)
bt = torch.from_numpy(boxes).cuda(); st = torch.from_numpy(scores).cuda()
out = G.differentiable_nms_from_boxes_batched(st, bt)
torch.cuda.synchronize()
m = O.iou2d(boxes[0], boxes[0]); ref = O.differentiable_nms(scores[0], m)
bad = np.nonzero(out[0][0].cpu().numpy() != ref["prob"])[0]
print(N, kind, "from-boxes == oracle:", bad.size == 0, bad[:6].tolist(), flush=True)
