import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import groomed_nms_amd as G
from groomed_nms_amd import synthetic, overlaps
for N in (500, 4096):
    boxes_np, scores_np = synthetic.batch_2d(7, 1, N, "clustered")
    bt = torch.from_numpy(boxes_np).cuda(); st = torch.from_numpy(scores_np).cuda()
    def one_call():
        with torch.no_grad():
            return G.differentiable_nms_with_iou2d_batched(st, bt)
    def ref_api():                      # the reference's call shape: matrix built, then differentiable_nms(scores, iou) incl. its host sync
        with torch.no_grad():
            iou = overlaps.iou(bt[0], bt[0])
            return G.differentiable_nms(st[0], iou)
    def numpy_api():
        iou = overlaps.iou(boxes_np[0], boxes_np[0])      # NumPy in -> NumPy out (GPU in between)
        return G.differentiable_nms(scores_np[0].astype(np.float64), iou)
    for name, fn in (("batched one-call, B=1, no sync", one_call), ("iou + differentiable_nms (GPU tensors, host sync)", ref_api), ("NumPy in/out (PCIe both ways)", numpy_api)):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): fn()
        torch.cuda.synchronize()
        print("N=%d %-52s %.1f us/image" % (N, name, (time.perf_counter() - t0) / 100 * 1e6))
