"""Times one step (IoU matrix + differentiable_nms forward + backward w.r.t. scores) for every mode of the layer, plus the
classical `_nms` C entry and soft sort, on one GPU.  usage: python tools/mode_times.py [--boxes 4096] [--batch 8]
Prints one JSON line per configuration (HIP events around 20 steps after 5 warm-ups)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import groomed_nms_amd as G  # noqa: E402
from groomed_nms_amd import overlaps, synthetic  # noqa: E402
from groomed_nms_amd.nms import gpu_nms  # noqa: E402

MODES = {
    "grouped+masked linear (default)": dict(),
    "grouped+masked linear, group_size 2": dict(group_size=2),
    "grouped+masked linear, sorted output": dict(return_sorted_prob=True),
    "grouped+masked sigmoidal": dict(pruning_method="sigmoidal", temperature=0.1),
    "grouped+masked soft_nms": dict(pruning_method="soft_nms", temperature=0.5),
    "grouped unmasked linear": dict(mask_group_boxes=False),
    "grouped unmasked sigmoidal": dict(mask_group_boxes=False, pruning_method="sigmoidal", temperature=0.1),
    "ungrouped linear": dict(group_boxes=False),
    "ungrouped sigmoidal": dict(group_boxes=False, pruning_method="sigmoidal", temperature=0.1),
}


def timed(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--boxes", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--kind", default="clustered")
    ap.add_argument("--only", default="", help="substring filter on the mode name")
    a = ap.parse_args()
    B, N = a.batch, a.boxes
    boxes_np, scores_np = synthetic.batch_2d(1000, B, N, a.kind)
    boxes = torch.from_numpy(boxes_np).cuda()
    scores = torch.from_numpy(scores_np).cuda().requires_grad_(True)
    w = torch.linspace(-1.0, 2.0, N, device="cuda").repeat(B, 1).contiguous()
    iou_buf = torch.empty((B, N, N), dtype=torch.float32, device="cuda")
    for name, kw in MODES.items():
        if a.only and a.only not in name:
            continue

        def step():
            prob = G.differentiable_nms_with_iou2d_batched(scores, boxes, iou_out=iou_buf, **kw)[0]
            scores.grad = None
            torch.autograd.backward(prob, w)

        def step_matrix_in():
            prob = G.differentiable_nms_batched(scores, overlaps.iou_batched(boxes, out=iou_buf), **kw)[0]
            scores.grad = None
            torch.autograd.backward(prob, w)
        t1, t2 = timed(step), timed(step_matrix_in)
        print(json.dumps({"mode": name, "B": B, "N": N, "ms_per_step_boxes_in": round(t1, 4), "boxes_per_s": round(B * N / (t1 * 1e-3)),
                          "ms_per_step_matrix_in": round(t2, 4)}), flush=True)
    if a.only:
        return
    # classical NMS through the reference's C symbol (host pointers in, blocking): lib/nms/gpu_nms.pyx:16-31
    dets = np.concatenate([boxes_np[0], scores_np[0][:, None]], 1).astype(np.float32)
    import time
    for _ in range(3):
        keep = gpu_nms(dets, 0.4)
    t0 = time.perf_counter()
    for _ in range(20):
        keep = gpu_nms(dets, 0.4)
    t = (time.perf_counter() - t0) / 20
    print(json.dumps({"mode": "_nms (host pointers, blocking, 1 image)", "N": N, "ms_per_call": round(t * 1e3, 4), "kept": len(keep),
                      "boxes_per_s": round(N / t)}), flush=True)
    if N <= 4096:
        s1 = torch.from_numpy(np.sort(scores_np[0])[::-1].copy()).cuda()
        iou1 = overlaps.iou_batched(boxes[:1])[0]
        t = timed(lambda: G.soft_sort(s1, iou1, temperature=0.01), iters=10, warm=3)
        print(json.dumps({"mode": "soft_sort (C@s, C@iou on fp32 MFMA)", "N": N, "ms_per_call": round(t, 4),
                          "gemm_tflops": round(2.0 * N ** 3 / (t * 1e-3) / 1e12, 1)}), flush=True)


if __name__ == "__main__":
    main()
