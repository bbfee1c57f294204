"""The after-RPN part of one training step of the reference, kept on the device end to end (one JSON line):
   decode (lib/rpn_util.py:872) -> top-500 of the foreground scores (lib/loss/rpn_3d.py:731-737) -> IoU matrix + GrooMeD-NMS
   (:772-791) -> best box per ground truth (:801-826) -> after-NMS AP loss (:1117-1131) -> backward to the scores.
Synthetic inputs of the reference's shapes (B images, 32 x 110 x 36 anchors, <= 500 boxes into the NMS).  usage: python tools/pipeline_demo.py"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import groomed_nms_amd as G  # noqa: E402
from groomed_nms_amd import proposals as PR  # noqa: E402
from groomed_nms_amd.aploss import ap_loss_batched  # noqa: E402


def main():
    rng = np.random.default_rng(0)
    B, A, F, K, M = 8, 32 * 110 * 36, 2000, 500, 8
    dev = torch.device("cuda")
    ctr = np.stack([rng.uniform(0, 1760, A), rng.uniform(0, 512, A)], 1)
    wh = rng.uniform(16, 200, (A, 2))
    anchors = torch.from_numpy(np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(np.float32)).to(dev)
    deltas = torch.from_numpy((rng.standard_normal((B, A, 4)) * 0.2).astype(np.float32)).to(dev)
    logits = torch.from_numpy(rng.standard_normal((B, A)).astype(np.float32)).to(dev).requires_grad_(True)
    fg = torch.from_numpy(np.stack([rng.choice(A, F, replace=False) for _ in range(B)]).astype(np.int32)).to(dev)
    fg_num = torch.full((B,), F, dtype=torch.int32, device=dev)
    # 3D side for the target assignment: cuboids of the selected boxes and of M ground truths per image
    def cuboids(n):
        return np.stack([rng.uniform(-20, 20, n), rng.uniform(0.5, 2.5, n), rng.uniform(6, 50, n), rng.uniform(1.4, 2, n),
                         rng.uniform(1.3, 2, n), rng.uniform(3, 5, n), rng.uniform(-3.1, 3.1, n)], 1).astype(np.float32)
    pred3d = torch.from_numpy(np.stack([cuboids(A) for _ in range(1)]).repeat(B, 0)).to(dev)
    gt3d = torch.from_numpy(np.stack([cuboids(M) for _ in range(B)])).to(dev)
    p2 = torch.tensor([[721.5, 0, 609.6, 44.9], [0, 721.5, 172.9, 0.22], [0, 0, 1, 0.0027], [0, 0, 0, 1]], device=dev).repeat(B, 1, 1)
    gt2d = PR.projected_boxes_2d(gt3d, p2, 1.0)

    def step():
        scores = torch.sigmoid(logits)                                                  # stock PyTorch (the network's output)
        boxes = PR.bbox_transform_inv(anchors, deltas, means=[0, 0, 0, 0], stds=[0.1, 0.1, 0.2, 0.2])
        idx, num, s_sel, b_sel = PR.select_topk(scores.detach(), K, fg, fg_num, boxes)
        s_sel = torch.gather(scores, 1, idx.clamp(min=0)) * (idx >= 0)                   # differentiable gather of the same boxes
        prob = G.differentiable_nms_with_iou2d_batched(s_sel, b_sel, counts=num)[0]
        p3 = torch.gather(pred3d, 1, idx.clamp(min=0).unsqueeze(-1).expand(-1, -1, 7))
        targets, best, _ = PR.best_targets(p3, b_sel, gt3d, gt2d, 0.1, num, None)
        loss = ap_loss_batched(prob, targets, counts=num).mean()
        logits.grad = None
        loss.backward()
        return loss

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(json.dumps({"pipeline": "decode -> top-K -> IoU + GrooMeD-NMS -> best targets -> AP loss -> backward", "images": B, "anchors_per_image": A,
                      "foreground_candidates": F, "boxes_into_nms": K, "ms_per_step": round(dt * 1e3, 4), "loss": round(float(loss.detach()), 6),
                      "grad_nonzero": int((logits.grad != 0).sum())}))


if __name__ == "__main__":
    main()
