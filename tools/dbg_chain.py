"""Developer check of the multi-workgroup leader scan: one-call and two-call layers against the oracle at a few sizes."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import groomed_nms_amd as G
from groomed_nms_amd import synthetic, overlaps
from oracle import oracle as O
for (B, N, kind) in ((2, 1500, "uniform"), (8, 4096, "uniform"), (8, 4096, "clustered"), (3, 2500, "clustered"), (2, 8192, "uniform")):
    boxes, scores = synthetic.batch_2d(5, B, N, kind)
    bt = torch.from_numpy(boxes).cuda(); st = torch.from_numpy(scores).cuda()
    counts = torch.tensor([N] + [max(1, N // 2 + 7)] * (B - 1), dtype=torch.int32).cuda()
    for rep in range(3):
        out = G.differentiable_nms_with_iou2d_batched(st, bt, counts=counts)
        iou = overlaps.iou_batched(bt)
        out2 = G.differentiable_nms_batched(st, iou, counts=counts)
    torch.cuda.synchronize()
    ok = all(torch.equal(a, b) for a, b in zip(out[:6], out2[:6]))
    n1 = int(counts[1]) if B > 1 else N
    b = min(1, B - 1)
    m = O.iou2d(boxes[b][:n1], boxes[b][:n1])
    ref = O.differentiable_nms(scores[b][:n1], m)
    okr = np.array_equal(out[0][b, :n1].cpu().numpy(), ref["prob"]) and out[2][b, :int(out[4][b])].tolist() == list(ref["valid"])
    print(B, N, kind, "one-call == two-call:", ok, " == oracle:", okr, flush=True)
    if not ok:
        for i, (a, b2) in enumerate(zip(out[:6], out2[:6])):
            if not torch.equal(a, b2):
                d = (a != b2).nonzero()
                print("  output", i, "differs at", d.shape[0], "entries; first", d[:4].tolist(), a[tuple(d[0])].item(), b2[tuple(d[0])].item())
        for bb in range(B):
            nn = int(counts[bb]); mm = O.iou2d(boxes[bb][:nn], boxes[bb][:nn]); rr = O.differentiable_nms(scores[bb][:nn], mm)
            print("  image", bb, "one-call prob ok", np.array_equal(out[0][bb, :nn].cpu().numpy(), rr["prob"]), "two-call prob ok", np.array_equal(out2[0][bb, :nn].cpu().numpy(), rr["prob"]))
