import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import groomed_nms_amd as G
from groomed_nms_amd import synthetic, overlaps
cfgs = ((3, 500, {}), (2, 4096, {}), (1, 64, {}), (2, 1001, dict(nms_threshold=0.6)), (2, 2300, dict(group_size=3)),
        (2, 900, dict(nms_threshold=0.5)), (2, 900, dict(nms_threshold=0.005)), (2, 900, dict(nms_threshold=0.2)), (8, 8192, {}))
i = int(sys.argv[1])
B, N, kw = cfgs[i]
par, scores = synthetic.batch_3d(11, B, N, clustered=True, per=16)
pt = torch.from_numpy(par).cuda()
counts = torch.tensor([N] + [max(1, N // 2)] * (B - 1), dtype=torch.int32).cuda()
s1 = torch.from_numpy(scores).cuda().requires_grad_(True)
s2 = torch.from_numpy(scores).cuda().requires_grad_(True)
print("cfg", i, B, N, kw, flush=True)
out1 = G.differentiable_nms_with_iou3d_batched(s1, pt, counts=counts, **kw)
torch.cuda.synchronize(); print(" one-call ok", flush=True)
ov = overlaps.iou3d_batched(pt, from_params=True, nms_overlap=True, nms_threshold=kw.get("nms_threshold", 0.4))
torch.cuda.synchronize(); print(" overlap ok", flush=True)
out2 = G.differentiable_nms_batched(s2, ov, counts=counts, **kw)
torch.cuda.synchronize(); print(" matrix-in ok", flush=True)
print(" equal:", all(torch.equal(a, b) or torch.allclose(a, b, atol=0, rtol=0, equal_nan=True) for a, b in zip(out1[:6], out2)))
