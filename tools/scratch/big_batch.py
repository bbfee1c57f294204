"""B*N*N past 2^31 / 2^32 elements: the last image of a big batch must equal the same image run alone."""
import sys, torch, numpy as np
sys.path.insert(0, ".")
import groomed_nms_amd as G
from groomed_nms_amd import synthetic, overlaps

def case(dim, B, N):
    if dim == 2:
        boxes, scores = synthetic.batch_2d(5, B, N, "clustered", per=48)
        fn = G.differentiable_nms_with_iou2d_batched
    else:
        boxes, scores = synthetic.batch_3d(5, B, N, True)
        fn = G.differentiable_nms_with_iou3d_batched
    bt = torch.from_numpy(boxes).cuda(); st = torch.from_numpy(scores).cuda().requires_grad_(True)
    w = torch.rand((B, N), device="cuda")
    out = fn(st, bt)
    (out[0] * w).sum().backward()
    iou = out[-1] if isinstance(out[-1], torch.Tensor) and out[-1].dim() == 3 else None
    for img in (B - 1, B // 2):
        s1 = torch.from_numpy(scores[img:img + 1]).cuda().requires_grad_(True)
        o1 = fn(s1, bt[img:img + 1])
        (o1[0] * w[img:img + 1]).sum().backward()
        for a, b in zip(out[:6], o1[:6]):
            assert torch.equal(a[img:img + 1], b), (dim, B, N, img)
        assert torch.equal(st.grad[img:img + 1], s1.grad)
        if iou is not None:
            assert torch.equal(iou[img], o1[-1][0])
    print("ok", dim, B, N, "matrix", None if iou is None else tuple(iou.shape), "valid", int(out[4].sum()))
    del out, iou
    torch.cuda.empty_cache()

case(2, 32, 16384)
case(3, 32, 16384)
case(2, 72, 8192)
