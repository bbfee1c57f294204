import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import groomed_nms_amd as G
from groomed_nms_amd import synthetic, overlaps, _lib
boxes, scores = synthetic.batch_2d(3, 2, 300, "clustered", per=20)
bt = torch.from_numpy(boxes).cuda(); st = torch.from_numpy(scores).cuda()
ref = G.differentiable_nms_with_iou2d_batched(st, bt)
def same(out, what):
    for a, b in zip(out[:6], ref[:6]):
        assert torch.equal(a, b), what
    print("ok:", what)
# non-contiguous boxes / scores
big = torch.zeros((2, 300, 6), device="cuda"); big[..., 1:5] = bt
same(G.differentiable_nms_with_iou2d_batched(st, big[..., 1:5]), "strided boxes view")
sbig = torch.zeros((2, 600), device="cuda"); sbig[:, ::2] = st
same(G.differentiable_nms_with_iou2d_batched(sbig[:, ::2], bt), "strided scores view")
same(G.differentiable_nms_with_iou2d_batched(st.double(), bt.double()), "float64 inputs")
# matrix with padded leading dimension
iou = overlaps.iou_batched(bt)
pad = torch.zeros((2, 300, 304), device="cuda"); pad[..., :300] = iou
same(G.differentiable_nms_batched(st, pad[..., :300]), "matrix with ld=304")
# iou requiring grad
iou_g = iou.clone().requires_grad_(True); s_g = st.clone().requires_grad_(True)
out = G.differentiable_nms_batched(s_g, iou_g)
(out[0] * torch.rand_like(out[0])).sum().backward()
print("ok: grad_iou", float(iou_g.grad.abs().sum()) > 0, iou_g.grad.shape)
# single-image API: numpy float64 in, CPU out
v, iv, p = G.differentiable_nms(scores[0].astype(np.float64), iou[0].cpu().numpy())
assert not p.is_cuda and p.shape == (300,) and v.dtype == torch.int64
print("ok: numpy in -> cpu out", len(v), len(iv))
# empty
v, iv, p = G.differentiable_nms(torch.zeros(0, device="cuda"), torch.zeros((0, 0), device="cuda"))
print("ok: N=0", v.shape, iv.shape, p.shape)
out = G.differentiable_nms_with_iou2d_batched(torch.zeros((0, 5), device="cuda"), torch.zeros((0, 5, 4), device="cuda"))
print("ok: B=0", out[0].shape)
try:
    G.differentiable_nms(torch.rand(5, device="cuda"), torch.rand((5, 5), device="cuda"), pruning_method="bogus")
except NotImplementedError as e:
    print("ok: NotImplementedError", e)
try:
    G.differentiable_nms_batched(torch.rand((1, 20000), device="cuda"), torch.rand((1, 4, 4), device="cuda"))
except Exception as e:
    print("ok: too large / bad shape ->", type(e).__name__, str(e)[:80])
