import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import groomed_nms_amd as G
from groomed_nms_amd import synthetic
B, N = 8, 500
boxes_np, scores_np = synthetic.batch_2d(5, B, N, "clustered", per=25)
boxes = torch.from_numpy(boxes_np).cuda()
class Layer(torch.nn.Module):
    def forward(self, scores, boxes):
        return G.differentiable_nms_with_iou2d_batched(scores, boxes)[0]
layer = Layer()
s_static = torch.from_numpy(scores_np).cuda().requires_grad_(True)
graphed = torch.cuda.make_graphed_callables(layer, (s_static, boxes))
w = torch.rand((B, N), device="cuda")
for trial in range(3):
    _, sc = synthetic.batch_2d(50 + trial, B, N, "clustered", per=25)
    s1 = torch.from_numpy(sc).cuda().requires_grad_(True)
    s2 = torch.from_numpy(sc).cuda().requires_grad_(True)
    p1 = graphed(s1, boxes); (p1 * w).sum().backward()
    p2 = layer(s2, boxes); (p2 * w).sum().backward()
    print("trial", trial, torch.equal(p1, p2), torch.equal(s1.grad, s2.grad), float(p1.sum()))
def step(fn):
    s = s_static
    s.grad = None
    p = fn(s, boxes)
    torch.autograd.backward(p, w)
for name, fn in (("eager", layer), ("graphed", graphed)):
    for _ in range(20): step(fn)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): step(fn)
    torch.cuda.synchronize()
    print("%-8s %.1f us/step (B=%d, N=%d)" % (name, (time.perf_counter() - t0) / 300 * 1e6, B, N))
