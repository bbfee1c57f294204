import sys, time, torch, json
sys.path.insert(0,'/root/repo')
from groomed_nms_amd import groomed_nms as GN, synthetic
def time_it(fn, n=2000, warm=100):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
for N in (1024, 512, 256, 128, 256, 512, 1024):
    b, s = synthetic.batch_2d(1, 8, N, "clustered")
    boxes = torch.from_numpy(b).cuda(); scores = torch.from_numpy(s).cuda().requires_grad_(True)
    w = torch.ones_like(scores); buf = torch.empty((8, N, N), device="cuda")
    def step():
        prob = GN.differentiable_nms_with_iou2d_batched(scores, boxes, iou_out=buf, index_lists=False)[0]
        scores.grad = None
        torch.autograd.backward(prob, w)
    def fwd():
        GN.differentiable_nms_with_iou2d_batched(scores, boxes, iou_out=buf, index_lists=False)
    print(N, "step", round(time_it(step),1), "fwd only", round(time_it(fwd),1), flush=True)
