import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import groomed_nms_amd as G
from groomed_nms_amd import synthetic, _lib
from groomed_nms_amd._lib import GnmsParams, ptr, stream_ptr, check
lib = _lib.load()
B, N = 8, 4096
P = GnmsParams(); lib.gnms_default_params(ctypes.byref(P))
ws = torch.empty((lib.gnms_workspace_bytes(B, N, ctypes.byref(P)),), dtype=torch.uint8, device="cuda")
prob = torch.empty((B, N), device="cuda")
rng = np.random.default_rng(0)
def boxes_case(kind):
    if kind == "clustered":
        return synthetic.batch_2d(1000, B, N, "clustered")[0]
    b = np.zeros((B, N, 4), np.float32)
    if kind == "spread_x":      # disjoint boxes along x: ~1 active row per tile
        x = rng.permuted(np.tile(np.arange(N, dtype=np.float32) * 20, (B, 1)), axis=1)
        b[..., 0] = x; b[..., 2] = x + 10; b[..., 3] = 10
    elif kind == "same_x":      # every box overlaps every hull in x and y: no culling
        y = rng.permuted(np.tile(np.arange(N, dtype=np.float32) * 0.01, (B, 1)), axis=1)
        b[..., 1] = y; b[..., 3] = y + 50; b[..., 2] = 30
    return b
for kind in ("clustered", "spread_x", "same_x"):
    bx = torch.from_numpy(boxes_case(kind)).cuda()
    sc = torch.from_numpy(synthetic.batch_2d(1, B, N, "uniform")[1]).cuda()
    check(lib.gnms_forward_from_boxes(ptr(bx), ptr(sc), B, N, None, ctypes.byref(P), ptr(prob), None, None, None, None, None, ptr(ws), ws.numel(), stream_ptr()), "f")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5): lib.gnms_profile_bitmask_boxes(ptr(bx), B, N, None, P.nms_threshold, ptr(ws), ws.numel(), stream_ptr())
    e0.record()
    for _ in range(50): lib.gnms_profile_bitmask_boxes(ptr(bx), B, N, None, P.nms_threshold, ptr(ws), ws.numel(), stream_ptr())
    e1.record(); torch.cuda.synchronize()
    print("%-10s bitmask_boxes %.1f us" % (kind, e0.elapsed_time(e1) / 50 * 1e3))
