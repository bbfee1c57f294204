import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from groomed_nms_amd import _lib, synthetic, groomed_nms as G
from groomed_nms_amd._lib import ptr, check, GnmsParams
lib = _lib.load()
B, N = 8, 4096
for kind in ("clustered", "uniform"):
    b, s = synthetic.batch_2d(1, B, N, kind)
    bt, st = torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda()
    P = GnmsParams(); lib.gnms_default_params(ctypes.byref(P))
    nbytes = lib.gnms_workspace_bytes(B, N, ctypes.byref(P))
    ws = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    prob = torch.empty((B, N), device="cuda")
    check(lib.gnms_forward_from_boxes(ptr(bt), ptr(st), B, N, None, ctypes.byref(P), ptr(prob), None, None, None, None, None, ptr(ws), nbytes, None), "f")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5): check(lib.gnms_profile_bitmask_boxes(ptr(bt), B, N, None, 0.4, ptr(ws), nbytes, None), "p")
    e0.record()
    for _ in range(50): check(lib.gnms_profile_bitmask_boxes(ptr(bt), B, N, None, 0.4, ptr(ws), nbytes, None), "p")
    e1.record(); e1.synchronize()
    print(kind, "bitmask_boxes back-to-back us", e0.elapsed_time(e1) / 50 * 1e3)
