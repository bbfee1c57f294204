"""Developer tool: phase breakdown (s_memtime ticks of thread 0 of image 0) of the per-image kernels.  Needs a build with
GNMS_EXTRA_FLAGS=-DGNMS_TIMING (python -m groomed_nms_amd.build --force); slots are the GNMS_TACC(n) markers in nms_kernels.h."""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from groomed_nms_amd import _lib, synthetic          # noqa: E402
from groomed_nms_amd._lib import GnmsParams, ptr, check  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--boxes", type=int, default=4096)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--kind", default="clustered")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--lists", action="store_true", help="ask for the valid / invalid index lists (K6 then compacts and sorts)")
ap.add_argument("--counts", type=int, default=0, help="boxes per image actually used (ragged: counts[b] = this; 0 = all)")
ap.add_argument("--two-calls", action="store_true", help="gnms_iou2d, then the matrix-in layer gnms_forward: the chain kernels alone on the machine")
a = ap.parse_args()
lib = _lib.load()
B, N = a.batch, a.boxes
P = GnmsParams()
lib.gnms_default_params(ctypes.byref(P))
boxes_np, scores_np = synthetic.batch_2d(1000, B, N, a.kind)
boxes, scores = torch.from_numpy(boxes_np).cuda(), torch.from_numpy(scores_np).cuda()
nbytes = lib.gnms_workspace_bytes(B, N, ctypes.byref(P))
ws = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
prob = torch.empty((B, N), device="cuda")
iou = torch.empty((B, N, N), device="cuda")
vl = torch.empty((B, N), dtype=torch.int64, device="cuda") if a.lists else None
il = torch.empty((B, N), dtype=torch.int64, device="cuda") if a.lists else None
nv = torch.empty((B,), dtype=torch.int32, device="cuda") if a.lists else None
ni = torch.empty((B,), dtype=torch.int32, device="cuda") if a.lists else None
lp = lambda t: ptr(t) if t is not None else None
n4 = (4 * N + 255) // 256 * 256
off_gx = 15 * n4
names = {16: "one launch: chain waits for the sort flags", 17: "one launch: sort workgroup 0, start to flag", 18: "one launch: last table workgroup, start to flag (incl. its wait for the sort)", 21: "resolve (wave 15): table words to registers", 22: "resolve (wave 15): dirty check + AND pass", 23: "resolve (wave 15): in-block fixed point + publish", 24: "resolve (wave 15): barrier", 20: "leaders_sym resolve rounds (count)", 5: "CHAIN leaders total", 6: "CHAIN attribute total", 7: "CHAIN groups total", 15: "CHAIN finalize total", 0: "leaders prologue (table 0)", 1: "leaders resolve / prefetch+far push", 2: "leaders barrier A", 3: "leaders near push + table store / sym scan: wait for the other workgroups",
         4: "leaders barrier B / sym scan: rem + fast-tail stage A", 8: "groups keys / fast tail: loads + cap check", 9: "groups radix / fast tail: store acks + barrier", 10: "groups runs", 11: "groups rescoring", 12: "finalize classify", 13: "finalize sort",
         14: "finalize output"}
tot = np.zeros(28, np.int64)
for rep in range(a.reps + 2):
    ws[off_gx:off_gx + 224].zero_()
    if a.two_calls:
        check(lib.gnms_iou2d(ptr(boxes), ptr(boxes), B, N, N, ptr(iou), N, None), "iou")
        check(lib.gnms_forward(ptr(scores), ptr(iou), B, N, N, None, ctypes.byref(P), ptr(prob), None, lp(vl), lp(il), lp(nv), lp(ni), ptr(ws), nbytes, None), "fwd")
    else:
        check(lib.gnms_forward_with_iou2d(ptr(boxes), ptr(scores), B, N, N, None, ctypes.byref(P), ptr(iou), ptr(prob), None, lp(vl), lp(il), lp(nv), lp(ni),
                                          ptr(ws), nbytes, None), "fwd")
    torch.cuda.synchronize()
    t = ws[off_gx:off_gx + 224].cpu().numpy().view(np.int64)
    if rep >= 2:
        tot += t
for k in range(28):
    if tot[k]:
        print("slot %2d %-40s %9.0f ticks" % (k, names.get(k, ""), tot[k] / a.reps))
