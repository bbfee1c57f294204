"""Developer: phase ticks of bitmask_boxes_kernel (timing build): wave 0 and wave 15 of rank block 32 of image 0."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from groomed_nms_amd import _lib, synthetic
from groomed_nms_amd._lib import GnmsParams, ptr, check
lib = _lib.load()
B, N = 8, 4096
P = GnmsParams(); lib.gnms_default_params(ctypes.byref(P))
for kind in ("uniform", "clustered"):
    b, s = synthetic.batch_2d(1000, B, N, kind)
    boxes, scores = torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda()
    nbytes = lib.gnms_workspace_bytes(B, N, ctypes.byref(P))
    ws = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    prob = torch.empty((B, N), device="cuda"); iou = torch.empty((B, N, N), device="cuda")
    n4 = (4 * N + 255) // 256 * 256
    off = 14 * n4                                                # xsol
    reps = 20
    for rep in range(reps + 2):
        if rep == 2:
            ws[off:off + 128].zero_()
        check(lib.gnms_forward_with_iou2d(ptr(boxes), ptr(scores), B, N, N, None, ctypes.byref(P), ptr(iou), ptr(prob), None, None, None, None, None, ptr(ws), nbytes, None), "fwd")
        torch.cuda.synchronize()
    t = ws[off:off + 128].cpu().numpy().view(np.int64) / reps
    names = ["", "zero rowbuf + barrier", "column gathers + hull", "rows loop + scatter to LDS", "barrier (slowest wave)", "row write"]
    for w, base in (("wave 0", 0), ("wave 15", 8)):
        print(kind, w, " | ".join("%s %.0f" % (names[q], t[base + q]) for q in range(1, 6)))

    # (round 6) the launch's timeline: start / end of every workgroup (last call), 10-ns ticks of the constant-rate clock
    per = nbytes // B
    nwg = 32                                                   # workgroups per image (two rank blocks each since round 6)
    tl = np.stack([ws[b_ * per + off + 256: b_ * per + off + 256 + 16 * nwg].cpu().numpy().view(np.int64).reshape(nwg, 2) for b_ in range(B)])
    t0 = tl[:, :, 0].min()
    st, en = (tl[:, :, 0] - t0) / 100.0, (tl[:, :, 1] - t0) / 100.0
    print(kind, "timeline, us from the first workgroup's start: starts min / median / max %.2f / %.2f / %.2f, ends min / median / max %.2f / %.2f / %.2f, "
          "a workgroup's own time median / max %.2f / %.2f" % (st.min(), np.median(st), st.max(), en.min(), np.median(en), en.max(), np.median(en - st), (en - st).max()))
    for b_ in (0, 7):
        print(kind, "image %d: start by rank block (us):" % b_, " ".join("%.1f" % v for v in st[b_, ::2]), "| end:", " ".join("%.1f" % v for v in en[b_, ::2]))
