"""Round-2 diagnostics on the GPU box: (1) soft-sort gradient error against the goldens and the fp64 oracle adjoint,
(2) guard-band statistics of the 3D NMS overlap (pairs inside the band, decision flips of the un-guarded expression)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import groomed_nms_amd as G                    # noqa: E402
from groomed_nms_amd import overlaps, synthetic  # noqa: E402
from oracle import oracle as O                 # noqa: E402
from conftest import Golden                    # noqa: E402

out = {}
g = Golden("misc.npz")
tags = sorted({k.split("/")[0] for k in g.keys if k.startswith("softsort_")})
rows = []
for tag in tags:
    s, m, t, w = g[f"{tag}/scores"], g[f"{tag}/iou"], float(g[f"{tag}/temperature"]), g[f"{tag}/w"]
    for mt, kw in (("gm", dict(group_boxes=True, mask_group_boxes=True)), ("gu", dict(group_boxes=True, mask_group_boxes=False)),
                   ("un", dict(group_boxes=False))):
        st = torch.from_numpy(s).cuda().requires_grad_(True)
        mtx = torch.from_numpy(m).cuda().requires_grad_(True)
        v, iv, p = G.differentiable_nms(st, mtx, sorting_method="soft", sorting_temperature=t, temperature=0.1, **kw)
        (p * torch.from_numpy(w).cuda()).sum().backward()
        gs = st.grad.cpu().numpy()
        gold = g[f"{tag}/{mt}/grad_scores"]
        ref = O.differentiable_nms(s, m, sorting_method="soft", sorting_temperature=t, temperature=0.1, grad_prob=w, **kw)
        rows.append(dict(tag=tag, mode=mt, n=len(s), temperature=t, max_abs_grad=float(np.abs(gold).max()),
                         err_vs_golden=float(np.abs(gs - gold).max()), err_vs_fp64_oracle=float(np.abs(gs - ref["grad_scores"]).max()),
                         golden_vs_fp64_oracle=float(np.abs(gold - ref["grad_scores"]).max())))
out["soft_sort_grad"] = rows

band = []
for N, clustered in ((4096, True), (4096, False), (8192, True)):
    par, sc = synthetic.batch_3d(31 + N + int(clustered), 1, N, clustered=clustered, per=64)
    pt = torch.from_numpy(par).cuda()
    exact = overlaps.iou3d_batched(pt, from_params=True, nms_overlap=True)[0]
    guarded = overlaps.iou3d_batched(pt, from_params=True, nms_overlap=True, nms_threshold=0.4)[0]
    fast = overlaps.iou3d_batched(pt, from_params=True, nms_overlap=True, nms_threshold=-100.0)[0]      # band far away: the raw re-associated values
    d = (fast - exact).abs()
    band.append(dict(N=N, clustered=clustered, max_abs_fast_minus_exact=float(d.max()),
                     pairs_within_8e6=int(((exact - 0.4).abs() <= 8e-6).sum()), pairs_within_2e6=int(((exact - 0.4).abs() <= 2e-6).sum()),
                     flips_unguarded=int(((fast > 0.4) != (exact > 0.4)).sum()), flips_guarded=int(((guarded > 0.4) != (exact > 0.4)).sum()),
                     guarded_equals_exact_in_band=bool(torch.equal(guarded[(exact - 0.4).abs() <= 4e-6], exact[(exact - 0.4).abs() <= 4e-6]))))
out["guard_band_3d"] = band
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "diag_r02.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))
