#!/usr/bin/env python3
"""The reference's C symbol `_nms` (lib/nms/gpu_nms.hpp:1-2: host pointers in and out, blocking) through the Python wrapper gpu_nms: time per
call at the sizes the reference calls it with (500 ... every anchor of an image), JSON lines.   python tools/nms_host.py"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from groomed_nms_amd import synthetic          # noqa: E402
from groomed_nms_amd.nms import gpu_nms        # noqa: E402

for n in (500, 2000, 4096, 16384, 126720):
    rng = np.random.default_rng(n)
    boxes = synthetic.clustered_boxes_2d(rng, n, 64)
    scores = np.sort(synthetic.tie_free_scores(rng, n))[::-1]
    dets = np.ascontiguousarray(np.concatenate([boxes, scores[:, None]], 1).astype(np.float32))
    for _ in range(5):
        keep = gpu_nms(dets, 0.4)
    reps = 200 if n <= 16384 else 10
    best = 1e9
    for w in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            keep = gpu_nms(dets, 0.4)
        best = min(best, (time.perf_counter() - t0) / reps * 1e6)
    # the same boxes in random order: the wrapper's argsort + gather as the reference's gpu_nms.pyx has them (sorted input skips both, round 6)
    sh = dets[np.random.default_rng(1).permutation(n)]
    for _ in range(3):
        gpu_nms(sh, 0.4)
    t0 = time.perf_counter()
    for _ in range(max(reps // 4, 3)):
        gpu_nms(sh, 0.4)
    unsorted = (time.perf_counter() - t0) / max(reps // 4, 3) * 1e6
    print(json.dumps({"what": "gpu_nms(dets, 0.4): _nms, host pointers, blocking", "n": n, "kept": int(len(keep)), "us_per_call": round(best, 1),
                      "us_per_call_unsorted_input": round(unsorted, 1)}))
