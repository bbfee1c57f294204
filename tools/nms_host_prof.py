"""gpu_nms at n = 4096 for the tracer (rocprofv3 --kernel-trace --stats -- python tools/nms_host_prof.py)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from groomed_nms_amd import synthetic          # noqa: E402
from groomed_nms_amd.nms import gpu_nms        # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(n)
boxes = synthetic.clustered_boxes_2d(rng, n, 64)
scores = np.sort(synthetic.tie_free_scores(rng, n))[::-1]
dets = np.ascontiguousarray(np.concatenate([boxes, scores[:, None]], 1).astype(np.float32))
for _ in range(300 if n <= 16384 else 12):
    gpu_nms(dets, 0.4)
