"""Times gnms_aploss (after-NMS AP loss) with HIP events on torch's current stream.
usage: python tools/aploss_time.py   -> one JSON line per configuration."""
import json
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from groomed_nms_amd.aploss import ap_loss_batched  # noqa: E402


def main():
    rng = np.random.default_rng(0)
    for B, N, F in ((8, 500, 20), (8, 500, 100), (64, 500, 20), (8, 2047, 64), (8, 2048, 64), (8, 4096, 64), (8, 4096, 1024), (8, 16384, 256), (8, 16384, 4096), (256, 512, 32)):
        lg = torch.from_numpy(rng.uniform(0, 1, (B, N)).astype(np.float32)).cuda().requires_grad_(True)
        tg = np.zeros((B, N), np.float32)
        for b in range(B):
            tg[b, rng.choice(N, F, replace=False)] = 1
        tg = torch.from_numpy(tg).cuda()
        for _ in range(5):
            ap_loss_batched(lg, tg).sum().backward()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        K = 50
        e0.record()
        for _ in range(K):
            loss = ap_loss_batched(lg, tg)
        e1.record()
        torch.cuda.synchronize()
        fwd = e0.elapsed_time(e1) / K
        e0.record()
        for _ in range(K):
            lg.grad = None
            ap_loss_batched(lg, tg).sum().backward()
        e1.record()
        torch.cuda.synchronize()
        both = e0.elapsed_time(e1) / K
        print(json.dumps({"B": B, "N": N, "positives": F, "loss_ms": round(fwd, 4), "loss_and_backward_ms": round(both, 4),
                          "boxes_per_s": round(B * N / (both * 1e-3))}))


if __name__ == "__main__":
    main()
