"""Times the training tail (layer -> best targets -> AP loss, + backward) as one host call (proposals.training_tail, C++ binding), as the
three separate entries, and the AP loss alone -- eager, HIP events on the current stream.  One JSON line per configuration."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import groomed_nms_amd as G                                   # noqa: E402
from groomed_nms_amd import proposals as PR, synthetic       # noqa: E402
from groomed_nms_amd.aploss import ap_loss_batched            # noqa: E402


def timed(fn, k=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k


def main():
    dev = torch.device("cuda")
    for B, N, M in ((2, 500, 8), (8, 512, 8), (8, 4096, 16)):
        b2, sc = synthetic.batch_2d(3, B, N, "clustered", per=24)
        p3, _ = synthetic.batch_3d(4, B, N, clustered=True, per=24)
        o = np.argsort(-sc, axis=1, kind="stable")
        sc, b2, p3 = np.take_along_axis(sc, o, 1), np.take_along_axis(b2, o[:, :, None], 1), np.take_along_axis(p3, o[:, :, None], 1)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        s = t(sc).requires_grad_(True)
        bx, pp, gp, gb = t(b2), t(p3), t(p3[:, :M]), t(b2[:, :M])

        def one():
            s.grad = None
            PR.training_tail(s, bx, pp, gp, gb, 0.3)[0].sum().backward()

        def three():
            s.grad = None
            prob = G.differentiable_nms_with_iou2d_batched(s, bx, index_lists=False)[0]
            tg = PR.best_targets(pp, bx, gp, gb, 0.3)[0]
            ap_loss_batched(prob, tg).sum().backward()
        lg = t(sc).requires_grad_(True)
        tg0 = PR.best_targets(pp, bx, gp, gb, 0.3)[0]

        def ap():
            lg.grad = None
            ap_loss_batched(lg, tg0).sum().backward()
        print(json.dumps({"B": B, "N": N, "gt": M, "training_tail_fwd_bwd_ms": round(timed(one), 4), "three_calls_fwd_bwd_ms": round(timed(three), 4),
                          "aploss_fwd_bwd_ms": round(timed(ap), 4)}), flush=True)


if __name__ == "__main__":
    main()
