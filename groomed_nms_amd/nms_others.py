"""Comparison baselines of the reference's lib/nms_others.py (pure-Python there as well; test-only callers,
test/test_differentiable_nms_forward.py:111-114): Soft-NMS with index tracking and Girshick NMS with `shift`."""
import math

import numpy as np

__all__ = ["navneeth_soft_nms", "girshick_nms"]


def navneeth_soft_nms(boxes, sigma=0.5, Nt=0.4, threshold=0.001, method=0, shift=1):
    """Soft-NMS (Bodla et al.) with the index tracking of lib/nms_others.py:6-116; returns the kept original indices in
    the reference's slot order.  method 0 hard, 1 linear, 2 gaussian.  Formulated on a slot permutation: rows never move,
    `slots[p]` names the box sitting in slot p (the reference swaps rows and its keep_orig array in lockstep, :45-60,
    :98-110), and -- unlike the reference -- the caller's array is left untouched."""
    geom = np.asarray(boxes, dtype=np.float64)[:, :4]
    score = np.array(np.asarray(boxes)[:, 4], dtype=np.float64)
    n = geom.shape[0]
    slots = list(range(n))
    live = n
    for i in range(n):
        if i >= live:
            break                                   # the reference keeps iterating over dead slots, a no-op (:18)
        # slot of the best remaining score; the first maximum wins (strict '<' at :32)
        best = i
        for p in range(i + 1, live):
            if score[slots[best]] < score[slots[p]]:
                best = p
        slots[i], slots[best] = slots[best], slots[i]
        ax1, ay1, ax2, ay2 = geom[slots[i]]
        area_a = (ax2 - ax1 + shift) * (ay2 - ay1 + shift)
        p = i + 1
        while p < live:
            j = slots[p]
            bx1, by1, bx2, by2 = geom[j]
            iw = min(ax2, bx2) - max(ax1, bx1) + shift
            ih = min(ay2, by2) - max(ay1, by1) + shift
            if iw > 0 and ih > 0:
                ov = iw * ih / float(area_a + (bx2 - bx1 + shift) * (by2 - by1 + shift) - iw * ih)
                if method == 1:
                    weight = 1 - ov if ov > Nt else 1
                elif method == 2:
                    weight = math.exp(-(ov * ov) / sigma)
                else:
                    weight = 0 if ov > Nt else 1
                score[j] = weight * score[j]
                if score[j] < threshold:            # discard: the last live slot takes this place and is examined next
                    slots[p], slots[live - 1] = slots[live - 1], slots[p]
                    live -= 1
                    continue
            p += 1
    return np.asarray(slots[:live], dtype=np.int64)


def girshick_nms(dets, thresh, shift=1):
    """lib/nms_others.py:119-150: greedy NMS with a configurable pixel `shift`.  The reference returns `keep_orig`, built
    as kept index + N_dropped where N_dropped is recomputed after every round as len(order) - len(inds) on the ALREADY
    filtered order (:148) -- which is always 0 -- so the list equals the kept indices; reproduced as such."""
    from .nms._host import greedy_nms
    return [np.int64(i) for i in greedy_nms(np.asarray(dets), thresh, shift=shift, rule="le_keep")]
