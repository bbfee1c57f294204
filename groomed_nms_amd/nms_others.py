"""Comparison baselines of the reference's lib/nms_others.py (pure-Python there as well; test-only callers,
test/test_differentiable_nms_forward.py:111-114): Soft-NMS with index tracking and Girshick NMS with `shift`."""
import math

import numpy as np

__all__ = ["navneeth_soft_nms", "girshick_nms"]


def navneeth_soft_nms(boxes, sigma=0.5, Nt=0.4, threshold=0.001, method=0, shift=1):
    """lib/nms_others.py:6-116.  Like the reference it reorders/rescoring `boxes` IN PLACE and returns the kept
    original indices.  method 0 hard, 1 linear, 2 gaussian."""
    n = boxes.shape[0]
    tracker = np.arange(n)
    live = n
    for i in range(n):
        if i >= live:                      # the reference keeps looping over dead slots; nothing happens there
            break
        best = i + int(np.argmax(boxes[i:live, 4])) if live > i else i
        if boxes[best, 4] <= boxes[i, 4]:
            best = i                       # strict '<' in the reference's scan (:32): first maximum wins
        boxes[[i, best]] = boxes[[best, i]]
        tracker[[i, best]] = tracker[[best, i]]
        tx1, ty1, tx2, ty2 = boxes[i, 0], boxes[i, 1], boxes[i, 2], boxes[i, 3]
        pos = i + 1
        while pos < live:
            x1, y1, x2, y2 = boxes[pos, 0], boxes[pos, 1], boxes[pos, 2], boxes[pos, 3]
            area = (x2 - x1 + shift) * (y2 - y1 + shift)
            iw = min(tx2, x2) - max(tx1, x1) + shift
            if iw > 0:
                ih = min(ty2, y2) - max(ty1, y1) + shift
                if ih > 0:
                    ua = float((tx2 - tx1 + shift) * (ty2 - ty1 + shift) + area - iw * ih)
                    ov = iw * ih / ua
                    if method == 1:
                        weight = 1 - ov if ov > Nt else 1
                    elif method == 2:
                        weight = math.exp(-(ov * ov) / sigma)
                    else:
                        weight = 0 if ov > Nt else 1
                    boxes[pos, 4] = weight * boxes[pos, 4]
                    if boxes[pos, 4] < threshold:
                        boxes[pos] = boxes[live - 1]
                        tracker[[live - 1, pos]] = tracker[[pos, live - 1]]
                        live -= 1
                        pos -= 1
            pos += 1
    return tracker[:live]


def girshick_nms(dets, thresh, shift=1):
    """lib/nms_others.py:119-150 (returns its keep_orig list, computed as in the reference: i + N_dropped)."""
    x1, y1, x2, y2, scores = dets[:, 0], dets[:, 1], dets[:, 2], dets[:, 3], dets[:, 4]
    areas = (x2 - x1 + shift) * (y2 - y1 + shift)
    order = scores.argsort()[::-1]
    keep_orig = []
    n_dropped = 0
    while order.size > 0:
        i = order[0]
        keep_orig.append(i + n_dropped)
        rest = order[1:]
        w = np.maximum(0.0, np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]) + shift)
        h = np.maximum(0.0, np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]) + shift)
        inter = w * h
        ovr = inter / (areas[i] + areas[rest] - inter)
        inds = np.where(ovr <= thresh)[0]
        order = order[inds + 1]
        n_dropped = order.shape[0] - inds.shape[0]
    return keep_orig
