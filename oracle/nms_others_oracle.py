"""Pure-Python restatement of lib/nms_others.py (Soft-NMS with index tracking, Girshick NMS with
`shift`) -- TEST INFRASTRUCTURE, small cases only.  Parity status: pinned against
tests/golden/misc.npz (reference outputs captured by tests/golden/make_golden.py)."""
import math

import numpy as np


def soft_nms(boxes, sigma=0.5, Nt=0.4, threshold=0.001, method=0, shift=1):
    """lib/nms_others.py:6-116 navneeth_soft_nms.  Works on a float64 copy; returns kept original indices."""
    b = np.array(boxes, dtype=np.float64, copy=True)
    n = b.shape[0]
    idx = list(range(n))
    i = 0
    while i < n:      # `for i in range(N)` (:18) keeps the initial N, but iterations past the live N are no-ops
        # select the max-score box among [i, n) and swap it into slot i (:19-60)
        maxpos = i
        maxscore = b[i, 4]
        for pos in range(i + 1, n):
            if maxscore < b[pos, 4]:
                maxscore = b[pos, 4]
                maxpos = pos
        b[[i, maxpos]] = b[[maxpos, i]]
        idx[i], idx[maxpos] = idx[maxpos], idx[i]
        tx1, ty1, tx2, ty2 = b[i, 0], b[i, 1], b[i, 2], b[i, 3]
        pos = i + 1
        while pos < n:                                     # :64-112
            x1, y1, x2, y2 = b[pos, 0], b[pos, 1], b[pos, 2], b[pos, 3]
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
                    b[pos, 4] = weight * b[pos, 4]
                    if b[pos, 4] < threshold:              # discard: swap with the last live box (:98-110)
                        b[pos] = b[n - 1]
                        idx[n - 1], idx[pos] = idx[pos], idx[n - 1]
                        n -= 1
                        pos -= 1
            pos += 1
        i += 1
    return np.asarray(idx[:n], dtype=np.int64)


def girshick_nms(dets, thresh, shift=1):
    """lib/nms_others.py:119-150, including its keep_orig = i + N_dropped bookkeeping (:135,148)."""
    dets = np.asarray(dets)
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
        n_dropped = order.shape[0] - inds.shape[0]          # always 0 (:148) -- kept as in the reference
    return np.asarray(keep_orig, dtype=np.int64)
