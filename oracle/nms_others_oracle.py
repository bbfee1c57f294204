"""Pure-Python restatement of lib/nms_others.py (Soft-NMS with index tracking, Girshick NMS with
`shift`) -- TEST INFRASTRUCTURE, small cases only.  Parity status: pinned against
tests/golden/misc.npz (reference outputs captured by tests/golden/make_golden.py)."""
import math

import numpy as np


def soft_nms(boxes, sigma=0.5, Nt=0.4, threshold=0.001, method=0, shift=1):
    """lib/nms_others.py:6-116 navneeth_soft_nms, restated with the overlaps of the selected box against all live boxes
    computed at once (they do not depend on the slot shuffling) and the slot bookkeeping done on index arrays.
    Returns the kept original indices in the reference's slot order."""
    b = np.array(boxes, dtype=np.float64, copy=True)
    geom, score = b[:, :4], b[:, 4].copy()
    n = len(b)
    where = np.arange(n)                       # where[p] = original index of the box in slot p   (keep_orig, :15)
    live = n
    for i in range(n):                         # :18 (iterations with i >= live do nothing)
        if i >= live:
            continue
        seg = score[where[i:live]]
        best = i + int(np.argmax(seg))         # np.argmax returns the FIRST maximum == the strict '<' scan of :29-34
        where[[i, best]] = where[[best, i]]    # :45-60
        a = geom[where[i]]
        w_all = np.minimum(a[2], geom[:, 2]) - np.maximum(a[0], geom[:, 0]) + shift        # against every box, by original index
        h_all = np.minimum(a[3], geom[:, 3]) - np.maximum(a[1], geom[:, 1]) + shift
        area_all = (geom[:, 2] - geom[:, 0] + shift) * (geom[:, 3] - geom[:, 1] + shift)
        area_a = (a[2] - a[0] + shift) * (a[3] - a[1] + shift)
        p = i + 1
        while p < live:                        # :64-112
            j = where[p]
            if w_all[j] > 0 and h_all[j] > 0:
                ov = w_all[j] * h_all[j] / float(area_a + area_all[j] - w_all[j] * h_all[j])
                if method == 1:
                    wgt = 1 - ov if ov > Nt else 1
                elif method == 2:
                    wgt = math.exp(-(ov * ov) / sigma)
                else:
                    wgt = 0 if ov > Nt else 1
                score[j] = wgt * score[j]
                if score[j] < threshold:       # :98-110
                    where[[p, live - 1]] = where[[live - 1, p]]
                    live -= 1
                    continue
            p += 1
    return where[:live].astype(np.int64)


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
