"""cpu_nms(dets, thresh) -- the reference's CPU routine lib/nms/cpu_nms.pyx:17-68 (Cython, host code by
definition: it is the CPU sibling of gpu_nms, not a fallback of it).  Suppression rule '>=' (:65),
+1-pixel areas (:24), fp32 arithmetic."""
import numpy as np


def cpu_nms(dets, thresh):
    dets = np.asarray(dets, dtype=np.float32)
    x1, y1, x2, y2, scores = dets[:, 0], dets[:, 1], dets[:, 2], dets[:, 3], dets[:, 4]
    areas = (x2 - x1 + np.float32(1)) * (y2 - y1 + np.float32(1))
    order = scores.argsort()[::-1]
    ndets = dets.shape[0]
    suppressed = np.zeros(ndets, dtype=bool)
    thresh = np.float32(thresh)
    keep = []
    for _i in range(ndets):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(int(i))
        rest = order[_i + 1:]
        rest = rest[~suppressed[rest]]
        w = np.maximum(np.float32(0), np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]) + np.float32(1))
        h = np.maximum(np.float32(0), np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]) + np.float32(1))
        inter = w * h
        ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr >= thresh]] = True
    return keep
