"""Host-side greedy NMS used by the three CPU entry points of the reference's lib/nms package (cpu_nms, py_cpu_nms) and
by lib/nms_others.girshick_nms.  They differ only in the pixel convention (`shift`), the arithmetic width and the
comparison at exactly IoU == thresh, so one routine serves all of them here:

    rule "le_keep"   keep the boxes with IoU <= thresh   (py_cpu_nms.py:35, nms_others.py:146)   -> NaN overlaps are dropped
    rule "ge_drop"   drop the boxes with IoU >= thresh   (cpu_nms.pyx:65)                       -> NaN overlaps are kept
"""
import numpy as np


def overlap_with(anchor, others, shift, dtype):
    """IoU of one box against many with the (x2 - x1 + shift) pixel convention (nms_kernel.cu:24-32 has shift = 1)."""
    sh = dtype(shift)
    zero = dtype(0)
    iw = np.maximum(zero, np.minimum(anchor[2], others[:, 2]) - np.maximum(anchor[0], others[:, 0]) + sh)
    ih = np.maximum(zero, np.minimum(anchor[3], others[:, 3]) - np.maximum(anchor[1], others[:, 1]) + sh)
    inter = iw * ih
    area_anchor = (anchor[2] - anchor[0] + sh) * (anchor[3] - anchor[1] + sh)
    area_others = (others[:, 2] - others[:, 0] + sh) * (others[:, 3] - others[:, 1] + sh)
    return inter / (area_anchor + area_others - inter)


def greedy_nms(dets, thresh, shift=1, rule="le_keep", dtype=None):
    """Returns the kept ORIGINAL indices in descending-score order."""
    dets = np.asarray(dets)
    if dtype is None:
        dtype = dets.dtype.type
    boxes = dets[:, :4].astype(dtype, copy=False)
    remaining = dets[:, 4].argsort()[::-1]
    thresh = dtype(thresh)
    kept = []
    while remaining.size:
        top, rest = remaining[0], remaining[1:]
        kept.append(int(top))
        ov = overlap_with(boxes[top], boxes[rest], shift, dtype)
        survive = (ov <= thresh) if rule == "le_keep" else ~(ov >= thresh)
        remaining = rest[survive]
    return kept
