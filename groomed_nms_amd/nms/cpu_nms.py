"""cpu_nms(dets, thresh): stand-in for the reference's Cython routine (lib/nms/cpu_nms.pyx:17-68) -- host code in the
reference as well (the CPU sibling of gpu_nms, not a fallback of it): float32 arithmetic, +1-pixel areas, a box is
dropped when its IoU with a kept box is >= thresh (:65)."""
import numpy as np

from ._host import greedy_nms


def cpu_nms(dets, thresh):
    return greedy_nms(np.asarray(dets, dtype=np.float32), thresh, shift=1, rule="ge_drop", dtype=np.float32)
