"""py_cpu_nms(dets, thresh): stand-in for the reference's NumPy baseline (lib/nms/py_cpu_nms.py:10-38): +1-pixel areas,
boxes whose IoU with a kept box is <= thresh survive."""
from ._host import greedy_nms


def py_cpu_nms(dets, thresh):
    return greedy_nms(dets, thresh, shift=1, rule="le_keep")
