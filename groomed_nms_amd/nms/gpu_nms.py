"""gpu_nms(dets, thresh, device_id=0) -- stand-in for the reference's Cython wrapper lib/nms/gpu_nms.pyx:16-31.
Calls the C symbol `_nms` of libgroomed_nms_hip.so, whose signature is the reference's
(lib/nms/gpu_nms.hpp:1-2): host pointers, boxes pre-sorted by score, blocking."""
import ctypes

import numpy as np

from .. import _lib


def gpu_nms(dets, thresh, device_id=0):
    lib = _lib.load()
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    boxes_num, boxes_dim = dets.shape
    if boxes_num == 0:
        return []
    keep = np.zeros(boxes_num, dtype=np.int32)                       # gpu_nms.pyx:21-22
    num_out = ctypes.c_int(0)
    scores = dets[:, 4]
    # Both call sites hand over boxes that are sorted already (lib/rpn_util.py:1258-1266 sorts `aboxes` by score in front of the NMS).  For
    # STRICTLY descending scores argsort()[::-1] is the identity, so sort and gather -- half of a call at n = 4096 -- are skipped; ties
    # or NaNs (the order among equal scores is argsort's) take the reference's two lines as they are.
    if boxes_num > 1 and bool(np.all(scores[1:] < scores[:-1])):
        order = None
        sorted_dets = dets
    else:
        order = scores.argsort()[::-1]                               # :25-26
        sorted_dets = np.ascontiguousarray(dets[order, :])           # :27-28
    lib._nms(keep.ctypes.data_as(ctypes.c_void_p), ctypes.cast(ctypes.byref(num_out), ctypes.c_void_p),
             sorted_dets.ctypes.data_as(ctypes.c_void_p), boxes_num, boxes_dim, float(thresh), int(device_id))   # :29
    if num_out.value == 0 and boxes_num > 0:
        msg = lib.gnms_last_error()
        raise _lib.GnmsError("_nms failed: %s" % (msg.decode() if msg else "no box kept"))
    keep = keep[:num_out.value]
    return list(keep.astype(np.int64)) if order is None else list(order[keep])   # :30-31
