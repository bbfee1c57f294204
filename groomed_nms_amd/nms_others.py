"""Comparison baselines of the reference's lib/nms_others.py (test-only callers there, test/test_differentiable_nms_forward.py:111-114)
on the GPU: Soft-NMS with index tracking (gnms_soft_nms, csrc/nms_others.hip) and Girshick NMS with `shift` (gnms_nms_sorted_shift,
csrc/classic_nms.hip).  ndarray in -> ndarray / list out, like the reference; there is no CPU implementation here."""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr
from .groomed_nms import _device

__all__ = ["navneeth_soft_nms", "girshick_nms"]


def navneeth_soft_nms(boxes, sigma=0.5, Nt=0.4, threshold=0.001, method=0, shift=1):
    """lib/nms_others.py:6-116: Soft-NMS (Bodla et al.) with the reference's slot bookkeeping; returns the kept original indices in
    the reference's slot order (`keep_orig[:N]`).  method 0 hard, 1 linear, 2 gaussian.  float64 arrays are processed in fp64, anything
    else in fp32 with fp64 overlaps (the reference's NumPy scalar arithmetic).  Unlike the reference the caller's array is left
    untouched (it decays the scores and swaps the rows of `boxes` in place, :37-60,:93-104)."""
    lib = _lib.load()
    dev = _device()
    arr = np.asarray(boxes)
    fp64 = arr.dtype == np.float64
    arr = np.ascontiguousarray(arr, dtype=np.float64 if fp64 else np.float32)
    n, dim = arr.shape
    if n == 0:
        return np.zeros((0,), np.int64)
    d = torch.from_numpy(arr).to(dev)
    keep = torch.empty((n,), dtype=torch.int64, device=dev)
    num = torch.zeros((1,), dtype=torch.int32, device=dev)
    ws = torch.empty((lib.gnms_soft_nms_workspace_bytes(n),), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(lib.gnms_soft_nms(ptr(d), n, dim, int(fp64), float(sigma), float(Nt), float(threshold), int(method), float(shift), ptr(keep),
                                ptr(num), ptr(ws), ws.numel(), _lib.stream_ptr(dev)), "gnms_soft_nms")
    return keep[:int(num.item())].cpu().numpy()


def girshick_nms(dets, thresh, shift=1):
    """lib/nms_others.py:119-150: greedy NMS with a configurable pixel `shift`, boxes with IoU <= thresh survive.  The reference returns
    `keep_orig`, built as kept index + N_dropped where N_dropped is recomputed after every round as len(order) - len(inds) on the ALREADY
    filtered order (:148) -- which is always 0 -- so the list equals the kept indices; reproduced as such.  The arithmetic runs in the
    dtype of `dets` like the reference's NumPy expressions: float64 arrays (what its own test feeds) in fp64 on the device
    (gnms_nms_sorted_shift_f64), anything else in fp32 (the dtype `gpu_nms` takes as well)."""
    lib = _lib.load()
    dev = _device()
    arr = np.asarray(dets)
    fp64 = arr.dtype == np.float64
    arr = np.ascontiguousarray(arr, dtype=np.float64 if fp64 else np.float32)
    n, dim = arr.shape
    if n == 0:
        return []
    order = arr[:, 4].argsort()[::-1]                                # :131
    d = torch.from_numpy(np.ascontiguousarray(arr[order])).to(dev)
    keep = torch.empty((n,), dtype=torch.int32, device=dev)
    num = torch.zeros((1,), dtype=torch.int32, device=dev)
    ws = torch.empty((lib.gnms_nms_workspace_bytes(n),), dtype=torch.uint8, device=dev)
    entry, what = (lib.gnms_nms_sorted_shift_f64, "gnms_nms_sorted_shift_f64") if fp64 else (lib.gnms_nms_sorted_shift, "gnms_nms_sorted_shift")
    with torch.cuda.device(dev):
        check(entry(ptr(d), n, dim, float(thresh), float(shift), 1, ptr(keep), ptr(num), ptr(ws), ws.numel(), _lib.stream_ptr(dev)), what)
    k = keep[:int(num.item())].cpu().numpy()
    return [np.int64(i) for i in order[k]]
