"""Pairwise overlaps on MI355X -- mirror of the overlap helpers of the reference's lib/core.py and
lib/math_3d.py that feed the NMS.  Same names and argument meaning:
  iou(box_a, box_b, mode, data_type)                 lib/core.py:480-532
  intersect(box_a, box_b, mode, data_type)           lib/core.py:178-243
  iou3d_approximate(c1, c2, mode, method)            lib/core.py:305-421
  get_corners_of_cuboid(x, y, z, w, h, l, ry)        lib/math_3d.py:364-490
'combinations' mode runs in the HIP kernels (csrc/iou_kernels.hip); 'list' mode is O(N) elementwise
tensor arithmetic.  ndarray in -> ndarray out.  float64 ndarrays -- what the inference call site passes
(lib/rpn_util.py:1295: `aboxes` is float64 after the hstack at :1258) -- keep their dtype like the reference's
NumPy branches: `iou` (combinations) and `get_corners_of_cuboid` run the same IEEE double operations on the
GPU (gnms_iou2d_f64, gnms_corners_of_cuboid_f64), so the matrix that lib/groomed_nms.py:36 then rounds to
fp32 is the reference's bit for bit; other ndarrays are computed in fp32.
"""
import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr, on_device
from .groomed_nms import _device

__all__ = ["iou", "intersect", "iou3d_approximate", "get_corners_of_cuboid", "iou_batched", "iou3d_batched"]


def _is_f64_array(x):
    return isinstance(x, np.ndarray) and x.dtype == np.float64


def _to_dev(x):
    if isinstance(x, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(x)).to(device=_device(), dtype=torch.float32), "numpy", None
    dev = x.device if x.device.type == "cuda" else _device()
    return x.detach().to(device=dev, dtype=torch.float32), "torch", x.device


def _back(t, kind, device):
    if kind == "numpy":
        return t.cpu().numpy()
    return t.to(device)


def iou_batched(boxes_a, boxes_b=None, out=None):
    """boxes_a [B,M,4], boxes_b [B,N,4] (CUDA fp32) -> [B,M,N].  boxes_b=None means boxes_a."""
    from .groomed_nms import _binding
    ext = _binding()
    if ext and boxes_a.is_cuda and boxes_a.dtype == torch.float32 and (boxes_b is None or boxes_b.dtype == torch.float32):
        try:
            return ext.iou2d(boxes_a, boxes_a if boxes_b is None else boxes_b, out)
        except RuntimeError as e:                           # (the library's own failures only; torch's -- OOM first of all -- pass through)
            if isinstance(e, torch.cuda.OutOfMemoryError) or not str(e).startswith("GNMS:"):
                raise
            raise _lib.GnmsError(str(e)) from None
    lib = _lib.load()
    boxes_a = boxes_a.contiguous()
    boxes_b = boxes_a if boxes_b is None else boxes_b.contiguous()
    B, M, _ = boxes_a.shape
    N = boxes_b.shape[1]
    if out is None:
        out = torch.empty((B, M, N), dtype=torch.float32, device=boxes_a.device)
    with torch.cuda.device(boxes_a.device):
        check(lib.gnms_iou2d(ptr(boxes_a), ptr(boxes_b), B, M, N, ptr(out), N, stream_ptr(boxes_a.device)), "gnms_iou2d")
    return out


def iou3d_batched(a, b=None, method="generalized", from_params=False, want_bev=False, nms_overlap=False, out=None, nms_threshold=None):
    """a [B,M,3,8] corners (or [B,M,7] params when from_params) -> iou_3d [B,M,N] (and iou_bev).
    nms_overlap=True returns 0.5*(1+giou), the matrix both reference callers hand to the NMS.  With nms_threshold (the threshold
    the layer will apply; square from-params problems) the HBM-bound kernel of gnms_nms_overlap3d_from_params writes it: within
    2e-6 of the exact operation order everywhere, and exactly that order wherever the threshold decision could depend on it."""
    lib = _lib.load()
    a = a.contiguous()
    if nms_overlap and from_params and b is None and nms_threshold is not None and not want_bev:
        B, N = a.shape[0], a.shape[1]
        o3 = out if out is not None else torch.empty((B, N, N), dtype=torch.float32, device=a.device)
        with torch.cuda.device(a.device):
            check(lib.gnms_nms_overlap3d_from_params(ptr(a), B, N, float(nms_threshold), ptr(o3), max(N, 1), stream_ptr(a.device)),
                  "gnms_nms_overlap3d_from_params")
        return o3
    b = a if b is None else b.contiguous()
    B, M = a.shape[0], a.shape[1]
    N = b.shape[1]
    m = 2 if nms_overlap else {"normal": 0, "generalized": 1}[method]
    o3 = out if out is not None else torch.empty((B, M, N), dtype=torch.float32, device=a.device)
    bev = torch.empty((B, M, N), dtype=torch.float32, device=a.device) if want_bev else None
    fn = lib.gnms_iou3d_from_params if from_params else lib.gnms_iou3d_approximate
    with torch.cuda.device(a.device):
        check(fn(ptr(a), ptr(b), B, M, N, m, ptr(bev), ptr(o3), N, stream_ptr(a.device)), "gnms_iou3d")
    return (bev, o3) if want_bev else o3


def intersect(box_a, box_b, mode='combinations', data_type=None):
    """lib/core.py:178-243.  combinations -> N x M (note: [b][a], as in the reference); list -> M."""
    a, kind, dev = _to_dev(box_a)
    b, _, _ = _to_dev(box_b)
    if mode == 'combinations':
        max_xy = torch.min(a[:, 2:4], b[:, 2:4].unsqueeze(1))
        min_xy = torch.max(a[:, 0:2], b[:, 0:2].unsqueeze(1))
    elif mode == 'list':
        max_xy = torch.min(a[:, 2:4], b[:, 2:4])
        min_xy = torch.max(a[:, 0:2], b[:, 0:2])
    else:
        raise ValueError('unknown mode {}'.format(mode))
    inter = torch.clamp(max_xy - min_xy, 0)
    return _back(inter[..., 0] * inter[..., 1], kind, dev)


def iou(box_a, box_b, mode='combinations', data_type=None):
    """lib/core.py:480-532.  combinations: M x N matrix from the HIP kernel; list: M values."""
    if mode == 'combinations':
        # (both float64: with a float32 box_b the reference computes area_b in float32 before promoting -- that mix takes the fp32 kernel)
        if _is_f64_array(box_a) and _is_f64_array(box_b):
            # the reference's NumPy branch in float64 (lib/core.py:205-207, 512-513), the dtype its inference call site passes
            lib = _lib.load()
            dev = _device()
            a = torch.from_numpy(np.ascontiguousarray(box_a[:, :4], dtype=np.float64)).to(dev)
            b = torch.from_numpy(np.ascontiguousarray(box_b[:, :4], dtype=np.float64)).to(dev)
            M, N = a.shape[0], b.shape[0]
            out = torch.empty((M, N), dtype=torch.float64, device=dev)
            with on_device(dev):
                check(lib.gnms_iou2d_f64(ptr(a), ptr(b), 1, M, N, ptr(out), max(N, 1), stream_ptr(dev)), "gnms_iou2d_f64")
            return out.cpu().numpy()
        a, kind, dev = _to_dev(box_a)
        b, _, _ = _to_dev(box_b)
        out = iou_batched(a[:, :4].unsqueeze(0), b[:, :4].unsqueeze(0))[0]
        return _back(out, kind, dev)
    if mode == 'list':
        a, kind, dev = _to_dev(box_a)
        b, _, _ = _to_dev(box_b)
        max_xy = torch.min(a[:, 2:4], b[:, 2:4])
        min_xy = torch.max(a[:, 0:2], b[:, 0:2])
        wh = torch.clamp(max_xy - min_xy, 0)
        inter = wh[:, 0] * wh[:, 1]
        area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
        area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        return _back(inter / (area_a + area_b - inter), kind, dev)
    raise ValueError('unknown mode {}'.format(mode))


def get_corners_of_cuboid(x3d, y3d, z3d, w3d, h3d, l3d, ry3d, iou_3d_convention=True):
    """lib/math_3d.py:364-490, iou_3d_convention=True: the one every caller passes (lib/loss/rpn_3d.py:746-750, lib/rpn_util.py:1303).
    N x 3 x 8.  The reference's own iou_3d_convention=False branch is not usable: its torch form assigns an [N] tensor to an [N, 4]
    slice (math_3d.py:423-425: a shape error for every N but 1 and 4), its NumPy form has no such branch at all (:451-476: the box-frame
    corners stay zero) -- there is no behaviour to be faithful to, so it raises here."""
    if not iou_3d_convention:
        raise NotImplementedError("iou_3d_convention=False: the reference's own branch is broken (lib/math_3d.py:423-425 raises a shape error for "
                                  "N not in {1, 4}; the NumPy form lacks the branch); every caller passes True")
    lib = _lib.load()
    kind = "numpy" if isinstance(x3d, np.ndarray) else "torch"
    if kind == "numpy":
        # lib/math_3d.py:438-490: the NumPy branch builds R and the box-frame corners with .astype(float), i.e. in float64 whatever
        # the dtype of the arguments -- the corners come back as a float64 array
        dev = _device()
        params = torch.from_numpy(np.stack([np.asarray(v, dtype=np.float64).reshape(-1) for v in (x3d, y3d, z3d, w3d, h3d, l3d, ry3d)], 1)).to(dev)
        n = params.shape[0]
        corners = torch.empty((n, 3, 8), dtype=torch.float64, device=dev)
        with on_device(dev):
            # np.cos / np.sin evaluate in the dtype of ry3d (float32 at lib/rpn_util.py:1303-1309)
            trig_f32 = int(np.asarray(ry3d).dtype == np.float32)
            check(lib.gnms_corners_of_cuboid_f64(ptr(params), n, trig_f32, ptr(corners), stream_ptr(dev)), "gnms_corners_of_cuboid_f64")
        return corners.cpu().numpy()
    cols = [torch.as_tensor(v) for v in (x3d, y3d, z3d, w3d, h3d, l3d, ry3d)]
    out_device = cols[0].device
    dev = out_device if out_device.type == "cuda" else _device()
    params = torch.stack([c.to(device=dev, dtype=torch.float32).reshape(-1) for c in cols], dim=1).contiguous()
    n = params.shape[0]
    corners = torch.empty((n, 3, 8), dtype=torch.float32, device=dev)
    with on_device(dev):
        check(lib.gnms_corners_of_cuboid(ptr(params), n, ptr(corners), stream_ptr(dev)), "gnms_corners_of_cuboid")
    return _back(corners, kind, out_device)


def corners_batched(params):
    """params [B,N,7] (x, y, z, w, h, l, ry; CUDA fp32) -> corners [B,N,3,8]: get_corners_of_cuboid for a whole batch in one launch."""
    lib = _lib.load()
    params = params.contiguous()
    B, N = params.shape[0], params.shape[1]
    corners = torch.empty((B, N, 3, 8), dtype=torch.float32, device=params.device)
    with on_device(params.device):
        check(lib.gnms_corners_of_cuboid(ptr(params), B * N, ptr(corners), stream_ptr(params.device)), "gnms_corners_of_cuboid")
    return corners


def iou3d_approximate(corners_3d_b1, corners_3d_b2, mode="list", method="normal"):
    """lib/core.py:305-421.  Returns (iou_bev, iou_3d).  Inputs are NOT modified (the reference overwrites
    the y row of its inputs through a view, :379-380)."""
    c1, kind, dev = _to_dev(corners_3d_b1)
    c2, _, _ = _to_dev(corners_3d_b2)
    if c1.dim() == 2:
        c1, c2 = c1.unsqueeze(0), c2.unsqueeze(0)
    if method not in ("normal", "generalized"):
        raise ValueError("unknown method {}".format(method))
    if mode == "combinations":
        bev, i3 = iou3d_batched(c1.unsqueeze(0), c2.unsqueeze(0), method=method, want_bev=True)
        return _back(bev[0], kind, dev), _back(i3[0], kind, dev)
    if mode == "list":
        # O(N): the diagonal of the pairwise problem, done box by box on the same kernel
        n = c1.shape[0]
        bev, i3 = iou3d_batched(c1.reshape(n, 1, 3, 8), c2.reshape(n, 1, 3, 8), method=method, want_bev=True)
        return _back(bev.reshape(n), kind, dev), _back(i3.reshape(n), kind, dev)
    raise ValueError('unknown mode {}'.format(mode))
