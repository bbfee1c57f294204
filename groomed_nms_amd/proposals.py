"""Decode and selection of the boxes GrooMeD-NMS sees, on the GPU.

Host-side mirror of the reference pieces that sit directly in front of the layer:
  bbox_transform_inv        lib/rpn_util.py:872-934 (same name, arguments and return shape)
  select_topk               lib/loss/rpn_3d.py:731-737 / lib/rpn_util.py:1258-1266 (sort by score, keep the first K) -- without
                            the .cpu()/.numpy() round trip of rpn_3d.py:740-744
  projected_boxes_2d        lib/loss/rpn_3d.py:746-768 (cuboid -> corners -> projection -> 2D box)
  best_targets              lib/loss/rpn_3d.py:801-825 (best box per ground truth after the NMS)
All arithmetic runs in HIP kernels behind the C ABI (include/groomed_nms_hip.h); no CPU implementation lives here.
"""
import ctypes

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr, on_device

__all__ = ["bbox_transform_inv", "select_topk", "projected_boxes_2d", "best_targets", "training_tail"]


def _device():
    if not torch.cuda.is_available():
        raise _lib.GnmsError("needs an AMD GPU (torch.cuda.is_available() is False); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _binding():
    from .groomed_nms import _binding as b
    return b()


def _gnms(fn, *args):
    """a call into the C++ binding: the library's own failures (raised there as "GNMS: ...") become GnmsError, torch's stay what they are"""
    try:
        return fn(*args)
    except RuntimeError as e:
        if isinstance(e, torch.cuda.OutOfMemoryError) or not str(e).startswith("GNMS:"):
            raise
        raise _lib.GnmsError(str(e)) from None


def training_tail(scores, boxes2d, params3d, gt_params, gt_boxes, beta, counts=None, gt_counts=None, nms_threshold=0.4, temperature=0.01,
                  valid_box_prob_threshold=0.3, pruning_method="linear", group_size=100):
    """lib/loss/rpn_3d.py:772-825 + :1117-1131 in ONE host call (C++ binding; stream-ordered launches only, HIP-graph capturable): scores
    [B,N] sorted descending per image (:731-737), their boxes2d [B,N,4] / params3d [B,N,7], the ground truth [B,M,7] / [B,M,4] ->
    GrooMeD-NMS on the 2D overlaps -> best box per ground truth (beta) -> after-NMS AP loss.  Returns (loss [B], prob [B,N], targets [B,N])."""
    ext = _binding()
    if not ext:
        raise _lib.GnmsError("training_tail needs the C++ binding (groomed_nms_amd/gnms_torch*.so)")
    prune = {"linear": 0, "sigmoidal": 1, "soft_nms": 2}[pruning_method]
    c = counts.to(device=scores.device, dtype=torch.int32).contiguous() if counts is not None else None
    g = gt_counts.to(device=scores.device, dtype=torch.int32).contiguous() if gt_counts is not None else None
    return tuple(_gnms(ext.training_tail, scores.float(), boxes2d.float(), params3d, gt_params, gt_boxes, float(beta), c, g, float(nms_threshold),
                       float(temperature), float(valid_box_prob_threshold), prune, int(min(int(group_size), 2 ** 31 - 2))))


def _f4(v):
    if v is None:
        return None
    vals = [float(x) for x in (v.tolist() if hasattr(v, "tolist") else v)][:4]
    return (ctypes.c_float * 4)(*vals)


def bbox_transform_inv(boxes, deltas, means=None, stds=None):
    """lib/rpn_util.py:872-934.  boxes [A,>=4] anchors, deltas [A,4] or [B,A,4] -> predicted boxes, same shape as deltas.
    Unlike the reference (:903-913) `deltas` is not modified."""
    if boxes.shape[0] == 0:
        return torch.zeros((0, deltas.shape[1]), dtype=deltas.dtype)                  # :881-882
    lib = _lib.load()
    was_cuda = deltas.is_cuda
    dev = deltas.device if was_cuda else _device()
    three = deltas.dim() == 3
    d = deltas.detach().to(device=dev, dtype=torch.float32)
    d = (d if three else d.unsqueeze(0))[..., :4].contiguous()
    a = boxes.detach().to(device=dev, dtype=torch.float32)[:, :4].contiguous()
    B, A = d.shape[0], d.shape[1]
    out = torch.empty((B, A, 4), dtype=torch.float32, device=dev)
    m, s = _f4(means), _f4(stds)
    with on_device(dev):
        check(lib.gnms_bbox_transform_inv(ptr(a), ptr(d), B, A, m, s, ptr(out), stream_ptr()), "gnms_bbox_transform_inv")
    out = out if three else out[0]
    return out if was_cuda else out.cpu()


def select_topk(scores, k, candidates=None, candidate_counts=None, boxes=None):
    """scores [B,A] (GPU) -> (index [B,k] int64 padded with -1, count [B] int32, scores [B,k], boxes [B,k,4] or None): per image
    the k best-scoring candidates by descending score (ties: candidate order).  `candidates` [B,F] int32 indices with
    `candidate_counts` [B] restrict the choice (the foreground boxes of lib/loss/rpn_3d.py:731); None = every box."""
    if not scores.is_cuda:
        raise _lib.GnmsError("select_topk expects GPU tensors")
    ext = _binding()
    if ext:                                              # the C++ host path (csrc/torch_binding.cpp); below: ctypes, the same C ABI
        dev = scores.device
        cand = candidates.to(dev) if candidates is not None else None
        cnt = candidate_counts.to(dev) if (candidates is not None and candidate_counts is not None) else None
        idx, num, ssel, bsel = _gnms(ext.select_topk, scores, int(k), cand, cnt, boxes.to(dev) if boxes is not None else None)
        return idx, num, ssel, bsel
    lib = _lib.load()
    dev = scores.device
    s = scores.detach().to(torch.float32).contiguous()
    B, A = s.shape
    cand = cnt = None
    F = A
    if candidates is not None:
        cand = candidates.to(device=dev, dtype=torch.int32).contiguous()
        F = cand.shape[1]
        if candidate_counts is not None:
            cnt = candidate_counts.to(device=dev, dtype=torch.int32).contiguous()
    bx = boxes.detach().to(device=dev, dtype=torch.float32)[..., :4].contiguous() if boxes is not None else None
    idx = torch.empty((B, k), dtype=torch.int64, device=dev)
    num = torch.empty((B,), dtype=torch.int32, device=dev)
    ssel = torch.empty((B, k), dtype=torch.float32, device=dev)
    bsel = torch.empty((B, k, 4), dtype=torch.float32, device=dev) if bx is not None else None
    with on_device(dev):
        check(lib.gnms_select_topk(ptr(s), B, A, ptr(cand), F, ptr(cnt), int(k), ptr(bx), ptr(idx), ptr(num), ptr(ssel), ptr(bsel),
                                   stream_ptr()), "gnms_select_topk")
    return idx, num, ssel, bsel


def projected_boxes_2d(params, p2, scale_factor=None):
    """params [B,N,7] = (x3d, y3d, z3d, w3d, h3d, l3d, ry3d), p2 [B,4,4] (or [4,4]), scale_factor [B] / float / None
    -> [B,N,4] boxes (x1, y1, x2, y2) of the projected cuboids (lib/loss/rpn_3d.py:746-768)."""
    if not params.is_cuda:
        raise _lib.GnmsError("projected_boxes_2d expects GPU tensors")
    lib = _lib.load()
    dev = params.device
    p = params.detach().to(torch.float32).contiguous()
    B, N = p.shape[0], p.shape[1]
    P = torch.as_tensor(p2, dtype=torch.float32, device=dev)
    P = (P.unsqueeze(0).expand(B, 4, 4) if P.dim() == 2 else P).contiguous()
    sc = None
    if scale_factor is not None:
        sc = torch.as_tensor(scale_factor, dtype=torch.float32, device=dev).reshape(-1)
        sc = (sc.expand(B) if sc.numel() == 1 else sc).contiguous()
    out = torch.empty((B, N, 4), dtype=torch.float32, device=dev)
    with on_device(dev):
        check(lib.gnms_project_boxes3d(ptr(p), ptr(P), ptr(sc), B, N, ptr(out), stream_ptr()), "gnms_project_boxes3d")
    return out


def best_targets(pred_params, pred_boxes, gt_params, gt_boxes, beta, pred_counts=None, gt_counts=None):
    """lib/loss/rpn_3d.py:801-825 for a whole batch: pred_params [B,N,7], pred_boxes [B,N,4], gt_params [B,M,7], gt_boxes [B,M,4]
    (GPU) -> (targets [B,N] fp32 in {0,1}, best_index [B,M] int64 with -1 where no box scores above beta, best_score [B,M])."""
    if not pred_params.is_cuda:
        raise _lib.GnmsError("best_targets expects GPU tensors")
    ext = _binding()
    if ext:
        pc = pred_counts.to(device=pred_params.device, dtype=torch.int32).contiguous() if pred_counts is not None else None
        gc = gt_counts.to(device=pred_params.device, dtype=torch.int32).contiguous() if gt_counts is not None else None
        return tuple(_gnms(ext.best_targets, pred_params, pred_boxes.to(pred_params.device), gt_params.to(pred_params.device),
                           gt_boxes.to(pred_params.device), float(beta), pc, gc))
    lib = _lib.load()
    dev = pred_params.device
    pp = pred_params.detach().to(torch.float32).contiguous()
    pb = pred_boxes.detach().to(device=dev, dtype=torch.float32)[..., :4].contiguous()
    gp = gt_params.detach().to(device=dev, dtype=torch.float32).contiguous()
    gb = gt_boxes.detach().to(device=dev, dtype=torch.float32)[..., :4].contiguous()
    B, N, M = pp.shape[0], pp.shape[1], gp.shape[1]
    pc = pred_counts.to(device=dev, dtype=torch.int32).contiguous() if pred_counts is not None else None
    gc = gt_counts.to(device=dev, dtype=torch.int32).contiguous() if gt_counts is not None else None
    targets = torch.empty((B, N), dtype=torch.float32, device=dev)
    idx = torch.empty((B, M), dtype=torch.int64, device=dev)
    score = torch.empty((B, M), dtype=torch.float32, device=dev)
    with on_device(dev):
        check(lib.gnms_best_targets(ptr(pp), ptr(pb), ptr(gp), ptr(gb), B, N, M, ptr(pc), ptr(gc), float(beta), ptr(idx), ptr(score),
                                    ptr(targets), stream_ptr()), "gnms_best_targets")
    return targets, idx, score
