"""GrooMeD-NMS layer on MI355X -- host-side mirror of the reference module lib/groomed_nms.py.

Same public names, argument meaning, defaults, return convention and error behaviour as the
reference (file:line cited per function), so `from groomed_nms_amd.groomed_nms import *` can stand in
for `from lib.groomed_nms import *` (lib/rpn_util.py:18, lib/loss/rpn_3d.py:14).  All arithmetic runs
in hand-written HIP kernels behind the C ABI of libgroomed_nms_hip.so (include/groomed_nms_hip.h);
there is no CPU implementation here -- without the library or without a GPU the calls raise.
"""
import ctypes
import itertools

import numpy as np
import torch

from . import _lib
from ._lib import GnmsParams, check, ptr, stream_ptr, on_device

__all__ = ["differentiable_nms", "differentiable_nms_batched", "differentiable_nms_from_boxes_batched",
           "differentiable_nms_with_iou2d_batched", "differentiable_nms_with_iou3d_batched", "soft_sort", "pruning_function", "sigmoid_numpy",
           "cast_to_cpu_cuda_tensor", "get_groups", "indices_copy", "GroomedNMS", "LazyIndexList"]

_PRUNE = {"linear": 0, "sigmoidal": 1, "soft_nms": 2}


def _device():
    if not torch.cuda.is_available():
        raise _lib.GnmsError("GrooMeD-NMS needs an AMD GPU (torch.cuda.is_available() is False); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


_PARAMS = {}


def _params(nms_threshold, pruning_method, temperature, valid_box_prob_threshold, return_sorted_prob, group_boxes,
            mask_group_boxes, group_size, presorted=False, index_lists=True):
    """struct gnms_params for these keyword arguments; one object per distinct VALUE tuple (treated as immutable by every user:
    building the ctypes structure anew cost ~3 us per call).  The arguments are converted to plain Python numbers first and the
    cache is keyed on those: a 0-dim tensor threshold or temperature is re-read on every call, as the reference does
    (lib/groomed_nms.py:10 takes them as plain arguments), and is never pinned by the cache."""
    if pruning_method not in _PRUNE:
        raise NotImplementedError("Pruning method not implemented!")              # lib/groomed_nms.py:177-178
    key = (float(nms_threshold), _PRUNE[pruning_method], float(temperature), float(valid_box_prob_threshold), bool(return_sorted_prob),
           bool(group_boxes), bool(mask_group_boxes), int(min(int(group_size), 2 ** 31 - 2)), bool(presorted), bool(index_lists))
    p = _PARAMS.get(key)
    if p is None:
        p = GnmsParams(key[0], key[2], key[3], key[1], int(key[4]), int(key[5]), int(key[6]), key[7], int(key[8]))
        p.index_lists = key[9]
        if len(_PARAMS) < 256:
            _PARAMS[key] = p
    return p


# ----------------------------------------------------------------------------------------------------------------------------------
# The layer's host path.  Preferred: the C++ binding (csrc/torch_binding.cpp, built in-tree as gnms_torch*.so) -- outputs, workspace,
# the gnms_* call on torch's current stream and the autograd node all in C++.  Fallback (binding not built, or GNMS_BINDING=ctypes):
# the ctypes + torch.autograd.Function path below, which calls the same C ABI.  Both are the HIP library; neither computes anything
# on the host.
# ----------------------------------------------------------------------------------------------------------------------------------
_EXT = None
_MODE_MATRIX, _MODE_IOU2D, _MODE_IOU3D, _MODE_BOXES = 0, 1, 2, 3


def _binding():
    """the C++ autograd binding module, or False when it is unavailable / switched off"""
    global _EXT
    if _EXT is None:
        import os
        _EXT = False
        if os.environ.get("GNMS_BINDING", "") != "ctypes":
            _lib.load()                                   # (fails loudly when the HIP library itself is missing)
            try:
                from . import gnms_torch                  # noqa: F401  links libgroomed_nms_hip.so ($ORIGIN)
                if gnms_torch.abi_version() == 1:
                    _EXT = gnms_torch
            except ImportError:
                _EXT = False
    return _EXT


def _layer(ext, scores, src, counts, iou_out, mode, p):
    try:
        return ext.layer(scores, src, counts, iou_out, mode, p.nms_threshold, p.temperature, p.valid_box_prob_threshold, p.pruning_method,
                         bool(p.return_sorted_prob), bool(p.group_boxes), bool(p.mask_group_boxes), p.group_size, bool(p.presorted),
                         bool(getattr(p, "index_lists", True)))
    except NotImplementedError:
        raise
    except RuntimeError as e:
        # only the library's own failures (the C ABI's status codes + gnms_last_error(), raised by the binding as "GNMS: ...") become
        # GnmsError; torch's errors -- torch.cuda.OutOfMemoryError first of all -- reach the caller as they are, traceback included
        if isinstance(e, torch.cuda.OutOfMemoryError) or not str(e).startswith("GNMS:"):
            raise
        raise _lib.GnmsError(str(e)) from None


_WS_BYTES = {}


def _workspace_bytes(lib, B, N, params):
    """gnms_workspace_bytes, remembered per (B, N, group_boxes) -- the only fields it depends on (include/groomed_nms_hip.h)."""
    key = (B, N, params.group_boxes)
    nbytes = _WS_BYTES.get(key)
    if nbytes is None:
        nbytes = _WS_BYTES[key] = lib.gnms_workspace_bytes(B, N, ctypes.byref(params))
    return nbytes


def _outputs(B, N, dev, index_lists=True):
    """prob [B,N] fp32; order / valid / invalid [B,N] int64 and nvalid / ninvalid [B] int32 as views of ONE allocation each
    (finalize_kernel writes every entry, -1 padding included): three allocator calls instead of six on the launch-bound path.
    index_lists=False: valid / invalid are None (NULL in the C ABI), which lets the layer skip their compaction and sort."""
    prob = torch.empty((B, N), dtype=torch.float32, device=dev)
    lists = torch.empty((3 if index_lists else 1, B, N), dtype=torch.int64, device=dev)
    counts = torch.empty((2, B), dtype=torch.int32, device=dev)
    return prob, lists[0], (lists[1] if index_lists else None), (lists[2] if index_lists else None), counts[0], counts[1]


def _matrix_layout(iou):
    """Returns (tensor, ld) with unit column stride and image stride N*ld, copying only if needed."""
    B, N, _ = iou.shape
    if N == 0:
        return iou.contiguous(), max(N, 1)
    if iou.stride(2) == 1 and iou.stride(1) >= N and (B == 1 or iou.stride(0) == N * iou.stride(1)):
        return iou, iou.stride(1)
    return iou.contiguous(), N


class _GroomedNMSFunction(torch.autograd.Function):
    """prob = GrooMeD-NMS(scores, iou); gradients w.r.t. scores and (only if it requires grad) iou.
    The reference differentiates the same graph with autograd (lib/groomed_nms.py:111)."""

    @staticmethod
    def forward(ctx, scores, iou, counts, params):
        lib = _lib.load()
        B, N = scores.shape
        dev = scores.device
        scores_c = scores.contiguous()
        iou_c, ld = _matrix_layout(iou)
        prob, order, valid, invalid, nvalid, ninvalid = _outputs(B, N, dev, getattr(params, "index_lists", True))
        nbytes = _workspace_bytes(lib, B, N, params)
        ws = torch.empty((max(nbytes, 256),), dtype=torch.uint8, device=dev)
        with on_device(dev):
            check(lib.gnms_forward(ptr(scores_c), ptr(iou_c), B, N, ld, ptr(counts), ctypes.byref(params), ptr(prob), ptr(order),
                                   ptr(valid), ptr(invalid), ptr(nvalid), ptr(ninvalid), ptr(ws), ws.numel(), stream_ptr(dev)),
                  "gnms_forward")
        ctx.params = params
        ctx.ld = ld
        ctx.set_materialize_grads(False)      # no zero-filled "gradients" for the five index outputs
        ctx.save_for_backward(scores_c, iou_c, counts, ws)
        ctx.mark_non_differentiable(*[x for x in (order, valid, invalid, nvalid, ninvalid) if x is not None])
        return prob, order, valid, invalid, nvalid, ninvalid

    @staticmethod
    def backward(ctx, grad_prob, *unused):
        lib = _lib.load()
        scores_c, iou_c, counts, ws = ctx.saved_tensors
        B, N = scores_c.shape
        dev = scores_c.device
        if grad_prob is None:
            return None, None, None, None
        grad_prob = grad_prob.contiguous().float()
        grad_scores = torch.empty_like(scores_c)
        grad_iou = None
        if ctx.needs_input_grad[1]:
            grad_iou = torch.empty((B, N, ctx.ld), dtype=torch.float32, device=dev)
        with on_device(dev):
            check(lib.gnms_backward(ptr(grad_prob), ptr(scores_c), ptr(iou_c), B, N, ctx.ld, ptr(counts), ctypes.byref(ctx.params),
                                    ptr(grad_scores), ptr(grad_iou), ptr(ws), ws.numel(), stream_ptr(dev)), "gnms_backward")
        if grad_iou is not None and ctx.ld != N:
            grad_iou = grad_iou[:, :, :N]
        return grad_scores, grad_iou, None, None


class _GroomedNMSWithIouFunction(torch.autograd.Function):
    """(prob, ..., iou) = 2D IoU matrix + GrooMeD-NMS in one library call (gnms_forward_with_iou2d): same results as
    overlaps.iou_batched followed by _GroomedNMSFunction; in the grouped modes the layer runs from the boxes beside the matrix write.  The matrix comes
    back detached, as the reference's loss feeds it (lib/loss/rpn_3d.py:791 `.clone().detach()`)."""

    @staticmethod
    def forward(ctx, scores, boxes, counts, params, iou_out):
        lib = _lib.load()
        B, N = scores.shape
        dev = scores.device
        scores_c = scores.contiguous()
        boxes_c = boxes.contiguous()
        three_d = boxes_c.shape[-1] == 7          # [B,N,7] cuboid parameters -> gnms_forward_with_iou3d, [B,N,4] boxes -> ..._iou2d
        entry, what = (lib.gnms_forward_with_iou3d, "gnms_forward_with_iou3d") if three_d else (lib.gnms_forward_with_iou2d,
                                                                                                "gnms_forward_with_iou2d")
        iou = iou_out if iou_out is not None else torch.empty((B, N, N), dtype=torch.float32, device=dev)
        prob, order, valid, invalid, nvalid, ninvalid = _outputs(B, N, dev, getattr(params, "index_lists", True))
        nbytes = _workspace_bytes(lib, B, N, params)
        ws = torch.empty((max(nbytes, 256),), dtype=torch.uint8, device=dev)
        with on_device(dev):
            check(entry(ptr(boxes_c), ptr(scores_c), B, N, max(N, 1), ptr(counts), ctypes.byref(params), ptr(iou), ptr(prob), ptr(order),
                        ptr(valid), ptr(invalid), ptr(nvalid), ptr(ninvalid), ptr(ws), ws.numel(), stream_ptr(dev)), what)
        ctx.params = params
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(*[x for x in (order, valid, invalid, nvalid, ninvalid, iou) if x is not None])
        # The masked-group backward (the default) never reads the overlaps, so the matrix is not kept at all.  The unmasked / ungrouped
        # backward does read it: there it is saved through autograd, whose version counter then catches a caller that overwrites the
        # (possibly caller-provided) `iou_out` buffer between forward and backward instead of silently producing wrong gradients.
        # Grouped unmasked 2D: the backward solves its groups from the boxes (gnms_backward_from_boxes: bit-identical overlaps, no matrix reads).
        # (the forward's own from-boxes gate, alignment included: an unaligned view took the matrix path and keeps the matrix -- ADVICE r4)
        ctx.bwd_boxes = bool((not three_d) and params.group_boxes and not params.mask_group_boxes and not params.presorted
                             and boxes_c.data_ptr() % 16 == 0)
        ctx.reads_iou = not (params.group_boxes and params.mask_group_boxes) and not ctx.bwd_boxes
        if ctx.bwd_boxes:
            ctx.save_for_backward(scores_c, counts, ws, boxes_c)
        elif ctx.reads_iou:
            ctx.save_for_backward(scores_c, counts, ws, iou)
        else:
            ctx.save_for_backward(scores_c, counts, ws)
        return prob, order, valid, invalid, nvalid, ninvalid, iou

    @staticmethod
    def backward(ctx, grad_prob, *unused):
        if grad_prob is None:
            return None, None, None, None, None
        lib = _lib.load()
        if ctx.bwd_boxes:
            scores_c, counts, ws, boxes_c = ctx.saved_tensors
            B, N = scores_c.shape
            dev = scores_c.device
            grad_prob = grad_prob.contiguous().float()
            grad_scores = torch.empty_like(scores_c)
            with on_device(dev):
                check(lib.gnms_backward_from_boxes(ptr(grad_prob), ptr(boxes_c), ptr(scores_c), B, N, ptr(counts), ctypes.byref(ctx.params),
                                                   ptr(grad_scores), ptr(ws), ws.numel(), stream_ptr(dev)), "gnms_backward_from_boxes")
            return grad_scores, None, None, None, None
        if ctx.reads_iou:
            scores_c, counts, ws, iou_c = ctx.saved_tensors
        else:
            scores_c, counts, ws = ctx.saved_tensors
            iou_c = scores_c                                   # any valid device pointer: the masked backward does not dereference it
        B, N = scores_c.shape
        dev = scores_c.device
        grad_prob = grad_prob.contiguous().float()
        grad_scores = torch.empty_like(scores_c)
        with on_device(dev):
            check(lib.gnms_backward(ptr(grad_prob), ptr(scores_c), ptr(iou_c), B, N, max(N, 1), ptr(counts), ctypes.byref(ctx.params),
                                    ptr(grad_scores), None, ptr(ws), ws.numel(), stream_ptr(dev)), "gnms_backward")
        return grad_scores, None, None, None, None


class _GroomedNMSFromBoxesFunction(torch.autograd.Function):
    """prob = GrooMeD-NMS(scores, boxes) without ever building the N x N overlap matrix (gnms_forward_from_boxes):
    bit-identical to overlaps.iou_batched + _GroomedNMSFunction, gradient w.r.t. scores (the reference's loss detaches the
    overlaps, lib/loss/rpn_3d.py:791)."""

    @staticmethod
    def forward(ctx, scores, boxes, counts, params):
        lib = _lib.load()
        B, N = scores.shape
        dev = scores.device
        scores_c = scores.contiguous()
        boxes_c = boxes.contiguous()
        prob, order, valid, invalid, nvalid, ninvalid = _outputs(B, N, dev, getattr(params, "index_lists", True))
        nbytes = _workspace_bytes(lib, B, N, params)
        ws = torch.empty((max(nbytes, 256),), dtype=torch.uint8, device=dev)
        with on_device(dev):
            check(lib.gnms_forward_from_boxes(ptr(boxes_c), ptr(scores_c), B, N, ptr(counts), ctypes.byref(params), ptr(prob), ptr(order),
                                              ptr(valid), ptr(invalid), ptr(nvalid), ptr(ninvalid), ptr(ws), ws.numel(), stream_ptr(dev)),
                  "gnms_forward_from_boxes")
        ctx.params = params
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(scores_c, boxes_c, counts, ws)
        ctx.mark_non_differentiable(*[x for x in (order, valid, invalid, nvalid, ninvalid) if x is not None])
        return prob, order, valid, invalid, nvalid, ninvalid

    @staticmethod
    def backward(ctx, grad_prob, *unused):
        if grad_prob is None:
            return None, None, None, None
        lib = _lib.load()
        scores_c, boxes_c, counts, ws = ctx.saved_tensors
        B, N = scores_c.shape
        dev = scores_c.device
        grad_prob = grad_prob.contiguous().float()
        grad_scores = torch.empty_like(scores_c)
        with on_device(dev):
            check(lib.gnms_backward_from_boxes(ptr(grad_prob), ptr(boxes_c), ptr(scores_c), B, N, ptr(counts), ctypes.byref(ctx.params),
                                               ptr(grad_scores), ptr(ws), ws.numel(), stream_ptr(dev)), "gnms_backward_from_boxes")
        return grad_scores, None, None, None


class _SoftSortFunction(torch.autograd.Function):
    """soft_sort (lib/groomed_nms.py:131-165) for one image: HIP forward (row kernels + fp32 MFMA GEMM) and HIP backward
    (gnms_soft_sort_backward: the hand-derived adjoint of the same expressions, including the reference's last-axis
    broadcast of the row sums, :155; its two GEMMs on the MFMA kernel as well)."""

    @staticmethod
    def forward(ctx, scores, matrix, temperature):
        lib = _lib.load()
        N = scores.shape[0]
        dev = scores.device
        scores_c = scores.contiguous()
        C = torch.empty((N, N), dtype=torch.float32, device=dev)
        soft_scores = torch.empty((N,), dtype=torch.float32, device=dev)
        soft_matrix = None
        m_c, ld = None, N
        square = True
        if matrix is not None:
            # the reference multiplies C [N,N] into any [N,K] matrix (lib/groomed_nms.py:163 torch.matmul); the fused entry takes the
            # square case (the layer's), other widths run the convex-combination kernel and then the GEMM with ldb = K
            if matrix.dim() != 2 or matrix.shape[0] != N:
                raise ValueError("soft_sort: full_matrix must be [N, K] with N = len(scores) = %d, got %s" % (N, tuple(matrix.shape)))
            m_c = matrix.contiguous()
            square = m_c.shape[1] == N
            if square:
                soft_matrix = torch.empty((N, N), dtype=torch.float32, device=dev)
        one = GnmsParams()
        lib.gnms_default_params(ctypes.byref(one))
        ws = torch.empty((max(lib.gnms_workspace_bytes(1, max(N, 1), ctypes.byref(one)), 256),), dtype=torch.uint8, device=dev)
        with on_device(dev):
            check(lib.gnms_soft_sort(ptr(scores_c), ptr(m_c) if square else None, N, ld, float(temperature), ptr(C), ptr(soft_scores),
                                     ptr(soft_matrix), ptr(ws), ws.numel(), stream_ptr(dev)), "gnms_soft_sort")
        if matrix is not None and not square:
            soft_matrix = _sgemm(C, m_c) if (N > 0 and m_c.shape[1] > 0) else torch.zeros_like(m_c)
        ctx.temperature = float(temperature)
        ctx.set_materialize_grads(False)
        ctx.has_matrix = matrix is not None
        ctx.save_for_backward(scores_c, m_c if m_c is not None else scores_c, C, ws)
        if matrix is None:
            return soft_scores, C
        return soft_scores, C, soft_matrix

    @staticmethod
    def backward(ctx, g_soft, g_C, g_mat=None):
        """gnms_soft_sort_backward: the hand-derived adjoint as HIP kernels (row / column passes + the two MFMA GEMMs)."""
        lib = _lib.load()
        scores, matrix, C, ws = ctx.saved_tensors
        n = scores.shape[0]
        dev = scores.device
        has_m = ctx.has_matrix and g_mat is not None
        K = matrix.shape[1] if has_m else 0
        d_s = torch.empty_like(scores)
        if n == 0:
            return d_s, (torch.zeros_like(matrix) if ctx.has_matrix else None), None
        g_soft_c = g_soft.contiguous().float() if g_soft is not None else None
        g_C_c = g_C.contiguous().float() if g_C is not None else None
        g_mat_c = g_mat.contiguous().float() if has_m else None
        d_m = torch.empty_like(matrix) if has_m else None
        nb = lib.gnms_soft_sort_backward_scratch_bytes(n, K)
        scratch = torch.empty((nb,), dtype=torch.uint8, device=dev)
        with on_device(dev):
            check(lib.gnms_soft_sort_backward(ptr(scores), ptr(matrix) if has_m else None, n, K, max(K, 1), ctx.temperature, ptr(C),
                                              ptr(g_soft_c), ptr(g_C_c), ptr(g_mat_c), ptr(d_s), ptr(d_m), ptr(ws), ws.numel(),
                                              ptr(scratch), scratch.numel(), stream_ptr(dev)), "gnms_soft_sort_backward")
        if ctx.has_matrix and d_m is None:
            d_m = torch.zeros_like(matrix)
        return d_s, (d_m if ctx.has_matrix else None), None


class _PruneFunction(torch.autograd.Function):
    """pruning_function (lib/groomed_nms.py:167-189) on the HIP kernel with its adjoint, so that a caller composing it in an
    autograd graph gets the gradient the reference's plain torch ops would give."""

    @staticmethod
    def forward(ctx, x, nms_threshold, temperature, method):
        lib = _lib.load()
        xc = x.contiguous()
        out = torch.empty_like(xc)
        with on_device(xc.device):
            check(lib.gnms_pruning_function(ptr(xc), xc.numel(), float(nms_threshold), float(temperature), method, ptr(out),
                                            stream_ptr(xc.device)), "gnms_pruning_function")
        ctx.save_for_backward(xc)
        ctx.args = (float(nms_threshold), float(temperature), method)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (xc,) = ctx.saved_tensors
        gc = g.contiguous().float()
        gi = torch.empty_like(xc)
        thr, temp, method = ctx.args
        with on_device(xc.device):
            check(lib.gnms_pruning_function_backward(ptr(xc), ptr(gc), xc.numel(), thr, temp, method, ptr(gi), stream_ptr(xc.device)),
                  "gnms_pruning_function_backward")
        return gi, None, None, None


def _sgemm(a, b):
    """fp32 GEMM on the matrix cores (gnms_sgemm, v_mfma_f32_32x32x2_f32)."""
    lib = _lib.load()
    M, K = a.shape
    K2, N = b.shape
    assert K == K2
    d = torch.empty((M, N), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        check(lib.gnms_sgemm(ptr(a), ptr(b), ptr(d), M, N, K, a.stride(0), b.stride(0), N, stream_ptr(a.device)), "gnms_sgemm")
    return d


def differentiable_nms_batched(scores, iou, counts=None, nms_threshold=0.4, pruning_method="linear", temperature=0.01,
                               valid_box_prob_threshold=0.3, return_sorted_prob=False, group_boxes=True,
                               mask_group_boxes=True, group_size=100, presorted=False, index_lists=True):
    """Batched hard-sort GrooMeD-NMS: scores [B,N], iou [B,N,N] (CUDA fp32), counts [B] int32 or None.
    Returns (prob [B,N], order [B,N], valid [B,N], invalid [B,N], nvalid [B], ninvalid [B]); the lists are
    padded -- the first nvalid[b] / ninvalid[b] entries are meaningful.  No host synchronisation.
    index_lists=False (all batched entries): valid / invalid come back as None and the layer skips their compaction and sort --
    the training call site reads only the probabilities (lib/loss/rpn_3d.py:791 uses `[2]`)."""
    params = _params(nms_threshold, pruning_method, temperature, valid_box_prob_threshold, return_sorted_prob, group_boxes,
                     mask_group_boxes, group_size, presorted, bool(index_lists))
    if counts is not None:
        counts = counts.to(device=scores.device, dtype=torch.int32).contiguous()
    ext = _binding()
    if ext and scores.is_cuda:
        return tuple(_layer(ext, scores.float(), iou.float(), counts, None, _MODE_MATRIX, params)[:6])
    return _GroomedNMSFunction.apply(scores.float(), iou.float(), counts, params)


def differentiable_nms_with_iou2d_batched(scores, boxes, counts=None, iou_out=None, nms_threshold=0.4, pruning_method="linear",
                                          temperature=0.01, valid_box_prob_threshold=0.3, return_sorted_prob=False, group_boxes=True,
                                          mask_group_boxes=True, group_size=100, index_lists=True):
    """scores [B,N], boxes [B,N,4] -> (prob, order, valid, invalid, nvalid, ninvalid, iou [B,N,N]): the 2D IoU matrix AND the
    layer on it in one call -- what lib/loss/rpn_3d.py:772-791 does in two steps; identical results, and the matrix is
    returned for the caller's later use.  `iou_out` lets the caller provide the matrix buffer."""
    params = _params(nms_threshold, pruning_method, temperature, valid_box_prob_threshold, return_sorted_prob, group_boxes,
                     mask_group_boxes, group_size, False, bool(index_lists))
    if counts is not None:
        counts = counts.to(device=scores.device, dtype=torch.int32).contiguous()
    ext = _binding()
    if ext and scores.is_cuda:
        return tuple(_layer(ext, scores.float(), boxes.float(), counts, iou_out, _MODE_IOU2D, params))
    return _GroomedNMSWithIouFunction.apply(scores.float(), boxes.float(), counts, params, iou_out)


def differentiable_nms_with_iou3d_batched(scores, params3d, counts=None, iou_out=None, nms_threshold=0.4, pruning_method="linear",
                                          temperature=0.01, valid_box_prob_threshold=0.3, return_sorted_prob=False, group_boxes=True,
                                          mask_group_boxes=True, group_size=100, index_lists=True):
    """scores [B,N], params3d [B,N,7] = (x3d, y3d, z3d, w3d, h3d, l3d, ry3d) -> (prob, order, valid, invalid, nvalid, ninvalid,
    overlap [B,N,N]): the 3D NMS overlap 0.5*(1+GIoU3D) of lib/loss/rpn_3d.py:778-784 AND the layer on it in one call; identical
    to overlaps.iou3d_batched(from_params=True, nms_overlap=True) + differentiable_nms_batched."""
    params = _params(nms_threshold, pruning_method, temperature, valid_box_prob_threshold, return_sorted_prob, group_boxes,
                     mask_group_boxes, group_size, False, bool(index_lists))
    if counts is not None:
        counts = counts.to(device=scores.device, dtype=torch.int32).contiguous()
    if params3d.shape[-1] != 7:
        raise ValueError("params3d must be [B, N, 7]")
    ext = _binding()
    if ext and scores.is_cuda:
        return tuple(_layer(ext, scores.float(), params3d.float(), counts, iou_out, _MODE_IOU3D, params))
    return _GroomedNMSWithIouFunction.apply(scores.float(), params3d.float(), counts, params, iou_out)


def differentiable_nms_from_boxes_batched(scores, boxes, counts=None, nms_threshold=0.4, pruning_method="linear", temperature=0.01,
                                          valid_box_prob_threshold=0.3, return_sorted_prob=False, mask_group_boxes=True,
                                          group_size=100, index_lists=True):
    """From-boxes path: scores [B,N], boxes [B,N,4] = (x1,y1,x2,y2) -> the same six outputs as differentiable_nms_batched,
    bit-identical to building the 2D IoU matrix (overlaps.iou_batched) and running the layer on it, but the N x N matrix is
    never written to or read from HBM.  Grouped modes only (the defaults of scripts/config/groumd_nms.py)."""
    params = _params(nms_threshold, pruning_method, temperature, valid_box_prob_threshold, return_sorted_prob, True,
                     mask_group_boxes, group_size, False, bool(index_lists))
    if counts is not None:
        counts = counts.to(device=scores.device, dtype=torch.int32).contiguous()
    ext = _binding()
    if ext and scores.is_cuda:
        return tuple(_layer(ext, scores.float(), boxes.float(), counts, None, _MODE_BOXES, params)[:6])
    return _GroomedNMSFromBoxesFunction.apply(scores.float(), boxes.float(), counts, params)


def differentiable_nms(scores_unsorted, iou_unsorted, nms_threshold=0.4, pruning_method="linear", temperature=0.01,
                       valid_box_prob_threshold=0.3, return_sorted_prob=False, sorting_method="hard", sorting_temperature=None,
                       group_boxes=True, mask_group_boxes=True, group_size=100, debug=False):
    """
        GrooMeD-NMS: Grouped Mathematical Differentiable NMS -- same signature and returns as the
        reference lib/groomed_nms.py:10-129.

        :param scores_unsorted:          Unsorted scores of the boxes, Tensor or ndarray (N, )
        :param iou_unsorted:             Overlap matrix of the boxes, Tensor or ndarray (N, N)
        :return: valid_boxes_index   (K,)   original indices, by descending re-score
                 invalid_boxes_index (N-K,) original indices
                 (plain index tensors, as in the reference; with the module switch LAZY_INDEX_LISTS = True and GPU tensors in they
                  are LazyIndexList objects instead -- opt-in for training loops that read only the probabilities)
                 non_suppression_prob (N,)  re-scores in descending-input-score order (the reference's order)
        NumPy in -> CPU tensors out (lib/rpn_util.py:1319-1320 calls .numpy() on the result); tensors
        come back on the device of `iou_unsorted` (:60-62).  Computation always runs on the GPU.
    """
    if type(scores_unsorted) == np.ndarray:                                   # :34-36
        scores_unsorted = torch.from_numpy(scores_unsorted).float()
        iou_unsorted = torch.from_numpy(np.asarray(iou_unsorted)).float()
    elif (sorting_method != "soft" and not debug and not _PLAIN_COUNTS and scores_unsorted.is_cuda and iou_unsorted.is_cuda
          and scores_unsorted.dtype == torch.float32 and iou_unsorted.dtype == torch.float32 and scores_unsorted.dim() == 1
          and scores_unsorted.shape[0] > 0 and iou_unsorted.device == scores_unsorted.device):
        # GPU tensors in, hard sort (the training call site, lib/loss/rpn_3d.py:791): the whole call in the C++ binding
        if iou_unsorted.dim() != 2 or iou_unsorted.shape[0] != scores_unsorted.shape[0] or iou_unsorted.shape[1] != scores_unsorted.shape[0]:
            raise ValueError("iou_unsorted must be (N, N) with N = len(scores_unsorted)")       # (the same exception on every host path: ADVICE r5)
        ext = _binding()
        if ext and hasattr(ext, "single"):
            p = _params(nms_threshold, pruning_method, temperature, valid_box_prob_threshold, return_sorted_prob, group_boxes, mask_group_boxes,
                        group_size, False, True)
            try:
                o = ext.single(scores_unsorted, iou_unsorted, p.nms_threshold, p.temperature, p.valid_box_prob_threshold, p.pruning_method,
                               bool(p.return_sorted_prob), bool(p.group_boxes), bool(p.mask_group_boxes), p.group_size, False, not LAZY_INDEX_LISTS)
            except NotImplementedError:
                raise
            except RuntimeError as e:
                if isinstance(e, torch.cuda.OutOfMemoryError) or not str(e).startswith("GNMS:"):
                    raise
                raise _lib.GnmsError(str(e)) from None
            if not LAZY_INDEX_LISTS:
                return o[0], o[1], o[2]
            both = _LazyCounts(o[3], o[4])
            return LazyIndexList(o[0], both, 0, None), LazyIndexList(o[1], both, 1, None), o[2]
    out_device = iou_unsorted.device
    dev = out_device if out_device.type == "cuda" else _device()
    scores = scores_unsorted.to(device=dev, dtype=torch.float32)
    iou = iou_unsorted.to(device=dev, dtype=torch.float32)
    n = scores.shape[0]
    if iou.dim() != 2 or iou.shape[0] != n or iou.shape[1] != n:
        raise ValueError("iou_unsorted must be (N, N) with N = len(scores_unsorted)")
    kw = dict(nms_threshold=nms_threshold, pruning_method=pruning_method, temperature=temperature,
              valid_box_prob_threshold=valid_box_prob_threshold, return_sorted_prob=return_sorted_prob,
              group_boxes=group_boxes, mask_group_boxes=mask_group_boxes, group_size=group_size)
    if sorting_method == "soft":                                              # :42-45
        if sorting_temperature is None:
            sorting_temperature = temperature
        indices = torch.sort(scores.detach(), descending=True, stable=True)[1]   # :41
        if n == 0:
            soft_scores, soft_iou = scores, iou
        else:
            soft_scores, _, soft_iou = _SoftSortFunction.apply(scores, iou, sorting_temperature)
        prob, _, valid, invalid, nvalid, ninvalid = differentiable_nms_batched(
            soft_scores.unsqueeze(0), soft_iou.unsqueeze(0), presorted=True, **kw)
    else:
        prob, order, valid, invalid, nvalid, ninvalid = differentiable_nms_batched(scores.unsqueeze(0), iou.unsqueeze(0), **kw)
        indices = None
    non_suppression_prob = prob[0]
    if debug:
        print("\nInside diff NMS... After sorting")
        print(non_suppression_prob)
    if out_device == dev and LAZY_INDEX_LISTS and n > 0:
        # GPU tensors in: the two index lists have a data-dependent length K, which only the host can turn into a tensor shape.  They
        # come back as LazyIndexList objects that hold the padded device list and its device-side count and become real tensors on
        # first use -- training reads only the probabilities (lib/loss/rpn_3d.py:791 takes `[2]`), so its step never waits for the GPU.
        both = _LazyCounts(nvalid, ninvalid)
        return (LazyIndexList(valid[0], both, 0, indices), LazyIndexList(invalid[0], both, 1, indices), non_suppression_prob)
    counts = _counts_to_host(nvalid, ninvalid) if n > 0 else [0, 0]   # the one host round trip: K is data dependent (B = 1: [nvalid, ninvalid])
    valid_boxes_index = valid[0, :counts[0]]
    invalid_boxes_index = invalid[0, :counts[1]]
    if indices is not None:
        valid_boxes_index = indices[valid_boxes_index]
        invalid_boxes_index = indices[invalid_boxes_index]
    if out_device != dev:
        valid_boxes_index = valid_boxes_index.to(out_device)
        invalid_boxes_index = invalid_boxes_index.to(out_device)
        non_suppression_prob = non_suppression_prob.to(out_device)
    return valid_boxes_index, invalid_boxes_index, non_suppression_prob


_PLAIN_COUNTS = __import__("os").environ.get("GNMS_COUNTS_MAILBOX", "") == "0"


def _counts_to_host(nvalid, ninvalid):
    """[nvalid[0..B), ninvalid[0..B)] as Python ints: the one host round trip the reference's return convention needs
    (lib/groomed_nms.py:120-127), through gnms_counts_to_host (a tag-polled slot of pinned memory, no device-to-host copy)."""
    nv, ni = nvalid.contiguous(), ninvalid.contiguous()
    if _PLAIN_COUNTS:                                              # GNMS_COUNTS_MAILBOX=0 (developer A/B): torch's device-to-host copy
        return torch.cat([nv, ni]).tolist()
    ext = _binding()
    if ext and hasattr(ext, "counts_to_host"):
        try:
            return ext.counts_to_host(nv, ni)
        except RuntimeError as e:
            if not str(e).startswith("GNMS:"):
                raise
            raise _lib.GnmsError(str(e)) from None
    import ctypes
    lib = _lib.load()
    b = nv.shape[0]
    host = (ctypes.c_int32 * (2 * b))()
    with _lib.on_device(nv.device):
        _lib.check(lib.gnms_counts_to_host(nv.data_ptr(), ni.data_ptr(), b, ctypes.cast(host, ctypes.c_void_p), _lib.stream_ptr(nv.device)),
                   "gnms_counts_to_host")
    return list(host)


counts_to_host = _counts_to_host      # public name: the padded lists of the batched entries cut to size on the host, `valid[b, :counts_to_host(nvalid, ninvalid)[b]]`


# Module switch, OFF by default: differentiable_nms returns plain index tensors exactly like the reference (one host sync per call: their
# length K is data dependent).  A training loop that reads only the probabilities (lib/loss/rpn_3d.py:791 takes `[2]`) may set it to
# True: with GPU tensors in, the two lists then come back as LazyIndexList objects and the call never waits for the GPU.
LAZY_INDEX_LISTS = False


class _LazyCounts:
    """the device-side counts of one call, fetched once (for both lists) on first use"""
    __slots__ = ("_nv", "_ni", "_host")

    def __init__(self, nvalid, ninvalid):
        self._nv, self._ni, self._host = nvalid, ninvalid, None

    def get(self, which):
        if self._host is None:
            self._host = _counts_to_host(self._nv, self._ni)          # [nvalid[0], ninvalid[0]] (B = 1)
            self._nv = self._ni = None
        return int(self._host[which])


class LazyIndexList:
    """Opt-in stand-in (LAZY_INDEX_LISTS = True) for a 1-D int64 index tensor whose LENGTH is still on the GPU.  Holds the padded list
    and the device-side counts; `.t` is the real tensor -- the first access copies the count to the host, the one synchronisation the
    reference's return convention needs.  Attribute access, torch functions, arithmetic, comparisons, len(), iteration and NumPy
    conversion forward to `.t`.

    It is deliberately NOT a sequence (no __getitem__): used as an index, `boxes[lazy]` / `mask[lazy] = 1`, torch then hands the
    operation to __torch_function__, which substitutes `.t` -- with a __getitem__ it walked the object as a sequence of per-dimension
    indices instead (an IndexError on 1-D data, silently wrong data on tensors with enough dimensions).  Element access: `lazy.t[i]`.
    `isinstance(lazy, torch.Tensor)` is False; code that needs a real tensor takes `lazy.t`."""

    __slots__ = ("_padded", "_counts", "_which", "_map", "_t")

    def __init__(self, padded, counts, which, index_map=None):
        self._padded, self._counts, self._which, self._map, self._t = padded, counts, which, index_map, None

    @property
    def t(self):
        if self._t is None:
            k = self._counts.get(self._which)
            t = self._padded[:k]
            if self._map is not None:
                t = self._map[t]
            self._t = t
            self._padded = self._counts = self._map = None
        return self._t

    def __getattr__(self, name):                       # everything a tensor has: shape, device, cpu(), numpy(), tolist(), ...
        return getattr(self.t, name)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        def unwrap(x):
            if isinstance(x, LazyIndexList):
                return x.t
            if isinstance(x, (list, tuple)):
                return type(x)(unwrap(y) for y in x)
            return x
        return func(*unwrap(args), **{k: unwrap(v) for k, v in (kwargs or {}).items()})

    def __len__(self):
        return len(self.t)

    def __iter__(self):
        return iter(self.t)

    def __array__(self, dtype=None, copy=None):
        a = self.t.cpu().numpy()
        return a.astype(dtype) if dtype is not None else a

    def __repr__(self):
        return "LazyIndexList(%r)" % (self.t,) if self._t is not None else "LazyIndexList(<length still on the GPU>)"

    __hash__ = None


def _forward_dunder(name):
    def op(self, *args):
        return getattr(self.t, name)(*[a.t if isinstance(a, LazyIndexList) else a for a in args])
    op.__name__ = name
    return op


for _n in ("add", "sub", "mul", "floordiv", "truediv", "mod", "and", "or", "xor", "lshift", "rshift"):
    setattr(LazyIndexList, "__%s__" % _n, _forward_dunder("__%s__" % _n))
    setattr(LazyIndexList, "__r%s__" % _n, _forward_dunder("__r%s__" % _n))
for _n in ("lt", "le", "gt", "ge", "eq", "ne", "neg", "invert", "abs"):
    setattr(LazyIndexList, "__%s__" % _n, _forward_dunder("__%s__" % _n))
del _n


class GroomedNMS(torch.nn.Module):
    """nn.Module wrapper of the batched layer (parameter-free)."""

    def __init__(self, **kwargs):
        super().__init__()
        self.kwargs = kwargs

    def forward(self, scores, iou, counts=None):
        return differentiable_nms_batched(scores, iou, counts, **self.kwargs)


def soft_sort(scores, full_matrix=None, temperature=0.01):
    """lib/groomed_nms.py:131-165.  Returns (soft_sorted_scores, convex_comb_matrix[, soft_sorted_matrix])."""
    out_device = scores.device
    dev = out_device if out_device.type == "cuda" else _device()
    s = scores.to(device=dev, dtype=torch.float32)
    m = full_matrix.to(device=dev, dtype=torch.float32) if full_matrix is not None else None
    res = _SoftSortFunction.apply(s, m, temperature)
    return tuple(r.to(out_device) for r in res)


def pruning_function(iou, nms_threshold=0.4, temperature=0.01, pruning_method="linear"):
    """lib/groomed_nms.py:167-189.  Tensors run through the HIP kernel; an ndarray takes the reference's
    own NumPy branch (a plotting helper, plot/plot_nms_overlap_function.py:42-91) and returns an ndarray."""
    if pruning_method not in _PRUNE:
        raise NotImplementedError("Pruning method not implemented!")
    if type(iou) == np.ndarray:
        if pruning_method == "sigmoidal":
            return sigmoid_numpy((iou - nms_threshold) / temperature)
        if pruning_method == "linear":
            return iou
        return 1 - np.exp(-np.power(iou, 2) / temperature)
    if pruning_method == "linear":
        return iou                                                            # :173-174 (same tensor, as the reference)
    out_device = iou.device
    dev = out_device if out_device.type == "cuda" else _device()
    x = iou.to(device=dev, dtype=torch.float32)                           # differentiable moves: the gradient flows back to `iou`
    return _PruneFunction.apply(x, nms_threshold, temperature, _PRUNE[pruning_method]).to(out_device)


def sigmoid_numpy(x):
    """lib/groomed_nms.py:191-199 (NumPy, 1-D input as in the reference)."""
    y = np.zeros(x.shape)
    index_gt = np.where(x > 0)[0]
    y[index_gt] = 1.0 / (1 + np.exp(-x[index_gt]))
    index_lt = np.where(x <= 0)[0]
    y[index_lt] = np.exp(x[index_lt]) / (1 + np.exp(x[index_lt]))
    return y


def cast_to_cpu_cuda_tensor(input, reference_tensor):
    """lib/groomed_nms.py:201-206."""
    if reference_tensor.is_cuda and not input.is_cuda:
        input = input.cuda()
    if not reference_tensor.is_cuda and input.is_cuda:
        input = input.cpu()
    return input


def get_groups(iou_unsorted, group_threshold, scores_unsorted, group_size=100, return_original_indices=True):
    """lib/groomed_nms.py:208-270: list of LongTensors, one per group, in creation order; each holds the
    group's boxes by descending score (original indices, or score-rank positions when
    return_original_indices=False).  Grouping runs in the HIP kernels (bit matrix, leader scan,
    attribution, cap); only the ragged Python list is assembled on the host."""
    lib = _lib.load()
    out_device = iou_unsorted.device
    dev = out_device if out_device.type == "cuda" else _device()
    scores = scores_unsorted.detach().to(device=dev, dtype=torch.float32).contiguous()
    iou = iou_unsorted.detach().to(device=dev, dtype=torch.float32).contiguous()
    n = scores.shape[0]
    if n == 0:
        return []
    group_of = torch.empty((n,), dtype=torch.int32, device=dev)
    pos = torch.empty((n,), dtype=torch.int32, device=dev)
    ngroups = torch.zeros((1,), dtype=torch.int32, device=dev)
    one = GnmsParams()
    lib.gnms_default_params(ctypes.byref(one))
    ws = torch.empty((max(lib.gnms_workspace_bytes(1, n, ctypes.byref(one)), 256),), dtype=torch.uint8, device=dev)
    with on_device(dev):
        check(lib.gnms_get_groups(ptr(scores), ptr(iou), n, n, float(group_threshold), int(min(int(group_size), 2 ** 31 - 2)),
                                  ptr(group_of), ptr(pos), ptr(ngroups), ptr(ws), ws.numel(), stream_ptr(dev)), "gnms_get_groups")
    g = int(ngroups.item())
    group_of_h = group_of.cpu().numpy()
    pos_h = pos.cpu().numpy()
    ids = np.arange(n)
    if not return_original_indices:
        rank = torch.empty(n, dtype=torch.int64)
        rank[torch.sort(scores.cpu(), descending=True, stable=True)[1]] = torch.arange(n)
        ids = rank.numpy()
    groups = [[] for _ in range(g)]
    for i in np.argsort(pos_h, kind="stable"):
        if group_of_h[i] >= 0:
            groups[group_of_h[i]].append(int(ids[i]))
    return [torch.tensor(x, dtype=torch.int64, device=out_device) for x in groups]


def indices_copy(A, B, indA, indB=None, inplace=True):
    """lib/groomed_nms.py:272-337: copy B into A at 2-D positions; a 1-D indA expands to indA x indA (:301-307).
    Kept for API completeness (the layer itself never builds the N x N inversion matrix)."""
    shapeA = A.shape
    if not A.is_contiguous():
        A = A.contiguous()
    if not B.is_contiguous():
        B = B.contiguous()
    if indA.dim() == 1:
        idx = indA.detach().cpu().numpy()
        indA = torch.tensor(list(itertools.product(idx, idx)), dtype=torch.int64).reshape(-1, 2).to(A.device)
    if indB is None:
        indB = torch.tensor(list(itertools.product(np.arange(B.shape[0]), np.arange(B.shape[1]))), dtype=torch.int64).reshape(-1, 2)
    indB = indB.to(A.device)
    tailA = A.shape[2:]
    vA = A.reshape((A.shape[0] * A.shape[1],) + tuple(tailA))
    vB = B.reshape((B.shape[0] * B.shape[1],) + tuple(B.shape[2:]))
    if not inplace:
        vA = vA.clone()
    linA = indA[:, 0] * A.shape[1] + indA[:, 1]
    linB = indB[:, 0] * B.shape[1] + indB[:, 1]
    vA.index_copy_(0, linA, vB.index_select(0, linB))
    return vA.view(shapeA)
