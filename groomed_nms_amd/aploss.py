"""After-NMS AP loss on MI355X -- host-side mirror of the reference module lib/loss/aploss.py.

`APLoss` / `backpropAPLoss` keep the reference's names, arguments, defaults and return shapes
(lib/loss/aploss.py:12-97; used at lib/loss/rpn_3d.py:189 and :1117-1131 on the scores GrooMeD-NMS
rescored).  The ranking itself -- the reference's Python loop over the positives, O(N) tensor ops per
trip (:50-68) -- runs behind `gnms_aploss` (include/groomed_nms_hip.h): one workgroup per image up to 2047 boxes,
four launches that spread the positives over the machine above that (up to 16384 boxes per image).
No CPU implementation lives here: without the library or a GPU the calls raise.
"""
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr, on_device

__all__ = ["APLoss", "backpropAPLoss", "ap_loss_batched"]

MAX_BOXES = 16384                                      # GNMS_APLOSS_MAX_BOXES = GNMS_MAX_BOXES


def _device():
    if not torch.cuda.is_available():
        raise _lib.GnmsError("the AP loss needs an AMD GPU (torch.cuda.is_available() is False); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _binding():
    from .groomed_nms import _binding as b
    return b()


def _launch(logits2d, targets2d, counts, positive_label, negative_label):
    """logits2d/targets2d: contiguous fp32 [B][N] on the GPU -> (loss [B], grad [B][N])."""
    lib = _lib.load()
    B, N = logits2d.shape
    dev = logits2d.device
    loss = torch.empty((B,), dtype=torch.float32, device=dev)
    grad = torch.empty((B, N), dtype=torch.float32, device=dev)
    with on_device(dev):
        check(lib.gnms_aploss(ptr(logits2d), ptr(targets2d), B, N, ptr(counts), float(positive_label), float(negative_label),
                              ptr(loss), ptr(grad), stream_ptr()), "gnms_aploss")
    return loss, grad


class backpropAPLoss(torch.autograd.Function):
    """lib/loss/aploss.py:12-85.  forward computes loss AND d loss/d logits; backward scales the stored gradient."""

    @staticmethod
    def forward(ctx, logits, targets, delta=1.0, positive_label=1, negative_label=0):
        # delta is ignored exactly as in the reference, which overwrites it with 1.0 (:16)
        was_cuda = logits.is_cuda
        dev = logits.device if was_cuda else _device()
        shape = logits.shape
        lg = logits.detach().to(device=dev, dtype=torch.float32).reshape(1, -1).contiguous()
        tg = targets.detach().to(device=dev, dtype=torch.float32).reshape(1, -1).contiguous()
        n = lg.shape[1]
        if n == 0:
            raise RuntimeError("max(): Expected reduction dim to be specified for input.numel() == 0")   # torch.max(targets), :26
        loss, grad = _launch(lg, tg, None, positive_label, negative_label)
        grad = grad.reshape(shape)
        no_positive = bool((tg.max() <= 0).item())      # the reference branches on the same host-side test (:26)
        if not was_cuda:
            loss, grad = loss.cpu(), grad.cpu()
        ctx.grad = grad.to(logits.dtype)
        if no_positive:
            return loss.to(torch.float32)                # zeros, shape (1,) (:19, :28)
        return loss.reshape(()).to(torch.float32)        # 1 - metric.squeeze() (:78)

    @staticmethod
    def backward(ctx, grad_output):
        return ctx.grad * grad_output, None, None, None, None                                               # :80-85


class _APLossBatched(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, counts, positive_label, negative_label):
        lg = logits.detach().to(torch.float32).contiguous()
        tg = targets.detach().to(torch.float32).contiguous()
        loss, grad = _launch(lg, tg, counts, positive_label, negative_label)
        ctx.save_for_backward(grad)
        ctx.in_dtype = logits.dtype
        return loss

    @staticmethod
    def backward(ctx, grad_output):
        (grad,) = ctx.saved_tensors
        return (grad * grad_output.reshape(-1, 1)).to(ctx.in_dtype), None, None, None, None


def ap_loss_batched(logits, targets, active=None, counts=None, positive_label=1, negative_label=0):
    """The per-image loop of lib/loss/rpn_3d.py:1121-1127 as one launch: logits/targets [B][N] on the GPU -> loss [B].

    `active` (bool [B][N], the `accept_prob_active_img` mask of :1126) removes boxes from the ranking without
    compacting them: their target is replaced by a label that is neither positive nor negative, which the kernel
    ignores exactly like the reference ignores other labels (:30-31, :35).  `counts` (int32 [B]) bounds ragged
    images.  An image without a positive yields loss 0 and no gradient (:26-28).  No host synchronisation."""
    if not logits.is_cuda:
        raise _lib.GnmsError("ap_loss_batched expects GPU tensors")
    if logits.dim() != 2 or targets.shape != logits.shape:
        raise ValueError("logits and targets must both be [B][N]")
    tg = targets.detach().to(device=logits.device, dtype=torch.float32)
    if active is not None:
        ignore = float(min(positive_label, negative_label)) - 1.0
        tg = torch.where(active.to(logits.device), tg, torch.full_like(tg, ignore))
    if counts is not None:
        counts = counts.to(device=logits.device, dtype=torch.int32).contiguous()
    ext = _binding()
    if ext:                                              # the C++ autograd node (csrc/torch_binding.cpp); below: the ctypes path, same C ABI
        try:
            return ext.aploss(logits, tg, counts, float(positive_label), float(negative_label))
        except RuntimeError as e:
            if isinstance(e, torch.cuda.OutOfMemoryError) or not str(e).startswith("GNMS:"):
                raise
            raise _lib.GnmsError(str(e)) from None
    return _APLossBatched.apply(logits, tg, counts, positive_label, negative_label)


class APLoss(torch.nn.Module):
    """lib/loss/aploss.py:87-97."""

    def __init__(self, delta=1.0, positive_label=1, negative_label=0):
        super().__init__()
        self.delta = delta
        self.positive_label = positive_label
        self.negative_label = negative_label

    def forward(self, logits, targets):
        return backpropAPLoss.apply(logits, targets, self.delta, self.positive_label, self.negative_label)
