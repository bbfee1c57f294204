"""ctypes front-end of oracle/libgnms_oracle.so -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.  The
product package groomed_nms_amd/ never does (tests/test_abi_and_host.py::test_product_never_touches_the_oracle enforces it).
Every function is the CPU restatement of a reference function; the citation is in
oracle/gnms_oracle.c next to the C body.  Parity status: pinned (tests/test_oracle_golden.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgnms_oracle.so")
PRUNE = {"linear": 0, "sigmoidal": 1, "soft_nms": 2}


class _Params(ctypes.Structure):
    _fields_ = [("nms_threshold", ctypes.c_float), ("temperature", ctypes.c_float),
                ("valid_box_prob_threshold", ctypes.c_float),
                ("pruning_method", ctypes.c_int), ("return_sorted_prob", ctypes.c_int),
                ("group_boxes", ctypes.c_int), ("mask_group_boxes", ctypes.c_int), ("presorted", ctypes.c_int),
                ("group_size", ctypes.c_int64)]


def build(force=False):
    src = os.path.join(_HERE, "gnms_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.gnms_oracle_get_groups.restype = ctypes.c_int64
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t=ctypes.c_float):
    return a.ctypes.data_as(ctypes.POINTER(t)) if a is not None else None


def argsort_desc(v):
    v = _f32(v)
    out = np.zeros(len(v), np.int64)
    lib().gnms_oracle_argsort_desc(_p(v), ctypes.c_int64(len(v)), _p(out, ctypes.c_int64))
    return out


def pruning_function(x, nms_threshold=0.4, temperature=0.01, pruning_method="linear"):
    if pruning_method not in PRUNE:
        raise NotImplementedError("Pruning method not implemented!")
    x = _f32(x)
    out = np.empty_like(x)
    lib().gnms_oracle_prune(_p(x), ctypes.c_int64(x.size), ctypes.c_float(nms_threshold), ctypes.c_float(temperature),
                            PRUNE[pruning_method], _p(out))
    return out


def iou2d(a, b):
    a, b = _f32(a), _f32(b)
    out = np.empty((len(a), len(b)), np.float32)
    lib().gnms_oracle_iou2d(_p(a), ctypes.c_int64(len(a)), _p(b), ctypes.c_int64(len(b)), _p(out))
    return out


def iou2d_f64(a, b):
    """lib/core.py:205-207, 499-513, the NumPy branch on float64 boxes (the inference call site, lib/rpn_util.py:1295):
    out[i][j] = IoU(a_i, b_j) in IEEE double, operation for operation."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    max_xy = np.minimum(a[:, 2:4], np.expand_dims(b[:, 2:4], axis=1))       # [b][a]
    min_xy = np.maximum(a[:, 0:2], np.expand_dims(b[:, 0:2], axis=1))
    inter = np.clip(max_xy - min_xy, a_min=0, a_max=None)
    inter = inter[:, :, 0] * inter[:, :, 1]
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    union = np.expand_dims(area_a, 0) + np.expand_dims(area_b, 1) - inter
    with np.errstate(invalid="ignore", divide="ignore"):
        return (inter / union).T


def corners_of_cuboid_numpy_branch(params):
    """lib/math_3d.py:438-490, the NumPy branch: cos / sin in the dtype of the yaw column, everything else in float64
    (`.astype(float)` matrices, np.einsum over the three rotated axes in order)."""
    p = np.asarray(params)
    x, y, z, w, h, l, ry = [p[:, i] for i in range(7)]
    n = p.shape[0]
    R = np.zeros((n, 3, 3)).astype(float)
    R[:, 0, 0] = np.cos(ry); R[:, 0, 2] = np.sin(ry); R[:, 1, 1] = 1.0; R[:, 2, 0] = -np.sin(ry); R[:, 2, 2] = np.cos(ry)
    c = np.zeros((n, 3, 8)).astype(float)
    c[:, 0, [1, 3, 5, 6]] = l[:, np.newaxis]
    c[:, 1, [2, 3, 6, 7]] = h[:, np.newaxis]
    c[:, 2, [4, 5, 6, 7]] = w[:, np.newaxis]
    c[:, 0] -= l[:, np.newaxis] / 2
    c[:, 1] -= h[:, np.newaxis] / 2
    c[:, 2] -= w[:, np.newaxis] / 2
    out = np.zeros((n, 3, 8))
    for j in range(3):
        out[:, j] = (R[:, j, 0:1] * c[:, 0] + R[:, j, 1:2] * c[:, 1]) + R[:, j, 2:3] * c[:, 2]
    out[:, 0] += x[:, np.newaxis]
    out[:, 1] += y[:, np.newaxis]
    out[:, 2] += z[:, np.newaxis]
    return out


def corners_of_cuboid(params):
    params = _f32(params)
    out = np.empty((len(params), 3, 8), np.float32)
    lib().gnms_oracle_corners(_p(params), ctypes.c_int64(len(params)), _p(out))
    return out


def iou3d_approximate(ca, cb, generalized=False):
    ca, cb = _f32(ca), _f32(cb)
    bev = np.empty((len(ca), len(cb)), np.float32)
    i3 = np.empty((len(ca), len(cb)), np.float32)
    lib().gnms_oracle_iou3d(_p(ca), ctypes.c_int64(len(ca)), _p(cb), ctypes.c_int64(len(cb)), int(bool(generalized)),
                            _p(bev), _p(i3))
    return bev, i3


def get_groups(iou, group_threshold, scores, group_size=100):
    iou, scores = _f32(iou), _f32(scores)
    n = len(scores)
    flat = np.zeros(max(n, 1), np.int64)
    lens = np.zeros(max(n, 1), np.int64)
    g = lib().gnms_oracle_get_groups(_p(iou), _p(scores), ctypes.c_int64(n), ctypes.c_float(group_threshold),
                                     ctypes.c_int64(group_size), _p(flat, ctypes.c_int64), _p(lens, ctypes.c_int64))
    lens = lens[:g]
    out, off = [], 0
    for ln in lens:
        out.append(flat[off:off + ln].copy())
        off += ln
    return out


def _nms_core(scores, iou, grad_prob, want_grad_iou, presorted, nms_threshold, pruning_method, temperature,
              valid_box_prob_threshold, return_sorted_prob, group_boxes, mask_group_boxes, group_size):
    if pruning_method not in PRUNE:
        raise NotImplementedError("Pruning method not implemented!")
    scores, iou = _f32(scores), _f32(iou)
    n = len(scores)
    assert iou.shape == (n, n)
    P = _Params(nms_threshold, temperature, valid_box_prob_threshold, PRUNE[pruning_method], int(return_sorted_prob),
                int(bool(group_boxes)), int(bool(mask_group_boxes)), int(presorted), int(group_size))
    order = np.zeros(n, np.int64)
    prob = np.zeros(n, np.float32)
    valid = np.zeros(n, np.int64)
    invalid = np.zeros(n, np.int64)
    nvalid = ctypes.c_int64(0)
    gp = _f32(grad_prob) if grad_prob is not None else None
    gs = np.zeros(n, np.float32) if gp is not None else None
    gi = np.zeros((n, n), np.float32) if (gp is not None and want_grad_iou) else None
    rc = lib().gnms_oracle_nms(_p(scores), _p(iou), ctypes.c_int64(n), ctypes.byref(P), _p(order, ctypes.c_int64),
                               _p(prob), _p(valid, ctypes.c_int64), _p(invalid, ctypes.c_int64), ctypes.byref(nvalid),
                               _p(gp), _p(gs), _p(gi))
    if rc == -1:
        raise NotImplementedError("Pruning method not implemented!")
    nv = nvalid.value
    # NaN probabilities are in neither list (lib/groomed_nms.py:118-123), so count the invalid ones
    n_inv = int(np.sum(~np.isnan(prob))) - nv if not np.isnan(prob).any() else None
    if n_inv is None:
        n_inv = n - nv - int(np.isnan(prob).sum())
    return dict(order=order, prob=prob, valid=valid[:nv].copy(), invalid=invalid[:n_inv].copy(),
                grad_scores=gs, grad_iou=gi)


def soft_sort(scores, iou, temperature):
    scores = _f32(scores)
    n = len(scores)
    C = np.zeros((n, n), np.float32)
    ss = np.zeros(n, np.float32)
    iou_c = _f32(iou) if iou is not None else None
    sm = np.zeros((n, n), np.float32) if iou is not None else None
    lib().gnms_oracle_soft_sort(_p(scores), _p(iou_c), ctypes.c_int64(n), ctypes.c_float(temperature), _p(C), _p(ss), _p(sm))
    return ss, C, sm


def _soft_sort_backward(scores, iou, temperature, C, g_soft, g_mat):
    """Hand-derived backward of lib/groomed_nms.py:131-165 (double precision).
    Returns (dL/dscores, dL/diou) for upstream grads g_soft (n,) and g_mat (n,n)."""
    s = scores.astype(np.float64)
    C = C.astype(np.float64)
    n = len(s)
    order = argsort_desc(scores)
    shat = s[order]
    A = -np.abs(s[None, :] - shat[:, None])
    mx = A.max(1)
    E = np.exp((A - mx[:, None]) / temperature)
    Z = E.sum(1) + 1e-3
    dC = np.outer(g_soft, s)
    d_s = C.T @ g_soft
    d_iou = None
    if g_mat is not None:
        dC = dC + g_mat.astype(np.float64) @ iou.astype(np.float64).T
        d_iou = C.T @ g_mat.astype(np.float64)
    # C[i][j] = E[i][j] / Z[j]  (the reference's last-axis broadcast, lib/groomed_nms.py:155)
    dZ = -((dC * C).sum(0) / Z)
    dE = dC / Z[None, :] + dZ[:, None]
    dArg = dE * E / temperature            # d/d(A - mx)
    dA = dArg.copy()
    dmx = -dArg.sum(1)
    amax = A.argmax(1)
    dA[np.arange(n), amax] += dmx
    sg = np.sign(s[None, :] - shat[:, None])
    d_s = d_s + (dA * (-sg)).sum(0)
    d_shat = (dA * sg).sum(1)
    np.add.at(d_s, order, d_shat)
    return d_s.astype(np.float32), (d_iou.astype(np.float32) if d_iou is not None else None)


def differentiable_nms(scores_unsorted, iou_unsorted, nms_threshold=0.4, pruning_method="linear", temperature=0.01,
                       valid_box_prob_threshold=0.3, return_sorted_prob=False, sorting_method="hard",
                       sorting_temperature=None, group_boxes=True, mask_group_boxes=True, group_size=100,
                       grad_prob=None, want_grad_iou=False):
    """CPU oracle of lib/groomed_nms.py:10-129.  Returns a dict: valid, invalid, prob, order and, when
    grad_prob (dL/dprob) is given, grad_scores [and grad_iou]."""
    scores = _f32(scores_unsorted)
    iou = _f32(iou_unsorted)
    kw = dict(nms_threshold=nms_threshold, pruning_method=pruning_method, temperature=temperature,
              valid_box_prob_threshold=valid_box_prob_threshold, return_sorted_prob=return_sorted_prob,
              group_boxes=group_boxes, mask_group_boxes=mask_group_boxes, group_size=group_size)
    if sorting_method != "soft":
        return _nms_core(scores, iou, grad_prob, want_grad_iou, 0, **kw)
    if sorting_temperature is None:
        sorting_temperature = temperature
    indices = argsort_desc(scores)                                      # :41
    ss, C, sm = soft_sort(scores, iou, sorting_temperature)             # :45
    res = _nms_core(ss, sm, grad_prob, True, 1, **kw)
    res["valid"] = indices[res["valid"]]
    res["invalid"] = indices[res["invalid"]]
    res["order"] = indices
    if grad_prob is not None:
        gs, gi = _soft_sort_backward(scores, iou, sorting_temperature, C, res["grad_scores"], res["grad_iou"])
        res["grad_scores"], res["grad_iou"] = gs, (gi if want_grad_iou else None)
    return res


def classic_nms_sorted(sorted_dets, thresh, rule="gpu"):
    """lib/nms/nms_kernel.cu:91-144 `_nms` semantics on score-sorted boxes; rule in gpu|cpu|py."""
    d = _f32(sorted_dets)
    n, dim = d.shape if d.ndim == 2 else (0, 5)
    keep = np.zeros(max(n, 1), np.int32)
    num = ctypes.c_int32(0)
    lib().gnms_oracle_classic_nms(_p(d), ctypes.c_int64(n), ctypes.c_int64(dim), ctypes.c_float(thresh),
                                  {"gpu": 0, "cpu": 1, "py": 2}[rule], _p(keep, ctypes.c_int32), ctypes.byref(num))
    return keep[:num.value].copy()


def classic_nms(dets, thresh, rule="gpu"):
    """gpu_nms / cpu_nms / py_cpu_nms wrappers: sort by score (argsort()[::-1], gpu_nms.pyx:25-28), scan, map back."""
    dets = _f32(dets)
    if len(dets) == 0:
        return []
    order = dets[:, 4].argsort()[::-1]
    keep = classic_nms_sorted(dets[order], thresh, rule)
    return [int(i) for i in order[keep]]


def aploss(logits, targets, positive_label=1, negative_label=0):
    """lib/loss/aploss.py:14-78: returns (loss, grad) with grad = d loss / d logits (the reference stores it in forward)."""
    logits, targets = _f32(logits), _f32(targets)
    n = len(logits)
    loss = np.zeros(1, np.float32)
    grad = np.zeros(n, np.float32)
    lib().gnms_oracle_aploss(_p(logits), _p(targets), ctypes.c_int64(n), ctypes.c_float(positive_label), ctypes.c_float(negative_label),
                             _p(loss), _p(grad))
    return float(loss[0]), grad
