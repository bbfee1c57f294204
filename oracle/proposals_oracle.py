"""CPU oracle for the pieces in front of the NMS layer.  TEST INFRASTRUCTURE ONLY: imported by tests/ (and
nothing else); the product never routes through it.  Plain NumPy restatements, fp32, each citing the reference lines it follows.
Parity status: pinned by tests/golden/proposals.npz (outputs of the reference's own functions, tests/golden/make_golden.py)."""
import numpy as np

from . import oracle as O

f32 = np.float32


def bbox_transform_inv(boxes, deltas, means=None, stds=None):
    """lib/rpn_util.py:872-934.  boxes [A,4], deltas [A,4] or [B,A,4] -> same shape as deltas (deltas is not modified)."""
    boxes = np.asarray(boxes, f32)
    d = np.array(deltas, f32, copy=True)
    if boxes.shape[0] == 0:
        return np.zeros((0, d.shape[1]), f32)                                   # :881-882
    widths = boxes[:, 2] - boxes[:, 0] + f32(1.0)                               # :887
    heights = boxes[:, 3] - boxes[:, 1] + f32(1.0)                              # :888
    ctr_x = boxes[:, 0] + f32(0.5) * widths                                     # :889
    ctr_y = boxes[:, 1] + f32(0.5) * heights                                    # :890
    dx, dy, dw, dh = d[..., 0], d[..., 1], d[..., 2], d[..., 3]                 # :892-901 (views)
    if stds is not None:                                                        # :903-907
        dx *= f32(stds[0]); dy *= f32(stds[1]); dw *= f32(stds[2]); dh *= f32(stds[3])
    if means is not None:                                                       # :909-913
        dx += f32(means[0]); dy += f32(means[1]); dw += f32(means[2]); dh += f32(means[3])
    pcx = dx * widths + ctr_x                                                   # :915
    pcy = dy * heights + ctr_y                                                  # :916
    pw = np.exp(dw) * widths                                                    # :917
    ph = np.exp(dh) * heights                                                   # :918
    out = np.zeros(d.shape, f32)                                                # :920
    out[..., 0] = pcx - f32(0.5) * pw                                           # :924-934
    out[..., 1] = pcy - f32(0.5) * ph
    out[..., 2] = pcx + f32(0.5) * pw - f32(1.0)
    out[..., 3] = pcy + f32(0.5) * ph - f32(1.0)
    return out


def project_3d_points_in_4d_format(p2, points3):
    """lib/math_3d.py:47-72 with pad_ones=True.  p2 [4,4], points3 [3,M] -> [4,M]."""
    p2 = np.asarray(p2, f32)
    pts = np.vstack([np.asarray(points3, f32), np.ones((1, points3.shape[1]), f32)])     # :60-61 / :66-67
    c = (p2 @ pts).astype(f32)                                                           # :63 / :69
    ind = np.abs(c[2]) > f32(1e-2)                                                       # :64 / :70 (z_eps, :56)
    c[:2, ind] /= c[2, ind]                                                              # :72
    return c


def projected_boxes_2d(params, p2, scale=1.0):
    """lib/loss/rpn_3d.py:746-768: params [N,7] (x y z w h l ry) -> [N,4] (x1 y1 x2 y2) of the projected cuboids."""
    params = np.asarray(params, f32)
    corners = O.corners_of_cuboid(params)                                                # [N,3,8], lib/math_3d.py:364-435
    n = corners.shape[0]
    flat = corners.transpose(0, 2, 1).reshape(-1, 3).T                                   # rpn_3d.py:755
    proj = project_3d_points_in_4d_format(p2, flat)                                      # :758
    c2 = proj.T.reshape(n, 8, 4).transpose(0, 2, 1)                                      # :760
    box = np.stack([c2[:, 0].min(1), c2[:, 1].min(1), c2[:, 0].max(1), c2[:, 1].max(1)], 1)   # :762-766
    return (box * f32(scale)).astype(f32)                                                # :767


def select_topk(scores, candidates, k):
    """lib/loss/rpn_3d.py:731-737: candidates sorted by descending score (stable: ties keep candidate order), first min(k, len)."""
    scores = np.asarray(scores, f32)
    cand = np.arange(scores.shape[0]) if candidates is None else np.asarray(candidates, np.int64)
    order = O.argsort_desc(scores[cand])                                                 # torch.sort(descending=True), :732
    return cand[order[:min(k, len(order))]]                                              # :733-737


def best_targets(pred_params, pred_boxes, gt_params, gt_boxes, beta):
    """lib/loss/rpn_3d.py:801-825: -> (targets [N] in {0,1}, best_index [M] (-1: none above beta), best_score [M])."""
    pc = O.corners_of_cuboid(np.asarray(pred_params, f32))                              # :772 / :805
    gc = O.corners_of_cuboid(np.asarray(gt_params, f32))
    _, i3 = O.iou3d_approximate(pc, gc, generalized=True)                               # :813
    i2 = O.iou2d(np.asarray(pred_boxes, f32)[:, :4], np.asarray(gt_boxes, f32)[:, :4])  # :814
    score = (f32(0.5) * (f32(1.0) + i3)) * i2                                           # :819
    n, m = score.shape
    targets = np.zeros(n, f32)
    best = np.full(m, -1, np.int64)
    bscore = np.zeros(m, f32)
    for j in range(m):
        col = score[:, j]
        nan = np.isnan(col)
        i = int(np.argmax(nan)) if nan.any() else int(np.argmax(col))                   # torch.max: NaN is a maximum, first index
        bscore[j] = col[i]
        if col[i] > beta:                                                               # :822
            best[j] = i
            targets[i] = 1                                                              # :826
    return targets, best, bscore
