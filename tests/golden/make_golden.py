#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/ by IMPORTING the reference.

Runs ONLY in the build container (needs /root/reference, read-only).  Nothing of the reference
travels: this script writes *data* (inputs + the reference's outputs) as .npz fixtures.  The GPU
box, the tests, bench.py and smoke() only ever read the .npz files.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Reference entry points exercised (file:line under /root/reference):
  lib/groomed_nms.py:10   differentiable_nms      lib/groomed_nms.py:208  get_groups
  lib/groomed_nms.py:167  pruning_function        lib/groomed_nms.py:131  soft_sort
  lib/groomed_nms.py:272  indices_copy
  lib/core.py:480         iou                     lib/core.py:305         iou3d_approximate
  lib/math_3d.py:364      get_corners_of_cuboid
  lib/nms/py_cpu_nms.py:10 py_cpu_nms             lib/nms_others.py:6,119 navneeth_soft_nms, girshick_nms
  lib/loss/aploss.py:14   backpropAPLoss / APLoss (the consumer of the rescored scores)
  lib/rpn_util.py:872     bbox_transform_inv      lib/math_3d.py:47       project_3d_points_in_4D_format   
  lib/rpn_util.py:1489,1571,2013  convert_image_predictions_to_correct_entries, get_text_to_write_in_kitti_format,
                                  parse_kitti_result
Known-answer vectors KAT-1/KAT-2 come from test/test_differentiable_nms_forward.py:127-140.

Inputs are stored next to outputs: RNG streams differ across library versions, so nothing is ever
re-drawn from a seed on the GPU box.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


# ----------------------------------------------------------------------------------------------
# harness: import the reference without touching it
# ----------------------------------------------------------------------------------------------
def load_reference():
    # torch>=2 rejects uint8 masks; the reference builds one at lib/groomed_nms.py:56 and uses it
    # at :73.  Out-of-tree shim, applied before any call.
    orig = torch.Tensor.masked_fill_

    def masked_fill_compat(self, mask, value):
        return orig(self, mask.bool() if mask.dtype == torch.uint8 else mask, value)

    torch.Tensor.masked_fill_ = masked_fill_compat

    spec = importlib.util.spec_from_file_location("ref_groomed_nms", REF + "/lib/groomed_nms.py")
    gn = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gn)

    # lib/core.py, lib/math_3d.py import cv2/easydict/shapely/visdom at module top
    class _Stub(types.ModuleType):          # any attribute (cv2.FONT_*, EasyDict, Polygon ...) resolves
        def __getattr__(self, key):
            if key.startswith("__"):
                raise AttributeError(key)
            return object

    for name in ("cv2", "easydict", "shapely", "shapely.geometry", "visdom"):
        if name not in sys.modules:
            sys.modules[name] = _Stub(name)
    sys.path.insert(0, REF)
    import lib.core as core          # noqa: E402
    import lib.math_3d as math_3d    # noqa: E402
    import lib.nms_others as nms_others  # noqa: E402
    spec = importlib.util.spec_from_file_location("ref_py_cpu_nms", REF + "/lib/nms/py_cpu_nms.py")
    pcn = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pcn)
    return gn, core, math_3d, nms_others, pcn


gn, core, math_3d, nms_others, pcn = load_reference()


def tie_free_scores(rng, n, lo=0.0, hi=1.0):
    """fp32 scores distinct by construction (the reference's tie order is implementation-defined)."""
    while True:
        s = rng.uniform(lo, hi, size=n).astype(np.float32)
        if len(np.unique(s)) == n:
            return s


# ----------------------------------------------------------------------------------------------
# input generators
# ----------------------------------------------------------------------------------------------
def uniform_boxes_2d(rng, n):
    c = np.stack([rng.uniform(0, 1760, n), rng.uniform(0, 512, n)], 1)
    wh = rng.uniform(16, 136, size=(n, 2))
    return np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)


def clustered_boxes_2d(rng, n, per=16):
    k = max(1, n // per)
    base = uniform_boxes_2d(rng, k).astype(np.float64)
    bc = (base[:, :2] + base[:, 2:]) / 2
    bs = base[:, 2:] - base[:, :2]
    which = np.arange(n) % k
    c = bc[which] + rng.normal(0, 0.1, size=(n, 2)) * bs[which]
    s = bs[which] * np.exp(rng.normal(0, 0.1, size=(n, 2)))
    out = np.concatenate([c - s / 2, c + s / 2], 1).astype(np.float32)
    return out[rng.permutation(n)]


def boxes_3d(rng, n, clustered=False, per=8):
    def draw(m):
        return np.stack([rng.uniform(-30, 30, m), rng.uniform(0.5, 2.5, m), rng.uniform(5, 60, m),
                         rng.uniform(1.4, 2.0, m), rng.uniform(1.3, 2.0, m), rng.uniform(3, 5, m),
                         rng.uniform(-np.pi, np.pi, m)], 1)   # x y z w h l ry
    if not clustered:
        return draw(n).astype(np.float32)
    k = max(1, n // per)
    base = draw(k)
    which = np.arange(n) % k
    p = base[which].copy()
    p[:, :3] += rng.normal(0, 0.15, size=(n, 3))
    p[:, 3:6] *= np.exp(rng.normal(0, 0.05, size=(n, 3)))
    p[:, 6] += rng.normal(0, 0.05, size=n)
    return p[rng.permutation(n)].astype(np.float32)


def random_iou_like_reference_test(rng, n, symmetric=False):
    """test/test_differentiable_nms_forward.py:16-27: iou ~ U(0,1), unit diagonal."""
    m = rng.uniform(0, 1, size=(n, n)).astype(np.float32)
    if symmetric:
        m = (0.5 * (m + m.T)).astype(np.float32)
    np.fill_diagonal(m, 1.0)
    return m


def block_iou(rng, n, cuts):
    """test/test_differentiable_nms_backprop_on_subset.py:262-331: U(0.8,1) inside objects, 0 across."""
    d = rng.uniform(0.8, 1.0, size=(n, n))
    edges = [0] + list(cuts) + [n]
    blk = np.zeros((n, n), bool)
    for a, b in zip(edges[:-1], edges[1:]):
        blk[a:b, a:b] = True
    d[~blk] = 0
    np.fill_diagonal(d, 1)
    d = 0.5 * (d.T + d)
    return d.astype(np.float32)


# ----------------------------------------------------------------------------------------------
# running the reference
# ----------------------------------------------------------------------------------------------
def run_nms(scores, iou, w=None, want_iou_grad=False, **kw):
    """Reference forward (+ backward of L = sum_k w_k * prob_k).  Returns dict of numpy arrays."""
    s = torch.from_numpy(scores).clone().requires_grad_(True)
    m = torch.from_numpy(iou).clone().requires_grad_(bool(want_iou_grad))
    valid, invalid, prob = gn.differentiable_nms(s, m, **kw)
    out = {"valid": valid.numpy().astype(np.int64), "invalid": invalid.numpy().astype(np.int64),
           "prob": prob.detach().numpy().astype(np.float32)}
    if w is not None and prob.requires_grad and prob.numel() > 0:
        (prob * torch.from_numpy(w)).sum().backward()
        out["grad_scores"] = s.grad.numpy().astype(np.float32)
        if want_iou_grad:
            out["grad_iou"] = (m.grad if m.grad is not None else torch.zeros_like(m)).numpy().astype(np.float32)
    return out


MODES = [
    # tag, kwargs
    ("gm_lin",      dict(group_boxes=True,  mask_group_boxes=True,  pruning_method="linear")),
    ("gm_lin_gs2",  dict(group_boxes=True,  mask_group_boxes=True,  pruning_method="linear", group_size=2)),
    ("gm_lin_sorted", dict(group_boxes=True, mask_group_boxes=True, pruning_method="linear", return_sorted_prob=True)),
    ("gm_sig",      dict(group_boxes=True,  mask_group_boxes=True,  pruning_method="sigmoidal", temperature=0.1)),
    ("gm_soft",     dict(group_boxes=True,  mask_group_boxes=True,  pruning_method="soft_nms", temperature=0.5)),
    ("gu_lin",      dict(group_boxes=True,  mask_group_boxes=False, pruning_method="linear")),
    ("gu_lin_gs2",  dict(group_boxes=True,  mask_group_boxes=False, pruning_method="linear", group_size=2)),
    ("gu_sig",      dict(group_boxes=True,  mask_group_boxes=False, pruning_method="sigmoidal", temperature=0.1)),
    ("gu_soft",     dict(group_boxes=True,  mask_group_boxes=False, pruning_method="soft_nms", temperature=0.1)),
    ("un_lin",      dict(group_boxes=False, pruning_method="linear")),
    ("un_lin_sorted", dict(group_boxes=False, pruning_method="linear", return_sorted_prob=True)),
    ("un_sig",      dict(group_boxes=False, pruning_method="sigmoidal", temperature=0.1)),
    ("un_soft",     dict(group_boxes=False, pruning_method="soft_nms", temperature=0.5)),
    ("gm_lin_thr",  dict(group_boxes=True,  mask_group_boxes=True,  pruning_method="linear", nms_threshold=0.6,
                         valid_box_prob_threshold=0.5)),
]


def pack_case(store, name, scores, iou, modes=MODES, grad_iou=False, rng=None, extra=None):
    """Adds one input (scores, iou) and the reference's outputs under every mode."""
    n = len(scores)
    store[f"{name}/scores"] = scores
    store[f"{name}/iou"] = iou
    w = (rng.uniform(-1, 2, size=n).astype(np.float32) if rng is not None else np.ones(n, np.float32))
    store[f"{name}/w"] = w
    if extra:
        for k, v in extra.items():
            store[f"{name}/{k}"] = v
    for tag, kw in modes:
        res = run_nms(scores, iou, w=w, want_iou_grad=grad_iou, **kw)
        for k, v in res.items():
            store[f"{name}/{tag}/{k}"] = v
    # groups (lib/groomed_nms.py:208) on the score-sorted matrix, as differentiable_nms calls it (:85)
    if n > 0:
        order = torch.sort(torch.from_numpy(scores), descending=True)[1]
        ss = torch.from_numpy(scores)[order]
        mm = torch.from_numpy(iou)[order][:, order]
        for gs in (100, 2):
            groups = gn.get_groups(mm, 0.4, ss, group_size=gs)
            flat = np.concatenate([g.numpy() for g in groups]) if groups else np.zeros(0, np.int64)
            lens = np.array([len(g) for g in groups], np.int64)
            store[f"{name}/groups_gs{gs}/flat"] = flat.astype(np.int64)
            store[f"{name}/groups_gs{gs}/lens"] = lens


def make_f64_call_site():
    """The reference's INFERENCE call site, lib/rpn_util.py:1292-1320, restated on synthetic proposals: `aboxes` (float64 after the
    hstack at :1258, sorted by descending score, :1260-1262) -> lib.core.iou NumPy branch in float64 (:1295) -> differentiable_nms with
    NumPy arguments (:1319, which rounds to fp32 once, lib/groomed_nms.py:36); and, overlap_in_nms == "3d" (:1300-1314):
    get_corners_of_cuboid NumPy branch on the float32 `coords_3d_raw` (:1303-1309, float64 corners) -> .float() -> iou3d_approximate
    -> 0.5 * (1 + giou) -> NumPy -> differentiable_nms.  N = 500 (:1293), default NMS parameters.
    Inputs are stored as fp32 and the float64 proposals are DEFINED from them by one IEEE double multiplication (coordinates that are
    not fp32-representable, like get_2D_from_3D's); matrices are pinned by the SHA-256 of the reference's bytes."""
    import hashlib
    rng = np.random.default_rng(20260928)
    n = 500
    out = {}
    # ---- 2D: 200 cases ----
    kinds = [("uniform", None)] * 60 + [("clustered", 8)] * 50 + [("clustered", 32)] * 50 + [("clustered", 125)] * 40
    b32 = np.zeros((len(kinds), n, 4), np.float32)
    s32 = np.zeros((len(kinds), n), np.float32)
    valid, off, prob, sha = [], [0], [], []
    for c, (kind, per) in enumerate(kinds):
        b = uniform_boxes_2d(rng, n) if kind == "uniform" else clustered_boxes_2d(rng, n, per)
        sc = tie_free_scores(rng, n)
        o = np.argsort(-sc, kind="stable")                                      # lib/rpn_util.py:1260-1262
        b32[c], s32[c] = b[o], sc[o]
        aboxes = np.hstack((b32[c].astype(np.float64) * F64_SCALE, s32[c].astype(np.float64)[:, np.newaxis]))    # float64, :1258
        ious = core.iou(aboxes[:, 0:4], aboxes[:, 0:4], mode='combinations')     # :1295 -- NumPy branch, float64
        assert ious.dtype == np.float64
        keep, _, scores_new = gn.differentiable_nms(scores_unsorted=aboxes[:, 4], iou_unsorted=ious, nms_threshold=0.4)   # :1319
        valid.append(keep.numpy().astype(np.int16))
        off.append(off[-1] + len(valid[-1]))
        prob.append(scores_new.numpy().astype(np.float32))
        sha.append(np.frombuffer(hashlib.sha256(np.ascontiguousarray(ious).tobytes()).digest(), np.uint8))
    out["d2/boxes32"], out["d2/scores32"] = b32, s32
    out["d2/valid"], out["d2/valid_off"] = np.concatenate(valid), np.array(off, np.int32)
    out["d2/prob"], out["d2/iou_sha256"] = np.stack(prob), np.stack(sha)
    out["f64_scale"] = np.array(F64_SCALE, np.float64)
    # ---- 3D: 60 cases ----
    kinds3 = [False] * 20 + [True] * 40
    p32 = np.zeros((len(kinds3), n, 7), np.float32)
    s3 = np.zeros((len(kinds3), n), np.float32)
    valid, off, prob, sha = [], [0], [], []
    for c, clustered in enumerate(kinds3):
        p = boxes_3d(rng, n, clustered=clustered, per=8 if c % 2 else 25)
        sc = tie_free_scores(rng, n)
        o = np.argsort(-sc, kind="stable")
        p32[c], s3[c] = p[o], sc[o]
        raw = p32[c]                                                             # coords_3d_raw: float32 (:1186, :1211)
        corners = math_3d.get_corners_of_cuboid(x3d=raw[:, 0], y3d=raw[:, 1], z3d=raw[:, 2], w3d=raw[:, 3], h3d=raw[:, 4], l3d=raw[:, 5],
                                                ry3d=raw[:, 6])                  # :1303-1309 -- NumPy branch
        assert corners.dtype == np.float64
        c32 = torch.from_numpy(corners).float()                                  # :1310
        _, i3 = core.iou3d_approximate(c32.clone(), c32.clone(), mode="combinations", method="generalized")    # :1311
        ious = (0.5 * (1 + i3)).numpy()                                          # :1312-1313
        keep, _, scores_new = gn.differentiable_nms(scores_unsorted=s3[c].astype(np.float64), iou_unsorted=ious, nms_threshold=0.4)
        valid.append(keep.numpy().astype(np.int16))
        off.append(off[-1] + len(valid[-1]))
        prob.append(scores_new.numpy().astype(np.float32))
        sha.append(np.frombuffer(hashlib.sha256(np.ascontiguousarray(c32.numpy()).tobytes()).digest(), np.uint8))
        if c < 4:
            out[f"d3/corners32_{c}"] = c32.numpy()                               # a few in full: how far the device's sin / cos are
    out["d3/params32"], out["d3/scores32"] = p32, s3
    out["d3/valid"], out["d3/valid_off"] = np.concatenate(valid), np.array(off, np.int32)
    out["d3/prob"], out["d3/corners32_sha256"] = np.stack(prob), np.stack(sha)
    np.savez_compressed(os.path.join(OUT, "f64site.npz"), **out)


F64_SCALE = 1.0 + 2.0 ** -27          # one double multiplication turns fp32 coordinates into values that are not fp32-representable


def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)

    # ------------------------------------------------------------------ NMS layer goldens
    nms = {}
    # KAT-1 / KAT-2: test/test_differentiable_nms_forward.py:127-140 (temperature=0.1 there; unused by linear)
    kat1_iou = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [.9, .9, 1, 0], [0, 0, 0, 1]], np.float32)
    kat1_s = np.array([.99, .98, .8, .7], np.float32)
    kat2_iou = np.array([[1, 0, 0, 0, 0], [0, 1, 0, 0, 0], [.9, .9, 1, 0, 0], [.9, .9, 0, 1, 0], [0, 0, .9, .9, 1]],
                        np.float32)
    kat2_s = np.array([.99, .98, .8, .7, .6], np.float32)
    rng = np.random.default_rng(1234)
    pack_case(nms, "kat1", kat1_s, kat1_iou, grad_iou=True, rng=rng)
    pack_case(nms, "kat2", kat2_s, kat2_iou, grad_iou=True, rng=rng)
    perm = np.array([3, 0, 4, 2, 1])           # permuted KAT-2: unsorted input
    pack_case(nms, "kat2_perm", kat2_s[perm], kat2_iou[perm][:, perm], grad_iou=True, rng=rng)
    store_expected = {"kat1/expected_prob": np.array([0.990, 0.980, 0.000, 0.700], np.float32),
                      "kat2/expected_prob": np.array([0.990, 0.980, 0.000, 0.000, 0.600], np.float32)}
    nms.update(store_expected)

    # edge sizes
    pack_case(nms, "n0", np.zeros(0, np.float32), np.zeros((0, 0), np.float32), modes=MODES[:1] + MODES[9:10])
    pack_case(nms, "n1", np.array([0.7], np.float32), np.ones((1, 1), np.float32), rng=rng)
    pack_case(nms, "n1_low", np.array([0.2], np.float32), np.ones((1, 1), np.float32), rng=rng)

    # random U(0,1) IoU with unit diagonal (reference test generator), asymmetric and symmetric.
    # Ungrouped / unmasked inverses of such dense matrices are ill-conditioned beyond a few boxes,
    # so the dense-random cases keep only the default (masked) modes at larger n.
    for n, sym in ((5, False), (12, True), (33, False)):
        pack_case(nms, f"rand{n}", tie_free_scores(rng, n, 0.4, 1.0), random_iou_like_reference_test(rng, n, sym),
                  grad_iou=(n <= 12), rng=rng)
    masked_only = [m for m in MODES if m[0].startswith("gm_")]
    for n, sym in ((64, True), (130, False)):
        pack_case(nms, f"rand{n}", tie_free_scores(rng, n, 0.0, 1.0), random_iou_like_reference_test(rng, n, sym),
                  modes=masked_only, rng=rng)

    # NaN off-diagonal: the box falls out of every group (lib/groomed_nms.py:249-262)
    s = np.array([0.9, 0.8, 0.7, 0.6], np.float32)
    m = np.array([[1, .5, 0, 0], [np.nan, 1, 0, 0], [0, .9, 1, 0], [0, 0, 0, 1.]], np.float32)
    pack_case(nms, "nan_offdiag", s, m, modes=MODES[:2] + MODES[5:6], rng=rng)

    # block-structured IoU (backprop test generator), 1/2/3 objects, n=25 and a cap-exceeding n=120 single object
    for tag, n, cuts in (("block1", 25, []), ("block2", 25, [5]), ("block3", 25, [5, 9]), ("block_cap", 120, [110])):
        pack_case(nms, tag, tie_free_scores(rng, n), block_iou(rng, n, cuts), grad_iou=(n <= 25), rng=rng,
                  modes=(MODES if n <= 25 else masked_only + MODES[5:7]))
    np.savez_compressed(os.path.join(OUT, "nms_small.npz"), **nms)

    # ------------------------------------------------------------------ box-derived goldens (2D)
    box = {}
    for tag, gen, n in (("uni64", uniform_boxes_2d, 64), ("clu64", clustered_boxes_2d, 64),
                        ("uni256", uniform_boxes_2d, 256), ("clu256", clustered_boxes_2d, 256),
                        ("clu250", clustered_boxes_2d, 250)):
        b = gen(rng, n)
        s = tie_free_scores(rng, n)
        m = core.iou(torch.from_numpy(b), torch.from_numpy(b), mode="combinations").numpy()
        m_np = core.iou(b, b, mode="combinations")          # numpy branch (lib/core.py:512-513), fp32 in -> fp32
        assert m_np.dtype == np.float32
        box[f"{tag}/iou_numpy_branch_maxdiff"] = np.array(np.nanmax(np.abs(m_np - m)), np.float32)
        modes = MODES if n <= 64 else [m_ for m_ in MODES if not m_[0].endswith("_sorted")]
        pack_case(box, tag, s, m, modes=modes, rng=rng, extra={"boxes": b})
    # rectangular overlap (M != N) as used for best-target assignment (lib/loss/rpn_3d.py:801-825)
    a = uniform_boxes_2d(rng, 37)
    g = np.concatenate([a[:5] + rng.normal(0, 3, size=(5, 4)).astype(np.float32), uniform_boxes_2d(rng, 6)])
    box["rect/a"] = a
    box["rect/b"] = g
    box["rect/iou"] = core.iou(torch.from_numpy(a), torch.from_numpy(g), mode="combinations").numpy()
    # zero-area box: 0/0 -> NaN on its own diagonal (lib/core.py:507-508)
    z = uniform_boxes_2d(rng, 6)
    z[2, 2:] = z[2, :2]
    box["zero_area/boxes"] = z
    box["zero_area/iou"] = core.iou(torch.from_numpy(z), torch.from_numpy(z), mode="combinations").numpy()
    np.savez_compressed(os.path.join(OUT, "boxes_2d.npz"), **box)

    # ------------------------------------------------------------------ 3D goldens
    d3 = {}
    # test/test_get_corners_of_cuboid_numpy.py:9-17 generator (np seed 0, m=5)
    np.random.seed(0)
    m5 = 5
    p5 = dict(x=30 * np.random.uniform(size=m5), y=10 * np.random.uniform(size=m5), z=15 * np.random.uniform(size=m5),
              l=4 * np.random.uniform(size=m5), w=5 * np.random.uniform(size=m5), h=6 * np.random.uniform(size=m5),
              r=np.random.uniform(low=-1.57, high=1.57, size=m5))
    params5 = np.stack([p5["x"], p5["y"], p5["z"], p5["w"], p5["h"], p5["l"], p5["r"]], 1).astype(np.float32)
    cases = [("m5", params5), ("uni64", boxes_3d(rng, 64)), ("clu64", boxes_3d(rng, 64, clustered=True)),
             ("clu200", boxes_3d(rng, 200, clustered=True))]
    for tag, p in cases:
        t = [torch.from_numpy(np.ascontiguousarray(p[:, i])) for i in range(7)]
        corners = math_3d.get_corners_of_cuboid(t[0], t[1], t[2], t[3], t[4], t[5], t[6])   # N x 3 x 8
        d3[f"{tag}/params"] = p                      # x y z w h l ry
        d3[f"{tag}/corners"] = corners.numpy().astype(np.float32)
        # iou3d_approximate mutates its inputs (lib/core.py:379-380): pass clones
        for method in ("normal", "generalized"):
            bev, i3 = core.iou3d_approximate(corners.clone(), corners.clone(), mode="combinations", method=method)
            d3[f"{tag}/{method}/iou_bev"] = bev.numpy().astype(np.float32)
            d3[f"{tag}/{method}/iou_3d"] = i3.numpy().astype(np.float32)
        # what the callers feed the NMS: 0.5*(1+giou) (lib/loss/rpn_3d.py:781, lib/rpn_util.py:1312)
        _, gi = core.iou3d_approximate(corners.clone(), corners.clone(), mode="combinations", method="generalized")
        d3[f"{tag}/nms_overlap"] = (0.5 * (1 + gi)).numpy().astype(np.float32)
    # rectangular 3D
    pa, pb = boxes_3d(rng, 19), boxes_3d(rng, 7, clustered=True)
    ca = math_3d.get_corners_of_cuboid(*[torch.from_numpy(np.ascontiguousarray(pa[:, i])) for i in range(7)])
    cb = math_3d.get_corners_of_cuboid(*[torch.from_numpy(np.ascontiguousarray(pb[:, i])) for i in range(7)])
    d3["rect/corners_a"] = ca.numpy()
    d3["rect/corners_b"] = cb.numpy()
    bev, i3 = core.iou3d_approximate(ca.clone(), cb.clone(), mode="combinations", method="generalized")
    d3["rect/iou_bev"] = bev.numpy()
    d3["rect/iou_3d"] = i3.numpy()
    # NMS on a 3D overlap matrix
    s = tie_free_scores(rng, 200)
    pack_case(d3, "clu200_nms", s, d3["clu200/nms_overlap"], modes=[m_ for m_ in MODES if not m_[0].endswith("_sorted")],
              rng=rng)
    np.savez_compressed(os.path.join(OUT, "boxes_3d.npz"), **d3)

    # ------------------------------------------------------------------ soft sort, helpers, classical NMS
    misc = {}
    # soft sort only terminates in the reference when the input is already score-sorted and the
    # temperature is small against the score gaps: iou' = C @ iou mixes rows but leaves columns in
    # input order (lib/groomed_nms.py:164), so get_groups' "leader column" (:249) is column k of the
    # input, and a diagonal <= threshold makes its while-loop spin forever (:247-262).
    for n, temp in ((16, 0.01), (16, 0.003), (40, 0.002)):
        s = np.sort(tie_free_scores(rng, n))[::-1].copy()
        b = clustered_boxes_2d(rng, n, per=4)
        m = core.iou(torch.from_numpy(b), torch.from_numpy(b), mode="combinations").numpy()
        ss, C, sm = gn.soft_sort(torch.from_numpy(s), full_matrix=torch.from_numpy(m), temperature=temp)
        tag = f"softsort_n{n}_t{temp}"
        misc[f"{tag}/scores"] = s
        misc[f"{tag}/iou"] = m
        misc[f"{tag}/temperature"] = np.array(temp, np.float32)
        misc[f"{tag}/soft_scores"] = ss.numpy()
        misc[f"{tag}/C"] = C.numpy()
        misc[f"{tag}/soft_matrix"] = sm.numpy()
        w = rng.uniform(-1, 2, size=n).astype(np.float32)
        misc[f"{tag}/w"] = w
        for mt, kw in (("gm", dict(group_boxes=True, mask_group_boxes=True)),
                       ("gu", dict(group_boxes=True, mask_group_boxes=False)),
                       ("un", dict(group_boxes=False))):
            res = run_nms(s, m, w=w, want_iou_grad=True, sorting_method="soft", sorting_temperature=temp,
                          temperature=0.1, **kw)
            for k, v in res.items():
                misc[f"{tag}/{mt}/{k}"] = v
    # pruning_function torch + numpy branches (lib/groomed_nms.py:167-189)
    x = rng.uniform(0, 1, size=(7, 7)).astype(np.float32)
    misc["prune/x"] = x
    for method, temp in (("linear", 0.01), ("sigmoidal", 0.1), ("sigmoidal", 0.01), ("soft_nms", 0.5), ("soft_nms", 0.1)):
        misc[f"prune/{method}_{temp}/torch"] = gn.pruning_function(torch.from_numpy(x), 0.4, temp, method).numpy()
        misc[f"prune/{method}_{temp}/numpy_row0"] = np.asarray(
            gn.pruning_function(x[0].astype(np.float64), 0.4, temp, method), np.float64)
    # indices_copy (lib/groomed_nms.py:272) in the mode differentiable_nms uses (1-D indA -> g x g block)
    A = np.zeros((6, 6), np.float32)
    Bm = rng.uniform(1, 2, size=(3, 3)).astype(np.float32)
    ind = np.array([4, 0, 3], np.int64)
    misc["indices_copy/A"] = A
    misc["indices_copy/B"] = Bm
    misc["indices_copy/ind"] = ind
    misc["indices_copy/out"] = gn.indices_copy(torch.from_numpy(A.copy()), torch.from_numpy(Bm), torch.from_numpy(ind)).numpy()
    # classical NMS family on seeded dets (integer-ish pixel boxes, +1 convention)
    for tag, n, gen in (("dets40", 40, clustered_boxes_2d), ("dets300", 300, clustered_boxes_2d),
                        ("dets_uni200", 200, uniform_boxes_2d)):
        b = gen(rng, n) if gen is uniform_boxes_2d else gen(rng, n, per=8)
        dets = np.concatenate([b, tie_free_scores(rng, n)[:, None]], 1).astype(np.float32)
        misc[f"{tag}/dets"] = dets
        for thr in (0.4, 0.7):
            misc[f"{tag}/py_cpu_nms_{thr}"] = np.asarray(pcn.py_cpu_nms(dets.copy(), thr), np.int64)
            misc[f"{tag}/girshick_nms_{thr}"] = np.asarray(nms_others.girshick_nms(dets.copy(), thr, shift=1), np.int64)
            misc[f"{tag}/girshick_nms_shift0_{thr}"] = np.asarray(nms_others.girshick_nms(dets.copy(), thr, shift=0), np.int64)
        for method in (0, 1, 2):
            misc[f"{tag}/soft_nms_m{method}"] = np.asarray(
                nms_others.navneeth_soft_nms(dets.astype(np.float64).copy(), sigma=0.5, Nt=0.4, threshold=0.001,
                                             method=method, shift=1), np.int64)
    np.savez_compressed(os.path.join(OUT, "misc.npz"), **misc)

    # ------------------------------------------------------------------ after-NMS AP loss (lib/loss/aploss.py)
    spec = importlib.util.spec_from_file_location("ref_aploss", REF + "/lib/loss/aploss.py")
    apm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(apm)
    apg = {}

    def ap_case(tag, logits, targets, upstream=1.0):
        x = torch.from_numpy(logits).clone().requires_grad_(True)
        t = torch.from_numpy(targets)
        loss = apm.APLoss()(x, t)
        (loss.sum() * upstream).backward()
        apg[f"{tag}/logits"] = logits
        apg[f"{tag}/targets"] = targets
        apg[f"{tag}/loss"] = loss.detach().numpy().reshape(-1).astype(np.float32)
        apg[f"{tag}/grad"] = x.grad.numpy().astype(np.float32)
        apg[f"{tag}/upstream"] = np.array(upstream, np.float32)

    for n, npos in ((12, 3), (50, 1), (50, 10), (257, 20), (500, 37), (500, 499), (1000, 120)):
        lg = rng.uniform(0, 1, size=n).astype(np.float32)          # rescored probabilities live in [0, 1]
        tg = np.zeros(n, np.float32)
        tg[rng.choice(n, size=npos, replace=False)] = 1
        ap_case(f"u{n}_{npos}", lg, tg)
    lg = rng.normal(0, 3, size=300).astype(np.float32)             # raw logits: the clamp saturates
    tg = (rng.uniform(size=300) < 0.1).astype(np.float32)
    ap_case("wide300", lg, tg, upstream=0.05)                      # after_nms_lambda
    tg2 = tg.copy()
    tg2[rng.choice(300, size=40, replace=False)] = -1              # ignored label: neither positive nor negative
    ap_case("ignore300", lg, tg2)
    ap_case("nopos", rng.uniform(0, 1, size=40).astype(np.float32), np.zeros(40, np.float32))
    ap_case("allpos", rng.uniform(0, 1, size=16).astype(np.float32), np.ones(16, np.float32))
    tie = np.round(rng.uniform(0, 1, size=64), 1).astype(np.float32)
    ap_case("ties64", tie, (rng.uniform(size=64) < 0.3).astype(np.float32))
    # an NMS output as the loss sees it: many exact zeros
    z = rng.uniform(0, 1, size=400).astype(np.float32)
    z[rng.uniform(size=400) < 0.7] = 0
    ap_case("zeros400", z, (rng.uniform(size=400) < 0.05).astype(np.float32))
    np.savez_compressed(os.path.join(OUT, "aploss.npz"), **apg)

    # ------------------------------------------------------------------ in front of the layer
    # lib/rpn_util.py imports torchvision (via lib/augmentations.py) and the compiled lib.nms.gpu_nms at module top: stubbed
    class _Stub2(types.ModuleType):
        def __getattr__(self, key):
            if key.startswith("__"):
                raise AttributeError(key)
            return object
    for name in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional", "lib.nms.gpu_nms"):
        if name not in sys.modules:
            sys.modules[name] = _Stub2(name)
    import lib.rpn_util as rpn_util  # noqa: E402
    rng = np.random.default_rng(4242)
    pg = {}
    # (1) bbox_transform_inv (lib/rpn_util.py:872-934): 2-D and 3-D deltas, with and without means/stds (the function scales its
    #     `deltas` argument in place: clones go in)
    for tag, B, A in (("d2_64", 0, 64), ("d3_3x500", 3, 500), ("d3_1x7", 1, 7)):
        ctr = rng.uniform(0, 1760, size=(A, 2))
        wh = rng.uniform(8, 200, size=(A, 2))
        anchors = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(np.float32)
        deltas = (rng.standard_normal((max(B, 1), A, 4)) * np.array([0.3, 0.3, 0.4, 0.4])).astype(np.float32)
        if B == 0:
            deltas = deltas[0]
        means = np.array([0.01, -0.02, 0.1, 0.05], np.float32)
        stds = np.array([0.14, 0.12, 0.3, 0.25], np.float32)
        pg[f"decode/{tag}/anchors"], pg[f"decode/{tag}/deltas"] = anchors, deltas
        pg[f"decode/{tag}/means"], pg[f"decode/{tag}/stds"] = means, stds
        pg[f"decode/{tag}/out_plain"] = rpn_util.bbox_transform_inv(torch.from_numpy(anchors), torch.from_numpy(deltas.copy())).numpy()
        pg[f"decode/{tag}/out_norm"] = rpn_util.bbox_transform_inv(torch.from_numpy(anchors), torch.from_numpy(deltas.copy()),
                                                                  means=means, stds=stds).numpy()
    # (2) projected 2D boxes (lib/loss/rpn_3d.py:746-768): the reference's corner and projection functions, composed as the loss does
    p2 = np.array([[721.5377, 0.0, 609.5593, 44.85728], [0.0, 721.5377, 172.854, 0.2163791], [0.0, 0.0, 1.0, 0.002745884],
                   [0.0, 0.0, 0.0, 1.0]], np.float32)
    for tag, n in (("p64", 64), ("p500", 500)):
        par = np.stack([rng.uniform(-20, 20, n), rng.uniform(0.5, 2.5, n), rng.uniform(4, 60, n), rng.uniform(1.4, 2.0, n),
                        rng.uniform(1.3, 2.0, n), rng.uniform(3, 5, n), rng.uniform(-np.pi, np.pi, n)], 1).astype(np.float32)
        if tag == "p64":
            par[:4, 2] = [0.004, -0.003, 1.0, 2.0]       # cuboids straddling the camera plane: |z| <= 1e-2 keeps the undivided value
        t = [torch.from_numpy(par[:, i].copy()) for i in range(7)]
        corners = math_3d.get_corners_of_cuboid(x3d=t[0], y3d=t[1], z3d=t[2], w3d=t[3], h3d=t[4], l3d=t[5], ry3d=t[6])   # N x 3 x 8
        flat = corners.transpose(1, 2).reshape((-1, 3)).transpose(0, 1)
        proj = math_3d.project_3d_points_in_4D_format(torch.from_numpy(p2), flat, pad_ones=True)
        c2 = proj.transpose(0, 1).reshape((-1, 8, 4)).transpose(1, 2)
        box = torch.stack([c2[:, 0].min(1)[0], c2[:, 1].min(1)[0], c2[:, 0].max(1)[0], c2[:, 1].max(1)[0]], 1) * 0.7
        pg[f"project/{tag}/params"], pg[f"project/{tag}/p2"], pg[f"project/{tag}/scale"] = par, p2, np.float32(0.7)
        pg[f"project/{tag}/boxes"] = box.numpy()
    # (3) selection (lib/loss/rpn_3d.py:731-737): torch.sort of the foreground scores, first min(K, #fg)
    for tag, A, F, K in (("t2000_700_500", 2000, 700, 500), ("t2000_120_500", 2000, 120, 500), ("t300_300_50", 300, 300, 50)):
        sc = rng.permutation(A).astype(np.float32) / A                        # distinct: torch.sort is not stable
        fg = np.sort(rng.choice(A, size=F, replace=False)).astype(np.int64)
        st = torch.from_numpy(sc)
        _, sorted_index = torch.sort(st[torch.from_numpy(fg)], descending=True)
        num = min(K, sorted_index.shape[0])
        sel = torch.from_numpy(fg)[sorted_index[:num]]
        pg[f"topk/{tag}/scores"], pg[f"topk/{tag}/fg"], pg[f"topk/{tag}/K"] = sc, fg.astype(np.int32), np.int32(K)
        pg[f"topk/{tag}/selected"] = sel.numpy()
    # (4) best box per ground truth after the NMS (lib/loss/rpn_3d.py:801-825): the reference's overlap functions, composed as there
    for tag, n, m in (("b300_6", 300, 6), ("b500_1", 500, 1), ("b40_12", 40, 12)):
        def cuboids(k):
            return np.stack([rng.uniform(-20, 20, k), rng.uniform(0.5, 2.5, k), rng.uniform(6, 50, k), rng.uniform(1.4, 2.0, k),
                             rng.uniform(1.3, 2.0, k), rng.uniform(3, 5, k), rng.uniform(-np.pi, np.pi, k)], 1).astype(np.float32)
        gt = cuboids(m)
        pred = cuboids(n)
        near = rng.integers(0, m, size=n // 2)                                  # half of the predictions sit near a ground truth
        pred[: n // 2] = gt[near] + (rng.standard_normal((n // 2, 7)) * [0.4, 0.1, 0.6, 0.05, 0.05, 0.1, 0.1]).astype(np.float32)

        def corners(par):
            t = [torch.from_numpy(par[:, i].copy()) for i in range(7)]
            return math_3d.get_corners_of_cuboid(x3d=t[0], y3d=t[1], z3d=t[2], w3d=t[3], h3d=t[4], l3d=t[5], ry3d=t[6])

        def boxes2d(par):                                                       # any consistent 2D boxes: the projected ones
            c = corners(par)
            flat = c.transpose(1, 2).reshape((-1, 3)).transpose(0, 1)
            pr = math_3d.project_3d_points_in_4D_format(torch.from_numpy(p2), flat, pad_ones=True)
            c2 = pr.transpose(0, 1).reshape((-1, 8, 4)).transpose(1, 2)
            return torch.stack([c2[:, 0].min(1)[0], c2[:, 1].min(1)[0], c2[:, 0].max(1)[0], c2[:, 1].max(1)[0]], 1).numpy()
        pb, gb = boxes2d(pred), boxes2d(gt)
        _, iou3 = core.iou3d_approximate(corners(pred).clone(), corners(gt).clone(), mode="combinations", method="generalized")
        iou2 = core.iou(torch.from_numpy(pb), torch.from_numpy(gb), mode="combinations")
        scores_with_gt = 0.5 * (1 + iou3) * iou2
        _, max_idx = torch.max(scores_with_gt, dim=0)
        beta = 0.3
        sel = max_idx[scores_with_gt.gather(0, max_idx.unsqueeze(0)).squeeze(0) > beta].flatten()
        tg = np.zeros(n, np.float32)
        tg[sel.numpy()] = 1
        pg[f"best/{tag}/pred_params"], pg[f"best/{tag}/pred_boxes"] = pred, pb
        pg[f"best/{tag}/gt_params"], pg[f"best/{tag}/gt_boxes"] = gt, gb
        pg[f"best/{tag}/beta"] = np.float32(beta)
        pg[f"best/{tag}/scores_with_gt"] = scores_with_gt.numpy()
        pg[f"best/{tag}/max_indices"] = max_idx.numpy()
        pg[f"best/{tag}/targets"] = tg
    np.savez_compressed(os.path.join(OUT, "proposals.npz"), **pg)

    # ------------------------------------------------------------------ KITTI result writer / evaluation hand-off
    # lib/rpn_util.py:1489-1545 convert_image_predictions_to_correct_entries, :1571-1631 get_text_to_write_in_kitti_format,
    # :2013-2040 parse_kitti_result.  The text is stored as bytes; the devkit's stats-file format is exercised on a synthetic file.
    import tempfile
    rng = np.random.default_rng(777)
    kg = {}

    class Conf(dict):
        __getattr__ = dict.__getitem__
    p2d = p2.astype(np.float64)
    for tag, n, has_un in (("k12", 12, False), ("k40_un", 40, True), ("k0", 0, False)):
        ctr = np.stack([rng.uniform(100, 1600, n), rng.uniform(100, 400, n)], 1)
        wh = rng.uniform(20, 200, size=(n, 2))
        bx = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1)
        boxes = np.concatenate([bx, rng.uniform(0.05, 1, (n, 1)), rng.integers(1, 4, (n, 1)).astype(np.float64),
                                ctr + rng.normal(0, 5, (n, 2)), rng.uniform(4, 70, (n, 1)),                 # projected 3D centre + depth
                                rng.uniform(1.4, 2.0, (n, 1)), rng.uniform(1.3, 2.0, (n, 1)), rng.uniform(3, 5, (n, 1)),
                                rng.uniform(-3 * np.pi, 3 * np.pi, (n, 1)), rng.uniform(0.2, 1, (n, 1))], 1)   # alpha (un-wrapped), un
        conf = Conf(lbls=["Car", "Pedestrian", "Cyclist"], has_un=has_un, use_un_for_score=has_un)
        kg[f"{tag}/boxes"] = boxes
        kg[f"{tag}/p2"] = p2d
        kg[f"{tag}/has_un"] = np.array(has_un)
        if n > 0:
            conv = rpn_util.convert_image_predictions_to_correct_entries(boxes.copy(), conf, p2d)
        else:
            conv = np.zeros((0, 17))
        kg[f"{tag}/converted"] = conv
        text = rpn_util.get_text_to_write_in_kitti_format(conv, conf)
        kg[f"{tag}/text"] = np.frombuffer(text.encode(), dtype=np.uint8)
    stats = rng.uniform(0, 1, size=(3, 41))
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        for row in stats:
            f.write(" ".join("%.6f" % v for v in row) + "\n")
        stats_path = f.name
    kg["stats/text"] = np.frombuffer(open(stats_path).read().encode(), dtype=np.uint8)
    kg["stats/r11"] = np.array(rpn_util.parse_kitti_result(stats_path, use_40=False))
    kg["stats/r40"] = np.array(rpn_util.parse_kitti_result(stats_path, use_40=True))
    os.unlink(stats_path)
    np.savez_compressed(os.path.join(OUT, "kitti_io.npz"), **kg)

    make_f64_call_site()

    for f in ("nms_small.npz", "boxes_2d.npz", "boxes_3d.npz", "misc.npz", "aploss.npz", "proposals.npz", "kitti_io.npz", "f64site.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "f64site":      # only the float64 call-site fixture (independent RNG stream)
        torch.manual_seed(0)
        torch.set_num_threads(4)
        make_f64_call_site()
        print("f64site.npz", os.path.getsize(os.path.join(OUT, "f64site.npz")) // 1024, "KiB")
    else:
        main()
