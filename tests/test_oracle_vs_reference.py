"""The oracle against the IMPORTED reference on fresh inputs (build container only).

`tests/golden/*.npz` pin the oracle to vectors the reference produced once; this test draws NEW inputs every time the suite runs here and runs
the reference itself (lib/groomed_nms.py:10 differentiable_nms with autograd, lib/core.py:480 iou) beside the oracle -- sizes, generators,
modes and pruning methods the fixtures do not hold.  /root/reference exists only in the build container: everywhere else (the GPU box) the
test is skipped, and nothing here is marked `gpu`.  Tolerances: the overlap matrix bit for bit; probabilities and gradients 2e-6 (the
reference multiplies an N x N inversion matrix where the oracle evaluates the closed form: different summation order); valid / invalid
index SETS equal (the order among exactly tied probabilities is implementation-defined in torch.sort)."""
import importlib.util
import os
import sys

import numpy as np
import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "lib")), reason="the reference is only present in the build container")


@pytest.fixture(scope="module")
def ref():
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_golden_for_import", os.path.join(here, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    sys.dont_write_bytecode = True
    spec.loader.exec_module(mg)                 # (its module body imports the reference out of tree; generation only runs under __main__)
    return mg


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    oracle.build()
    return oracle


CASES = [   # (seed, N, generator, kwargs)
    (101, 37, "uniform", {}),
    (102, 129, "clustered", {}),
    (103, 300, "clustered", dict(nms_threshold=0.55, group_size=3)),
    (104, 200, "uniform", dict(pruning_method="sigmoidal", temperature=0.1)),
    (105, 160, "clustered", dict(pruning_method="soft_nms", temperature=0.3, valid_box_prob_threshold=0.1)),
    (106, 150, "clustered", dict(mask_group_boxes=False)),
    (107, 120, "clustered", dict(group_boxes=False)),
    (108, 257, "clustered", dict(return_sorted_prob=True, nms_threshold=0.3)),
    (109, 2, "uniform", {}),
    (110, 1, "uniform", {}),
]


@pytest.mark.parametrize("seed,N,kind,kw", CASES)
def test_oracle_equals_the_reference_on_fresh_inputs(ref, O, seed, N, kind, kw):
    import torch
    rng = np.random.default_rng(seed + int.from_bytes(os.urandom(2), "little"))      # fresh inputs on every run
    boxes = ref.uniform_boxes_2d(rng, N) if kind == "uniform" else ref.clustered_boxes_2d(rng, N, per=12)
    scores = ref.tie_free_scores(rng, N)
    w = rng.uniform(-1.0, 2.0, size=N).astype(np.float32)
    bt = torch.from_numpy(boxes)
    iou_ref = ref.core.iou(bt, bt).float()
    iou_orc = O.iou2d(boxes, boxes)
    assert np.array_equal(iou_ref.numpy(), iou_orc), "overlap matrix differs from lib/core.py iou"
    st = torch.from_numpy(scores).clone().requires_grad_(True)
    valid, invalid, prob = ref.gn.differentiable_nms(st, iou_ref, **kw)
    (prob * torch.from_numpy(w)).sum().backward()
    got = O.differentiable_nms(scores, iou_orc, grad_prob=w, **kw)
    np.testing.assert_allclose(got["prob"], prob.detach().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(got["grad_scores"], st.grad.numpy(), rtol=0, atol=2e-6)
    assert set(int(i) for i in got["valid"]) == set(int(i) for i in np.asarray(valid)), "valid_boxes_index"
    assert set(int(i) for i in got["invalid"]) == set(int(i) for i in np.asarray(invalid)), "invalid_boxes_index"


@pytest.mark.parametrize("seed,N,clustered", [(201, 23, False), (202, 96, True), (203, 170, True)])
def test_oracle_3d_overlaps_equal_the_reference_on_fresh_inputs(ref, O, seed, N, clustered):
    """lib/math_3d.py get_corners_of_cuboid + lib/core.py:352 iou3d_approximate, both methods, and what the callers feed the
    layer (0.5 * (1 + giou), lib/loss/rpn_3d.py:781).  Corners 2e-5 (sin/cos rounding differs between libm and torch); the
    overlap arithmetic from the REFERENCE's corners 1e-6 -- the same split tests/test_oracle_golden.py makes."""
    import torch
    rng = np.random.default_rng(seed + int.from_bytes(os.urandom(2), "little"))
    p = ref.boxes_3d(rng, N, clustered=clustered)
    t = [torch.from_numpy(np.ascontiguousarray(p[:, i])) for i in range(7)]
    corners = ref.math_3d.get_corners_of_cuboid(*t)
    cn = corners.numpy().astype(np.float32)
    np.testing.assert_allclose(O.corners_of_cuboid(p), cn, atol=2e-5, rtol=1e-6)
    for method in ("normal", "generalized"):
        # iou3d_approximate mutates its inputs (lib/core.py:379-380): clones
        bev, i3 = ref.core.iou3d_approximate(corners.clone(), corners.clone(), mode="combinations", method=method)
        ob, o3 = O.iou3d_approximate(cn, cn, generalized=(method == "generalized"))
        np.testing.assert_allclose(ob, bev.numpy(), atol=1e-6, rtol=1e-6)
        np.testing.assert_allclose(o3, i3.numpy(), atol=1e-6, rtol=1e-6)
    # rectangular call: a against b
    q = ref.boxes_3d(rng, 11, clustered=True)
    cb = ref.math_3d.get_corners_of_cuboid(*[torch.from_numpy(np.ascontiguousarray(q[:, i])) for i in range(7)])
    bev, i3 = ref.core.iou3d_approximate(corners.clone(), cb.clone(), mode="combinations", method="generalized")
    ob, o3 = O.iou3d_approximate(cn, cb.numpy().astype(np.float32), generalized=True)
    np.testing.assert_allclose(ob, bev.numpy(), atol=1e-6, rtol=1e-6)
    np.testing.assert_allclose(o3, i3.numpy(), atol=1e-6, rtol=1e-6)


@pytest.mark.parametrize("seed,N,kind", [(301, 1, "uniform"), (302, 75, "clustered"), (303, 400, "clustered"), (304, 250, "uniform")])
def test_oracle_classic_nms_equals_the_reference_on_fresh_inputs(ref, O, seed, N, kind):
    """lib/nms/py_cpu_nms.py:10 and lib/nms_others.py:119 girshick_nms on fresh detections: kept index LISTS equal."""
    from oracle import nms_others_oracle as NO
    rng = np.random.default_rng(seed + int.from_bytes(os.urandom(2), "little"))
    b = ref.uniform_boxes_2d(rng, N) if kind == "uniform" else ref.clustered_boxes_2d(rng, N, per=8)
    dets = np.concatenate([b, ref.tie_free_scores(rng, N)[:, None]], 1).astype(np.float32)
    for thr in (0.3, 0.5, 0.75):
        want = [int(i) for i in ref.pcn.py_cpu_nms(dets.copy(), thr)]
        assert O.classic_nms(dets, thr, rule="py") == want
        assert O.classic_nms(dets, thr, rule="gpu") == want      # '>' and 'not <=' differ only at equality / NaN
        for shift in (0, 1):
            assert list(NO.girshick_nms(dets, thr, shift=shift)) == \
                [int(i) for i in ref.nms_others.girshick_nms(dets.copy(), thr, shift=shift)]


@pytest.mark.parametrize("seed,N,temp", [(401, 12, 0.01), (402, 33, 0.002), (403, 64, 0.001)])
def test_oracle_soft_sort_equals_the_reference_on_fresh_inputs(ref, O, seed, N, temp):
    """lib/groomed_nms.py:131 soft_sort on score-sorted inputs (the only inputs on which the reference's layer terminates with
    soft sorting, see tests/golden/make_golden.py): permutation matrix, soft scores and the row-mixed overlap matrix, 1e-5."""
    import torch
    rng = np.random.default_rng(seed + int.from_bytes(os.urandom(2), "little"))
    s = np.sort(ref.tie_free_scores(rng, N))[::-1].copy()
    b = ref.clustered_boxes_2d(rng, N, per=4)
    m = ref.core.iou(torch.from_numpy(b), torch.from_numpy(b), mode="combinations").numpy()
    ss, C, sm = ref.gn.soft_sort(torch.from_numpy(s), full_matrix=torch.from_numpy(m), temperature=temp)
    oss, oC, osm = O.soft_sort(s, m, temp)
    np.testing.assert_allclose(oC, C.numpy(), atol=1e-5)
    np.testing.assert_allclose(oss, ss.numpy(), atol=1e-5)
    np.testing.assert_allclose(osm, sm.numpy(), atol=1e-5)
