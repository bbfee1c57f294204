"""Pins the CPU oracle (oracle/) against the golden vectors captured from the reference
(tests/golden/make_golden.py) and the reference's two known-answer tests.  CPU only."""
import numpy as np
import pytest

from conftest import MODES, TOL, check_index_lists
from oracle import oracle as O
from oracle import nms_others_oracle as NO


def _all_cases(g):
    out = []
    for case in g.cases():
        if not g.has(f"{case}/scores"):
            continue
        for mode in g.modes(case):
            if mode in MODES:
                out.append((case, mode))
    return out


def _run_case(g, case, mode):
    s, m, w = g[f"{case}/scores"], g[f"{case}/iou"], g[f"{case}/w"]
    want_gi = g.has(f"{case}/{mode}/grad_iou")
    res = O.differentiable_nms(s, m, grad_prob=w if len(s) else None, want_grad_iou=want_gi, **MODES[mode])
    ref_prob = g[f"{case}/{mode}/prob"]
    np.testing.assert_allclose(res["prob"], ref_prob, atol=TOL, rtol=0, equal_nan=True)
    check_index_lists(res["valid"], res["invalid"], g[f"{case}/{mode}/valid"], g[f"{case}/{mode}/invalid"])
    if g.has(f"{case}/{mode}/grad_scores"):
        np.testing.assert_allclose(res["grad_scores"], g[f"{case}/{mode}/grad_scores"], atol=2e-4, rtol=1e-4)
    if want_gi:
        np.testing.assert_allclose(res["grad_iou"], g[f"{case}/{mode}/grad_iou"], atol=2e-4, rtol=1e-4)


def test_known_answer_vectors(golden_nms):
    """test/test_differentiable_nms_forward.py:127-140 expected outputs (3 decimals printed there)."""
    for case in ("kat1", "kat2"):
        res = O.differentiable_nms(golden_nms[f"{case}/scores"], golden_nms[f"{case}/iou"], temperature=0.1)
        np.testing.assert_allclose(res["prob"], golden_nms[f"{case}/expected_prob"], atol=5e-4)
    res = O.differentiable_nms(golden_nms["kat1/scores"], golden_nms["kat1/iou"])
    assert list(res["valid"]) == [0, 1, 3] and list(res["invalid"]) == [2]
    res = O.differentiable_nms(golden_nms["kat2/scores"], golden_nms["kat2/iou"])
    assert list(res["valid"]) == [0, 1, 4] and sorted(res["invalid"]) == [2, 3]


def test_nms_small_all_modes(golden_nms):
    cases = _all_cases(golden_nms)
    assert len(cases) > 100
    for case, mode in cases:
        try:
            _run_case(golden_nms, case, mode)
        except AssertionError as e:
            raise AssertionError(f"{case}/{mode}: {e}") from e


def test_nms_box_derived(golden_box2d, golden_box3d):
    n = 0
    for g in (golden_box2d, golden_box3d):
        for case, mode in _all_cases(g):
            try:
                _run_case(g, case, mode)
            except AssertionError as e:
                raise AssertionError(f"{case}/{mode}: {e}") from e
            n += 1
    assert n > 50


def test_get_groups(golden_nms, golden_box2d):
    for g in (golden_nms, golden_box2d):
        for case in g.cases():
            for gs in (100, 2):
                if not g.has(f"{case}/groups_gs{gs}/lens"):
                    continue
                s, m = g[f"{case}/scores"], g[f"{case}/iou"]
                order = O.argsort_desc(s)
                groups = O.get_groups(m[order][:, order], 0.4, s[order], group_size=gs)
                lens = g[f"{case}/groups_gs{gs}/lens"]
                flat = g[f"{case}/groups_gs{gs}/flat"]
                assert [len(x) for x in groups] == list(lens), case
                assert list(np.concatenate(groups) if groups else []) == list(flat), case


def test_iou2d_bit_exact(golden_box2d):
    for case in ("uni64", "clu64", "uni256", "clu256", "clu250"):
        b = golden_box2d[f"{case}/boxes"]
        got = O.iou2d(b, b)
        ref = golden_box2d[f"{case}/iou"]
        assert np.array_equal(got, ref, equal_nan=True), f"{case}: max diff {np.nanmax(np.abs(got - ref))}"
    got = O.iou2d(golden_box2d["rect/a"], golden_box2d["rect/b"])
    assert np.array_equal(got, golden_box2d["rect/iou"])
    got = O.iou2d(golden_box2d["zero_area/boxes"], golden_box2d["zero_area/boxes"])
    assert np.array_equal(got, golden_box2d["zero_area/iou"], equal_nan=True)
    assert np.isnan(got[2, 2])


def test_corners_and_iou3d(golden_box3d):
    g = golden_box3d
    for case in ("m5", "uni64", "clu64", "clu200"):
        c = O.corners_of_cuboid(g[f"{case}/params"])
        np.testing.assert_allclose(c, g[f"{case}/corners"], atol=2e-5, rtol=1e-6)
        # overlaps from the REFERENCE's corners: isolates the overlap arithmetic from sin/cos rounding
        for method in ("normal", "generalized"):
            bev, i3 = O.iou3d_approximate(g[f"{case}/corners"], g[f"{case}/corners"], generalized=(method == "generalized"))
            np.testing.assert_allclose(bev, g[f"{case}/{method}/iou_bev"], atol=1e-6, rtol=1e-6)
            np.testing.assert_allclose(i3, g[f"{case}/{method}/iou_3d"], atol=1e-6, rtol=1e-6)
        _, gi = O.iou3d_approximate(c, c, generalized=True)
        np.testing.assert_allclose(0.5 * (1 + gi), g[f"{case}/nms_overlap"], atol=TOL)
    bev, i3 = O.iou3d_approximate(g["rect/corners_a"], g["rect/corners_b"], generalized=True)
    np.testing.assert_allclose(bev, g["rect/iou_bev"], atol=1e-6)
    np.testing.assert_allclose(i3, g["rect/iou_3d"], atol=1e-6)


def test_pruning_function(golden_misc):
    g = golden_misc
    for method, temp in (("linear", 0.01), ("sigmoidal", 0.1), ("sigmoidal", 0.01), ("soft_nms", 0.5), ("soft_nms", 0.1)):
        got = O.pruning_function(g["prune/x"], 0.4, temp, method)
        np.testing.assert_allclose(got, g[f"prune/{method}_{temp}/torch"], atol=1e-6)
    with pytest.raises(NotImplementedError):
        O.pruning_function(g["prune/x"], 0.4, 0.1, "bogus")


def test_soft_sort_forward_backward(golden_misc):
    g = golden_misc
    tags = sorted({k.split("/")[0] for k in g.keys if k.startswith("softsort_")})
    assert len(tags) == 3
    for tag in tags:
        s, m, t = g[f"{tag}/scores"], g[f"{tag}/iou"], float(g[f"{tag}/temperature"])
        ss, C, sm = O.soft_sort(s, m, t)
        np.testing.assert_allclose(C, g[f"{tag}/C"], atol=1e-5)
        np.testing.assert_allclose(ss, g[f"{tag}/soft_scores"], atol=1e-5)
        np.testing.assert_allclose(sm, g[f"{tag}/soft_matrix"], atol=1e-5)
        for mt, kw in (("gm", dict(group_boxes=True, mask_group_boxes=True)),
                       ("gu", dict(group_boxes=True, mask_group_boxes=False)), ("un", dict(group_boxes=False))):
            res = O.differentiable_nms(s, m, sorting_method="soft", sorting_temperature=t, temperature=0.1,
                                       grad_prob=g[f"{tag}/w"], want_grad_iou=True, **kw)
            np.testing.assert_allclose(res["prob"], g[f"{tag}/{mt}/prob"], atol=TOL, err_msg=f"{tag}/{mt}")
            check_index_lists(res["valid"], res["invalid"], g[f"{tag}/{mt}/valid"], g[f"{tag}/{mt}/invalid"])
            np.testing.assert_allclose(res["grad_scores"], g[f"{tag}/{mt}/grad_scores"], atol=5e-3, rtol=2e-3,
                                       err_msg=f"{tag}/{mt}")
            np.testing.assert_allclose(res["grad_iou"], g[f"{tag}/{mt}/grad_iou"], atol=2e-4, rtol=1e-3,
                                       err_msg=f"{tag}/{mt}")


def test_classic_nms_family(golden_misc):
    g = golden_misc
    for tag in ("dets40", "dets300", "dets_uni200"):
        dets = g[f"{tag}/dets"]
        for thr in (0.4, 0.7):
            ref = list(g[f"{tag}/py_cpu_nms_{thr}"])
            assert O.classic_nms(dets, thr, rule="py") == ref
            # the GPU rule (strict >) differs from py_cpu_nms only at IoU == thresh exactly or NaN
            assert O.classic_nms(dets, thr, rule="gpu") == ref
            assert list(NO.girshick_nms(dets, thr, shift=1)) == list(g[f"{tag}/girshick_nms_{thr}"])
            assert list(NO.girshick_nms(dets, thr, shift=0)) == list(g[f"{tag}/girshick_nms_shift0_{thr}"])
        for method in (0, 1, 2):
            assert list(NO.soft_nms(dets, method=method)) == list(g[f"{tag}/soft_nms_m{method}"]), (tag, method)


def test_aploss_oracle_against_reference():
    """the after-NMS AP loss (lib/loss/aploss.py) -- oracle vs vectors captured from the reference."""
    from conftest import Golden
    g = Golden("aploss.npz")
    cases = g.cases()
    assert len(cases) >= 13
    for tag in cases:
        loss, grad = O.aploss(g[f"{tag}/logits"], g[f"{tag}/targets"])
        up = float(g[f"{tag}/upstream"])
        np.testing.assert_allclose(loss, g[f"{tag}/loss"][0], atol=1e-5, err_msg=tag)
        np.testing.assert_allclose(grad * up, g[f"{tag}/grad"], atol=1e-6, rtol=1e-4, err_msg=tag)
    loss, grad = O.aploss(g["nopos/logits"], g["nopos/targets"])
    assert loss == 0.0 and not grad.any()


def test_proposals_oracle_against_reference():
    """decode, projected boxes and the score top-K in front of the layer -- oracle vs vectors captured from
    lib/rpn_util.py::bbox_transform_inv, lib/math_3d.py::get_corners_of_cuboid + project_3d_points_in_4D_format, torch.sort."""
    from conftest import Golden
    import oracle.proposals_oracle as PO
    g = Golden("proposals.npz")
    for tag in ("d2_64", "d3_3x500", "d3_1x7"):
        a, d = g[f"decode/{tag}/anchors"], g[f"decode/{tag}/deltas"]
        keep = d.copy()
        np.testing.assert_allclose(PO.bbox_transform_inv(a, d), g[f"decode/{tag}/out_plain"], rtol=2e-6, atol=2e-4)
        np.testing.assert_allclose(PO.bbox_transform_inv(a, d, g[f"decode/{tag}/means"], g[f"decode/{tag}/stds"]),
                                   g[f"decode/{tag}/out_norm"], rtol=2e-6, atol=2e-4)
        assert np.array_equal(d, keep)
    for tag in ("p64", "p500"):
        got = PO.projected_boxes_2d(g[f"project/{tag}/params"], g[f"project/{tag}/p2"], float(g[f"project/{tag}/scale"]))
        np.testing.assert_allclose(got, g[f"project/{tag}/boxes"], rtol=1e-4, atol=5e-3)
    for tag in ("t2000_700_500", "t2000_120_500", "t300_300_50"):
        sel = PO.select_topk(g[f"topk/{tag}/scores"], g[f"topk/{tag}/fg"], int(g[f"topk/{tag}/K"]))
        assert np.array_equal(sel, g[f"topk/{tag}/selected"]), tag


def test_best_targets_oracle_against_reference():
    """best box per ground truth after the NMS (lib/loss/rpn_3d.py:801-825) -- oracle vs the reference's
    iou3d_approximate x iou composition."""
    from conftest import Golden
    import oracle.proposals_oracle as PO
    g = Golden("proposals.npz")
    for tag in ("b300_6", "b500_1", "b40_12"):
        tg, best, bscore = PO.best_targets(g[f"best/{tag}/pred_params"], g[f"best/{tag}/pred_boxes"], g[f"best/{tag}/gt_params"],
                                           g[f"best/{tag}/gt_boxes"], float(g[f"best/{tag}/beta"]))
        ref = g[f"best/{tag}/scores_with_gt"]
        np.testing.assert_allclose(bscore, ref.max(0), atol=1e-4)
        # the argmax may differ only where two predictions score within the tolerance of each other
        for j, i in enumerate(g[f"best/{tag}/max_indices"]):
            k = int(np.argmax(ref[:, j]))
            assert ref[k, j] - ref[int(i), j] <= 1e-6
        assert np.array_equal(tg, g[f"best/{tag}/targets"]), tag


def test_oracle_float64_call_site(golden_f64site):
    """The inference call site's float64 NumPy branches (lib/rpn_util.py:1292-1320): the oracle's float64 IoU restatement reproduces
    the reference's matrix bit for bit (SHA-256 of the bytes) on all 200 cases, and the layer on its fp32 rounding (lib/groomed_nms.py:36)
    returns the reference's keep lists and probabilities; the corner restatement likewise on the 3D cases (same host, same libm)."""
    import hashlib
    from conftest import f64site_aboxes
    from oracle import oracle as O
    g = golden_f64site
    off = g["d2/valid_off"]
    for c in range(g["d2/boxes32"].shape[0]):
        ab = f64site_aboxes(g, c)
        m = O.iou2d_f64(ab[:, :4], ab[:, :4])
        assert m.dtype == np.float64 and hashlib.sha256(np.ascontiguousarray(m).tobytes()).digest() == g["d2/iou_sha256"][c].tobytes(), c
        if c % 4 == 0:
            r = O.differentiable_nms(ab[:, 4].astype(np.float32), m.astype(np.float32))
            assert sorted(r["valid"].tolist()) == sorted(g["d2/valid"][off[c]:off[c + 1]].tolist()), c
            assert np.array_equal(r["prob"], g["d2/prob"][c]), c
    off = g["d3/valid_off"]
    for c in range(0, g["d3/params32"].shape[0], 3):
        c32 = O.corners_of_cuboid_numpy_branch(g["d3/params32"][c]).astype(np.float32)
        if g.has(f"d3/corners32_{c}"):
            assert np.abs(c32 - g[f"d3/corners32_{c}"]).max() <= 1e-5
        _, i3 = O.iou3d_approximate(c32, c32, generalized=True)
        ious = (np.float32(0.5) * (np.float32(1.0) + i3)).astype(np.float32)
        r = O.differentiable_nms(g["d3/scores32"][c], ious)
        # (np.cos on float32 is a SIMD kernel whose last bit depends on the host CPU: sets may differ where a corner does)
        if hashlib.sha256(np.ascontiguousarray(c32).tobytes()).digest() == g["d3/corners32_sha256"][c].tobytes():
            assert sorted(r["valid"].tolist()) == sorted(g["d3/valid"][off[c]:off[c + 1]].tolist()), c
            assert np.abs(r["prob"] - g["d3/prob"][c]).max() <= 1e-4, c
