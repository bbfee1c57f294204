"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol the header declares, argument
validation fails loudly before touching the device, host-side helpers match the reference's goldens, and the
product package never reaches into oracle/."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from groomed_nms_amd import build, _lib
    build.build()
    return _lib.load()


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "groomed_nms_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", text)
    return sorted(set(n for n in names if n.startswith("gnms_") or n == "_nms"))


def test_library_exports_every_declared_symbol(lib):
    from groomed_nms_amd import _lib
    declared = _declared_functions()
    assert "_nms" in declared and "gnms_forward" in declared and len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), "libgroomed_nms_hip.so does not export %s" % name
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared, "ctypes signatures and header disagree"
    assert lib.gnms_abi_version() == 1


def test_default_params_are_the_reference_defaults(lib):
    """lib/groomed_nms.py:10: nms_threshold=0.4, temperature=0.01, valid_box_prob_threshold=0.3, linear, hard,
    group_boxes, mask_group_boxes, group_size=100."""
    from groomed_nms_amd._lib import GnmsParams
    p = GnmsParams()
    lib.gnms_default_params(ctypes.byref(p))
    assert abs(p.nms_threshold - 0.4) < 1e-7 and abs(p.temperature - 0.01) < 1e-9 and abs(p.valid_box_prob_threshold - 0.3) < 1e-7
    assert (p.pruning_method, p.return_sorted_prob, p.group_boxes, p.mask_group_boxes, p.group_size, p.presorted) == (0, 0, 1, 1, 100, 0)
    import inspect
    from groomed_nms_amd import differentiable_nms
    sig = inspect.signature(differentiable_nms)
    assert list(sig.parameters) == ["scores_unsorted", "iou_unsorted", "nms_threshold", "pruning_method", "temperature",
                                    "valid_box_prob_threshold", "return_sorted_prob", "sorting_method", "sorting_temperature",
                                    "group_boxes", "mask_group_boxes", "group_size", "debug"]
    d = {k: v.default for k, v in sig.parameters.items()}
    assert (d["nms_threshold"], d["pruning_method"], d["temperature"], d["valid_box_prob_threshold"], d["return_sorted_prob"],
            d["sorting_method"], d["sorting_temperature"], d["group_boxes"], d["mask_group_boxes"], d["group_size"], d["debug"]) == \
        (0.4, "linear", 0.01, 0.3, False, "hard", None, True, True, 100, False)


def test_argument_validation_without_gpu(lib):
    """Every check below returns before any HIP call, so it runs on the CPU-only build container."""
    from groomed_nms_amd._lib import GnmsParams
    p = GnmsParams()
    lib.gnms_default_params(ctypes.byref(p))
    fake = ctypes.c_void_p(256 * 1024)          # never dereferenced on the host
    assert lib.gnms_workspace_bytes(0, 4096, ctypes.byref(p)) == 0
    w1 = lib.gnms_workspace_bytes(1, 4096, ctypes.byref(p))
    assert lib.gnms_workspace_bytes(8, 4096, ctypes.byref(p)) == 8 * w1 and w1 >= 4096 * 4096 // 8 and w1 % 256 == 0
    # N beyond the supported maximum
    rc = lib.gnms_forward(fake, fake, 1, 16385, 16385, None, ctypes.byref(p), fake, None, None, None, None, None, fake, 1 << 40, None)
    assert rc == -2 and b"GNMS_MAX_BOXES" in lib.gnms_last_error()
    # unknown pruning method == the reference's NotImplementedError (lib/groomed_nms.py:177-178)
    p.pruning_method = 7
    rc = lib.gnms_forward(fake, fake, 1, 64, 64, None, ctypes.byref(p), fake, None, None, None, None, None, fake, 1 << 30, None)
    assert rc == -2 and b"not implemented" in lib.gnms_last_error()
    p.pruning_method = 0
    # ld < N, short workspace, missing workspace, misaligned workspace
    assert lib.gnms_forward(fake, fake, 1, 64, 32, None, ctypes.byref(p), fake, None, None, None, None, None, fake, 1 << 30, None) == -1
    assert lib.gnms_forward(fake, fake, 1, 64, 64, None, ctypes.byref(p), fake, None, None, None, None, None, fake, 16, None) == -4
    assert lib.gnms_forward(fake, fake, 1, 64, 64, None, ctypes.byref(p), fake, None, None, None, None, None, None, 1 << 30, None) == -1
    assert lib.gnms_forward(fake, fake, 1, 64, 64, None, ctypes.byref(p), fake, None, None, None, None, None, ctypes.c_void_p(8), 1 << 30, None) == -1
    assert lib.gnms_backward(fake, fake, fake, 1, 64, 32, None, ctypes.byref(p), fake, None, fake, 1 << 30, None) == -1
    assert lib.gnms_iou2d(fake, fake, 1, 4, 8, fake, 4, None) == -1           # ld < N
    assert lib.gnms_iou2d(None, fake, 1, 4, 8, fake, 8, None) == -1           # null boxes
    assert lib.gnms_iou3d_approximate(fake, fake, 1, 4, 4, 9, None, fake, 4, None) == -1
    assert lib.gnms_pruning_function(fake, 10, 0.4, 0.1, 5, fake, None) == -2
    assert lib.gnms_nms_sorted(fake, 5, 3, 0.5, fake, fake, fake, 1 << 20, None) == -1   # boxes_dim < 4
    assert lib.gnms_nms_workspace_bytes(0) == 0 and lib.gnms_nms_workspace_bytes(1000) > 0
    from groomed_nms_amd import pruning_function
    with pytest.raises(NotImplementedError):
        pruning_function(torch.rand(3, 3), pruning_method="bogus")


def test_product_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import groomed_nms_amd as G
    from groomed_nms_amd._lib import GnmsError
    with pytest.raises(GnmsError):
        G.differentiable_nms(torch.rand(4), torch.eye(4))
    with pytest.raises(GnmsError):
        G.differentiable_nms(np.random.rand(4), np.eye(4))


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under groomed_nms_amd/ may import, link or call it."""
    pkg = os.path.join(ROOT, "groomed_nms_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".c")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "libgnms_oracle" not in text and "gnms_oracle_" not in text, f
    out = os.popen("ldd %s 2>/dev/null" % os.path.join(pkg, "libgroomed_nms_hip.so")).read()
    assert "oracle" not in out


def test_host_helpers_against_goldens(golden_misc):
    from groomed_nms_amd.nms import cpu_nms, py_cpu_nms
    from groomed_nms_amd import indices_copy, sigmoid_numpy, pruning_function, cast_to_cpu_cuda_tensor
    g = golden_misc
    for tag in ("dets40", "dets300", "dets_uni200"):
        d = g[f"{tag}/dets"]
        for thr in (0.4, 0.7):
            assert [int(i) for i in py_cpu_nms(d, thr)] == list(g[f"{tag}/py_cpu_nms_{thr}"])
            assert cpu_nms(d, thr) == list(g[f"{tag}/py_cpu_nms_{thr}"])
    out = indices_copy(torch.from_numpy(g["indices_copy/A"].copy()), torch.from_numpy(g["indices_copy/B"]), torch.from_numpy(g["indices_copy/ind"]))
    assert np.array_equal(out.numpy(), g["indices_copy/out"])
    x = g["prune/x"][0].astype(np.float64)
    for method, temp in (("linear", 0.01), ("sigmoidal", 0.1), ("soft_nms", 0.5)):
        np.testing.assert_allclose(pruning_function(x, 0.4, temp, method), g[f"prune/{method}_{temp}/numpy_row0"], atol=1e-12)
    assert np.allclose(sigmoid_numpy(np.array([-1.0, 0.0, 2.0])), 1 / (1 + np.exp(-np.array([-1.0, 0.0, 2.0]))))
    a, b = torch.zeros(2), torch.ones(3)
    assert cast_to_cpu_cuda_tensor(a, b) is a


def test_synthetic_generators_are_deterministic():
    from groomed_nms_amd import synthetic
    b1, s1 = synthetic.batch_2d(3, 2, 300, "clustered")
    b2, s2 = synthetic.batch_2d(3, 2, 300, "clustered")
    assert np.array_equal(b1, b2) and np.array_equal(s1, s2)
    assert b1.dtype == np.float32 and s1.dtype == np.float32
    assert all(len(np.unique(s1[i])) == 300 for i in range(2))              # tie-free by construction
    assert np.all(b1[..., 2] > b1[..., 0]) and np.all(b1[..., 3] > b1[..., 1])
    p, _ = synthetic.batch_3d(4, 1, 128)
    assert p.shape == (1, 128, 7)


def test_bench_gpus_flag_launches_or_refuses():
    """bench.py --gpus N is a launcher of its own: without N visible GPUs it exits non-zero with a message (here: no GPU at all),
    and a launcher-provided WORLD_SIZE that disagrees with --gpus is an error too -- it never silently times fewer GPUs."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_kitti_result_writer_against_reference_vectors(tmp_path):
    """the inference tail behind the NMS (lib/rpn_util.py:1489-1631, :2013-2076) -- converted entries, the 16-column
    result text byte for byte, the file writer, the devkit stats parser and the subprocess hand-off (a stand-in evaluator script)."""
    from conftest import Golden
    from groomed_nms_amd import kitti_io as K
    g = Golden("kitti_io.npz")

    class Conf(dict):
        __getattr__ = dict.__getitem__
    for tag in ("k12", "k40_un", "k0"):
        conf = Conf(lbls=["Car", "Pedestrian", "Cyclist"], has_un=bool(g[f"{tag}/has_un"]), use_un_for_score=bool(g[f"{tag}/has_un"]))
        boxes = g[f"{tag}/boxes"]
        keep = boxes.copy()
        conv = K.convert_image_predictions_to_correct_entries(boxes, conf, g[f"{tag}/p2"]) if len(boxes) else np.zeros((0, 17))
        assert np.array_equal(boxes, keep)                                                  # input untouched (deepcopy, :1490)
        np.testing.assert_allclose(conv, g[f"{tag}/converted"], rtol=1e-12, atol=1e-12)
        text = K.get_text_to_write_in_kitti_format(conv, conf)
        assert text.encode() == g[f"{tag}/text"].tobytes(), tag
        assert K.write_image_boxes_to_txt_file(conv, {"lbls": conf.lbls}, str(tmp_path), "000042") == text
        assert open(tmp_path / "000042.txt").read() == text
    stats = tmp_path / "stats_car_detection.txt"
    stats.write_bytes(g["stats/text"].tobytes())
    np.testing.assert_allclose(K.parse_kitti_result(str(stats), use_40=False), g["stats/r11"], rtol=1e-12)
    np.testing.assert_allclose(K.parse_kitti_result(str(stats), use_40=True), g["stats/r40"], rtol=1e-12)
    # the hand-off: any executable taking (results, gt) that leaves stats_<class>_*.txt files behind
    ev = tmp_path / "evaluate_object"
    ev.write_text("#!/bin/sh\ncp \"$1/stats_car_detection.txt\" \"$1/stats_car_detection_3d.txt\"\n")
    ev.chmod(0o755)
    res = K.run_kitti_eval_script(str(ev), str(tmp_path), str(tmp_path), ["Car", "Pedestrian"])
    assert set(res) == {"det_2d_car", "det_3d_car"} and np.allclose(res["det_3d_car"], g["stats/r40"])
    assert abs(K.convertAlpha2Rot(0.3, 10.0, 2.0) - (0.3 + np.arctan2(2.0, 10.0))) < 1e-12


def test_torch_binding_builds_and_loads():
    """the C++ autograd binding (csrc/torch_binding.cpp): built in-tree, importable without a GPU, same ABI version as the library;
    it refuses host tensors (there is no CPU path behind it either)."""
    from groomed_nms_amd import build as b
    path = b.build_torch_binding()
    assert os.path.exists(path) and os.path.dirname(path) == os.path.dirname(b.OUT)
    from groomed_nms_amd import gnms_torch
    assert gnms_torch.abi_version() == 1
    with pytest.raises(RuntimeError):
        gnms_torch.layer(torch.rand(1, 4), torch.rand(1, 4, 4), None, None, 0, 0.4, 0.01, 0.3, 0, False, True, True, 100, False, True)
