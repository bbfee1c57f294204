import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """Lazy view over one tests/golden/*.npz file; keys look like 'case/mode/field'."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name))
        self.keys = list(self.z.keys())

    def __getitem__(self, k):
        return self.z[k]

    def has(self, k):
        return k in self.z

    def cases(self):
        return sorted({k.split("/")[0] for k in self.keys})

    def modes(self, case):
        return sorted({k.split("/")[1] for k in self.keys if k.startswith(case + "/") and k.count("/") == 2
                       and not k.split("/")[1].startswith("groups_")})


@pytest.fixture(scope="session")
def golden_nms():
    return Golden("nms_small.npz")


@pytest.fixture(scope="session")
def golden_box2d():
    return Golden("boxes_2d.npz")


@pytest.fixture(scope="session")
def golden_box3d():
    return Golden("boxes_3d.npz")


@pytest.fixture(scope="session")
def golden_misc():
    return Golden("misc.npz")


@pytest.fixture(scope="session")
def golden_f64site():
    return Golden("f64site.npz")


def f64site_aboxes(g, c):
    """`aboxes` of case c of tests/golden/f64site.npz: float64 proposals defined from the stored fp32 arrays by one IEEE double
    multiplication (tests/golden/make_golden.py::make_f64_call_site), scores widened as np.hstack does (lib/rpn_util.py:1258)."""
    b = g["d2/boxes32"][c].astype(np.float64) * float(g["f64_scale"])
    return np.hstack((b, g["d2/scores32"][c].astype(np.float64)[:, np.newaxis]))


# mode tag -> kwargs, mirrors tests/golden/make_golden.py::MODES
MODES = {
    "gm_lin": dict(group_boxes=True, mask_group_boxes=True, pruning_method="linear"),
    "gm_lin_gs2": dict(group_boxes=True, mask_group_boxes=True, pruning_method="linear", group_size=2),
    "gm_lin_sorted": dict(group_boxes=True, mask_group_boxes=True, pruning_method="linear", return_sorted_prob=True),
    "gm_sig": dict(group_boxes=True, mask_group_boxes=True, pruning_method="sigmoidal", temperature=0.1),
    "gm_soft": dict(group_boxes=True, mask_group_boxes=True, pruning_method="soft_nms", temperature=0.5),
    "gu_lin": dict(group_boxes=True, mask_group_boxes=False, pruning_method="linear"),
    "gu_lin_gs2": dict(group_boxes=True, mask_group_boxes=False, pruning_method="linear", group_size=2),
    "gu_sig": dict(group_boxes=True, mask_group_boxes=False, pruning_method="sigmoidal", temperature=0.1),
    "gu_soft": dict(group_boxes=True, mask_group_boxes=False, pruning_method="soft_nms", temperature=0.1),
    "un_lin": dict(group_boxes=False, pruning_method="linear"),
    "un_lin_sorted": dict(group_boxes=False, pruning_method="linear", return_sorted_prob=True),
    "un_sig": dict(group_boxes=False, pruning_method="sigmoidal", temperature=0.1),
    "un_soft": dict(group_boxes=False, pruning_method="soft_nms", temperature=0.5),
    "gm_lin_thr": dict(group_boxes=True, mask_group_boxes=True, pruning_method="linear", nms_threshold=0.6,
                       valid_box_prob_threshold=0.5),
}

TOL = 1e-4   # north_star: outputs within 1e-4 (fp32) of the reference


def check_index_lists(got_valid, got_invalid, ref_valid, ref_invalid, prob_sorted_desc=None):
    """valid/invalid are compared as sets: the reference's order among equal probabilities (the many
    exact zeros) is whatever torch.sort does (lib/groomed_nms.py:117,121)."""
    assert sorted(map(int, got_valid)) == sorted(map(int, ref_valid))
    assert sorted(map(int, got_invalid)) == sorted(map(int, ref_invalid))
