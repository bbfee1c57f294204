"""KITTI result files and the hand-off to the devkit's evaluation binary -- host-side mirror of the reference's inference tail
(SURVEY.md 8-f4): what happens to the boxes the NMS kept.

Same names, arguments and text format as lib/rpn_util.py (host NumPy / file I/O there as well):
  convert_image_predictions_to_correct_entries  :1489-1545   projected 3D centre + depth -> camera coordinates, alpha -> rotation_y,
                                                              optional uncertainty-weighted score
  get_text_to_write_in_kitti_format             :1571-1631   one 16-column KITTI line per box, 6 decimals (the devkit's parser is
                                                              sensitive to the precision, :1551-1558)
  write_image_boxes_to_txt_file                 :1547-1569   <save_folder>/<id>.txt
  parse_kitti_result / run_kitti_eval_script    :2013-2040 / :2043-2076   the C++ devkit (data/kitti_split1/devkit/cpp) is run as a
                                                              subprocess on the result folder; its stats_*.txt files are averaged
  convertAlpha2Rot / convertRot2Alpha / snap_to_pi   lib/util.py:630-682, lib/math_3d.py:497-510 (NumPy and scalar branches)
`conf` is anything with the reference's fields (`lbls`, optionally `has_un`, `use_un_for_score`) as attributes or keys.
"""
import copy
import math
import os
import re
import subprocess

import numpy as np

__all__ = ["convert_image_predictions_to_correct_entries", "get_text_to_write_in_kitti_format", "write_image_boxes_to_txt_file",
           "parse_kitti_result", "run_kitti_eval_script", "convertAlpha2Rot", "convertRot2Alpha", "snap_to_pi"]


def _cfg(conf, key, default=None):
    if isinstance(conf, dict):
        return conf.get(key, default)
    return getattr(conf, key, default)


def snap_to_pi(ry3d):
    """lib/math_3d.py:497-510: wrap into (-pi, pi]."""
    if isinstance(ry3d, np.ndarray):
        while np.any(ry3d > math.pi):
            ry3d[ry3d > math.pi] -= 2 * math.pi
        while np.any(ry3d <= -math.pi):
            ry3d[ry3d <= -math.pi] += 2 * math.pi
        return ry3d
    while ry3d > math.pi:
        ry3d -= math.pi * 2
    while ry3d <= -math.pi:
        ry3d += math.pi * 2
    return ry3d


def convertAlpha2Rot(alpha, z3d, x3d):
    """lib/util.py:630-654 (ndarray branch :640-644; the scalar branch ends on atan2(x, z), :648-649)."""
    if isinstance(z3d, np.ndarray):
        return snap_to_pi(alpha + np.arctan2(-z3d, x3d) + 0.5 * math.pi)
    return snap_to_pi(alpha + math.atan2(x3d, z3d))


def convertRot2Alpha(ry3d, z3d, x3d):
    """lib/util.py:657-682."""
    if isinstance(z3d, np.ndarray):
        return snap_to_pi(ry3d - np.arctan2(-z3d, x3d) - 0.5 * math.pi)
    return snap_to_pi(ry3d - math.atan2(-z3d, x3d) - 0.5 * math.pi)


def convert_image_predictions_to_correct_entries(boxes_img_input, conf, p2):
    """lib/rpn_util.py:1489-1545.  boxes_img [N][>=13(14)]: x1 y1 x2 y2 score cls | x3d y3d (pixels of the projected centre) z3d
    (depth) | w3d h3d l3d | alpha [| un].  Returns a copy with the true 3D centre (y at the bottom face), rotation_y instead of
    alpha, the score optionally weighted by `un`, and the three projected-centre columns appended."""
    boxes_img = copy.deepcopy(np.asarray(boxes_img_input))
    score = boxes_img[:, 4]
    x3d_2d, y3d_2d, z3d_2d = boxes_img[:, 6], boxes_img[:, 7], boxes_img[:, 8]
    h3d = boxes_img[:, 10]
    alpha = boxes_img[:, 12]
    p2_inv = np.linalg.inv(p2)                                                  # :1507
    pts = np.vstack((boxes_img[:, 6:9].T, np.ones((1, boxes_img.shape[0]))))    # lib/math_3d.py:88-97 backproject_2d_pixels_in_4D_format
    pts[0] = pts[0] * pts[2]
    pts[1] = pts[1] * pts[2]
    coord3d = np.matmul(p2_inv, pts).T                                          # N x 4
    x3d, y3d, z3d = coord3d[:, 0], coord3d[:, 1], coord3d[:, 2]
    y3d = y3d + h3d / 2                                                         # :1514
    ry3d = snap_to_pi(convertAlpha2Rot(alpha, z3d, x3d))                        # :1517-1518
    if _cfg(conf, "has_un"):
        un = score * boxes_img[:, 13]                                           # :1520-1521
        if _cfg(conf, "use_un_for_score"):
            score = un                                                          # :1523-1525
    boxes_img[:, 4] = score
    boxes_img[:, 6] = x3d
    boxes_img[:, 7] = y3d
    boxes_img[:, 8] = z3d
    boxes_img[:, 12] = ry3d
    # :1541-1543 "save projections of the 3d centers as well": x3d_2d / y3d_2d / z3d_2d are VIEWS of columns 6-8 in the reference,
    # taken before those columns are overwritten above -- so what it appends are the NEW columns (the camera-space centre) once more.
    # Reproduced as is (downstream code indexes columns 14-16).
    centers = np.hstack((x3d_2d[:, None], y3d_2d[:, None], z3d_2d[:, None]))
    return np.hstack((boxes_img, centers))


def get_text_to_write_in_kitti_format(boxes_img, conf, convention="kitti", save_constraint=False, precision=6):
    """lib/rpn_util.py:1571-1631: `cls -1 -1 alpha x1 y1 x2 y2 h w l x y z ry score` per box."""
    fmt = '{:.' + str(int(precision)) + 'f}'
    boxes_img = np.asarray(boxes_img)
    n = boxes_img.shape[0]
    lbls = _cfg(conf, "lbls")
    cls_ind = boxes_img[:, 5].astype(int) - 1
    x1, y1, x2, y2, score = (boxes_img[:, i] for i in range(5))
    x3d, y3d, z3d, w3d, h3d, l3d, ry3d = (boxes_img[:, i] for i in range(6, 13))
    alpha = convertRot2Alpha(ry3d, z3d, x3d)
    out = ""
    if convention == "kitti":
        for i in range(n):
            out += ('{} -1 -1' + (' ' + fmt) * 13).format(lbls[cls_ind[i]], alpha[i], x1[i], y1[i], x2[i], y2[i], h3d[i], w3d[i], l3d[i],
                                                          x3d[i], y3d[i], z3d[i], ry3d[i], score[i])
            if save_constraint:
                out += ((' ' + fmt) * 4).format(*(boxes_img[i, 17 + k] for k in range(4)))
            out += '\n'
    return out


def write_image_boxes_to_txt_file(boxes_img, conf, save_folder, id, convention="kitti", write=True, save_constraint=False, precision=6):
    """lib/rpn_util.py:1547-1569: writes <save_folder>/<id>.txt (always with 6 decimals, :1551-1560) and returns the text."""
    text = ""
    if convention == "kitti":
        text = get_text_to_write_in_kitti_format(boxes_img, conf, convention=convention, save_constraint=save_constraint, precision=6)
        if write:
            with open(os.path.join(save_folder, id + '.txt'), 'w') as f:
                f.write(text)
        else:
            print("Not writing to file!!!")
    return text


def parse_kitti_result(respath, use_40=False):
    """lib/rpn_util.py:2013-2040: three lines (easy / moderate / hard) of 41 recall-sampled precisions -> their means over the 40
    (use_40) or 11 recall positions."""
    acc = np.zeros([3, 41], dtype=float)
    with open(respath, 'r') as f:
        for lind, line in enumerate(f):
            for i, num in enumerate(re.findall(r'([\d]+\.?[\d]*)', line)):
                acc[lind, i] = float(num)
    sl = slice(1, 41, 1) if use_40 else slice(0, 41, 4)
    return np.mean(acc[0, sl]), np.mean(acc[1, sl]), np.mean(acc[2, sl])


def run_kitti_eval_script(eval_binary_path, results_data, gt_folder, lbls, use_40=True):
    """lib/rpn_util.py:2043-2076: runs the devkit binary (`evaluate_object <results> <gt>`) and collects the per-class stats files
    into {'det_2d_car': [easy, mod, hard], 'or_car': ..., 'gr_car': ..., 'det_3d_car': ...}."""
    with open(os.devnull, 'w') as devnull:
        subprocess.check_output([eval_binary_path, results_data, gt_folder], stderr=devnull)
    results = {}
    for lbl in lbls:
        lbl = lbl.lower()
        for key, name in (("det_2d_", "stats_{}_detection.txt"), ("or_", "stats_{}_orientation.txt"),
                          ("gr_", "stats_{}_detection_ground.txt"), ("det_3d_", "stats_{}_detection_3d.txt")):
            path = os.path.join(results_data, name.format(lbl))
            if os.path.exists(path):
                results[key + lbl] = list(parse_kitti_result(path, use_40=use_40))
    return results
