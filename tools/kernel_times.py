"""Stand-alone durations of the matrix-write kernels on ONE box (rotating buffers, hipExtLaunchKernel events through
gnms_profile_events): 2D IoU, guarded 3D NMS overlap, exact-order 3D, plain fill.  python tools/kernel_times.py [--boxes N] [--batch B]"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from groomed_nms_amd import _lib, synthetic          # noqa: E402
from groomed_nms_amd._lib import ptr, check, stream_ptr  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--boxes", type=int, default=4096)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--reps", type=int, default=30)
a = ap.parse_args()
lib = _lib.load()
B, N = a.batch, a.boxes
dev = torch.device("cuda")
b2, _ = synthetic.batch_2d(1, B, N, "clustered")
b3, _ = synthetic.batch_3d(1, B, N, True)
boxes2 = torch.from_numpy(b2).to(dev)
par3 = torch.from_numpy(b3).to(dev)
nbuf = max(3, -(-(768 << 20) // (4 * B * N * N)))
bufs = [torch.empty((B, N, N), device=dev) for _ in range(nbuf)]
sp = stream_ptr(dev)


def timed(fn, slot=0):
    for i in range(3):
        fn(bufs[i % nbuf])
    torch.cuda.synchronize()
    check(lib.gnms_profile_events(1), "arm")
    for i in range(a.reps):
        fn(bufs[i % nbuf])
    torch.cuda.synchronize()
    check(lib.gnms_profile_events(0), "disarm")
    ms, n = ctypes.c_double(0), ctypes.c_int(0)
    check(lib.gnms_profile_collect(slot, ctypes.byref(ms), ctypes.byref(n)), "collect")
    return ms.value / max(n.value, 1)


def evt(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(3):
        fn(bufs[i % nbuf])
    torch.cuda.synchronize()
    e0.record()
    for i in range(a.reps):
        fn(bufs[i % nbuf])
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / a.reps


bytes_ = 4.0 * B * N * N
bev = torch.empty((B, N, N), dtype=torch.float32, device=dev)
K3 = "iou3d_sym_kernel, each pair once, non-temporal stores"
rows = [("gnms_iou2d(boxes, boxes) [iou2d_self_kernel up to N = 4096, write_staged_kernel above]", timed(lambda o: check(lib.gnms_iou2d(ptr(boxes2), ptr(boxes2), B, N, N, ptr(o), N, sp), "iou2d"))),
        ("3D NMS overlap thr=0.4 [%s]" % K3, timed(lambda o: check(lib.gnms_nms_overlap3d_from_params(ptr(par3), B, N, 0.4, ptr(o), N, sp), "o3"))),
        ("3D NMS overlap thr=-100 (no pair in the band) [%s]" % K3, timed(lambda o: check(lib.gnms_nms_overlap3d_from_params(ptr(par3), B, N, -100.0, ptr(o), N, sp), "o3"))),
        ("iou3d_kernel<METHOD 2> (exact order)", timed(lambda o: check(lib.gnms_iou3d_from_params(ptr(par3), ptr(par3), B, N, N, 2, None, ptr(o), N, sp), "o3e"))),
        ("iou3d_kernel<generalized> (exact order, the reference-signature entry)", timed(lambda o: check(lib.gnms_iou3d_from_params(ptr(par3), ptr(par3), B, N, N, 1, None, ptr(o), N, sp), "o3g"))),
        ("iou3d_kernel<generalized, +iou_bev> (two matrices; GB/s of ONE)", timed(lambda o: check(lib.gnms_iou3d_from_params(ptr(par3), ptr(par3), B, N, N, 1, ptr(bev), ptr(o), N, sp), "o3gb"))),
        ("plain fill (same launch events)", timed(lambda o: check(lib.gnms_profile_fill(ptr(o), B * N * N, sp), "fill"), slot=2)),
        ("plain fill, writers' geometry (8 rows x 1 KiB per wave)", timed(lambda o: check(lib.gnms_profile_fill_tiles(ptr(o), B, N, N, 8, 1, sp), "fillt"), slot=2)),
        ("plain fill, writers' geometry (16 rows x 1 KiB per wave)", timed(lambda o: check(lib.gnms_profile_fill_tiles(ptr(o), B, N, N, 16, 1, sp), "fillt"), slot=2)),
        ("plain fill, writers' geometry, ordinary stores (8 rows)", timed(lambda o: check(lib.gnms_profile_fill_tiles(ptr(o), B, N, N, 8, 0, sp), "fillt"), slot=2)),
        ("plain fill (events around back-to-back launches)", evt(lambda o: check(lib.gnms_profile_fill(ptr(o), B * N * N, sp), "fill")))]
for name, ms in rows:
    print("%-92s %8.4f ms  %7.1f GB/s  %.3f of 8 TB/s" % (name, ms, bytes_ / ms / 1e6, bytes_ / ms / 1e6 / 8000))
