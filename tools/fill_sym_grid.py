"""Store bandwidth of the SYMMETRIC writers' pattern (128 x 128 macro tiles direct + mirrored) from persistent workgroups that take the
tiles round robin, against the number of workgroups and their size.  python tools/fill_sym_grid.py [--boxes N]"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from groomed_nms_amd import _lib          # noqa: E402
from groomed_nms_amd._lib import ptr, check, stream_ptr  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--boxes", type=int, default=4096)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
lib = _lib.load()
B, N = a.batch, a.boxes
dev = torch.device("cuda")
nbuf = max(3, -(-(768 << 20) // (4 * B * N * N)))
bufs = [torch.empty((B, N, N), device=dev) for _ in range(nbuf)]
sp = stream_ptr(dev)
bytes_ = 4.0 * B * N * N
for tile, cpl in ((128, 2), (256, 4)):
    for block in (512, 1024):
        if tile == 256 and block == 512:
            continue
        for nt in (1, 0):
            for grid in (64, 96, 128, 160, 192, 248, 256, 512):
                os.environ["GNMS_FILL_GRID"] = str(grid)
                os.environ["GNMS_FILL_BLOCK"] = str(block)
                fn = lambda o: check(lib.gnms_profile_fill_sym(ptr(o), B, N, N, tile, cpl, nt, 2, sp), "fills")
                for i in range(3):
                    fn(bufs[i % nbuf])
                torch.cuda.synchronize()
                check(lib.gnms_profile_events(1), "arm")
                for i in range(a.reps):
                    fn(bufs[i % nbuf])
                torch.cuda.synchronize()
                check(lib.gnms_profile_events(0), "disarm")
                ms, n = ctypes.c_double(0), ctypes.c_int(0)
                check(lib.gnms_profile_collect(2, ctypes.byref(ms), ctypes.byref(n)), "collect")
                t = ms.value / max(n.value, 1)
                print("tile=%d block=%4d %s grid=%3d  %8.4f ms  %7.1f GB/s  %.3f of 8 TB/s" % (tile, block, "nt   " if nt else "plain", grid, t, bytes_ / t / 1e6, bytes_ / t / 1e6 / 8000), flush=True)
