#!/usr/bin/env python3
"""What the HBM takes from different STORE GEOMETRIES of an N x N matrix writer (no arithmetic), timed by the library's launch
events over rotating buffers like bench.py's ceiling: the linear fill, the band geometry of the 2D writers (persistent 16-wave
workgroups, R rows x 1 KiB per wave), and the symmetric-writer pattern (upper-triangular macro tiles, each stored directly and
mirrored).  python tools/store_geometry.py [--boxes 4096] [--batch 8]"""
import argparse
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from groomed_nms_amd import _lib  # noqa: E402
from groomed_nms_amd._lib import check, ptr, stream_ptr  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--boxes", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=8)
    args = ap.parse_args()
    lib = _lib.load()
    B, N = args.batch, args.boxes
    dev = torch.device("cuda", 0)
    nbytes = 4 * B * N * N
    n_buf = int(min(max(3, -(-(768 << 20) // nbytes)), 64))
    bufs = [torch.empty((B, N, N), dtype=torch.float32, device=dev) for _ in range(n_buf)]
    st = {"i": 0}

    def nxt():
        st["i"] = (st["i"] + 1) % n_buf
        return bufs[st["i"]]

    def rate(fn):
        ms, n = ctypes.c_double(0.0), ctypes.c_int(0)
        for _ in range(3):
            fn(nxt())
        torch.cuda.synchronize()
        check(lib.gnms_profile_collect(2, ctypes.byref(ms), ctypes.byref(n)), "collect")
        check(lib.gnms_profile_events(1), "events")
        for _ in range(max(8, 3 * n_buf)):
            fn(nxt())
        torch.cuda.synchronize()
        check(lib.gnms_profile_events(0), "events")
        check(lib.gnms_profile_collect(2, ctypes.byref(ms), ctypes.byref(n)), "collect")
        return nbytes * n.value / (ms.value * 1e-3) / 1e9, ms.value / n.value * 1e3

    sp = stream_ptr(dev)
    rows = []
    g, us = rate(lambda b: check(lib.gnms_profile_fill(ptr(b), B * N * N, sp), "fill"))
    rows.append({"pattern": "linear grid-stride float4 fill, non-temporal", "GB/s": round(g, 1), "us": round(us, 1)})
    for r in (8, 16):
        for nt in (1, 0):
            g, us = rate(lambda b: check(lib.gnms_profile_fill_tiles(ptr(b), B, N, N, r, nt, sp), "tiles"))
            rows.append({"pattern": "bands: persistent 16-wave WGs, %d rows x 1 KiB per wave, %s" % (r, "non-temporal" if nt else "plain stores"),
                         "GB/s": round(g, 1), "us": round(us, 1)})
    for tile, cpl in ((128, 2), (128, 4), (256, 4)):
        if N % tile:
            continue
        for persist in (0, 1):
            for nt in (1, 0):
                g, us = rate(lambda b: check(lib.gnms_profile_fill_sym(ptr(b), B, N, N, tile, cpl, nt, persist, sp), "sym"))
                rows.append({"pattern": "symmetric: %dx%d macro tiles direct + mirrored, %d-B row runs, %d floats per lane, %s, %s"
                                        % (tile, tile, tile * 4, cpl, "persistent strips" if persist else "one WG per tile",
                                           "non-temporal" if nt else "plain stores"), "GB/s": round(g, 1), "us": round(us, 1)})
    for r in rows:
        print(json.dumps(dict(r, N=N, B=B)))


if __name__ == "__main__":
    main()
