#!/usr/bin/env python3
"""bench.py -- GrooMeD-NMS fwd+bwd boxes/sec on MI355X (BASELINE.json metric).

A step = one pass of the hot path over one batch of synthetic proposals already resident in HBM:
    pairwise IoU matrix  ->  GrooMeD-NMS forward  ->  backward w.r.t. scores
as lib/loss/rpn_3d.py:772-791 runs it.  Default (2D): ONE library call builds the matrix and runs the layer
(gnms_forward_with_iou2d: the matrix is an output, the threshold bits come straight from the boxes), then gnms_backward;
--two-calls keeps the overlap kernel and the matrix-in layer gnms_forward apart; --dim 3 uses gnms_forward_with_iou3d.
The matrix is written into one of >= 3 rotating buffers (more than 256 MiB apart in time), so that the 256 MiB Infinity Cache
cannot absorb part of the write between repetitions.

Multi-GPU (weak scaling): every rank owns `--batch` images x `--boxes` boxes; images are independent units, so the layer has no
data-path collective (DESIGN.md "multi-GPU"); one 4-byte RCCL all-reduce per step (dist.StepHeartbeat) puts a real collective
round trip into the curve, as SURVEY.md 8-e asks.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python bench.py --gpus 8 ...          # starts 8 ranks itself (torch.distributed.run, one process per GPU, RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...

Prints ONE JSON line on rank 0 (see the driver contract).  Extra objects:
  roofline            the dominant HBM-bound launch of the timed step: algorithmic bytes / its HIP-event duration INSIDE the step
                      sequence (the library brackets the launch with events on its launch stream, gnms_profile_events), against the
                      8 TB/s HBM peak; `ceiling` = what a plain store (load) stream reaches on this device over the same buffers.
                      default: the launch that writes the matrix; --two-calls: bitmask_kernel, the one full read of the matrix
  roofline_iou / roofline_matrix_in   the other of the two under --two-calls
  cpu_baseline        the CPU oracle (a C port of the reference algorithm, single thread) timed on this host, N=1 only
  parity              max |d prob| and max |d grad_scores| of the timed entry against that oracle on every image of rank 0's batch,
                      and whether the valid-index sets agree -- the "max-|dscore| vs ref" half of BASELINE.json's metric
  other_kind          the same step on the other box generator (uniform <-> clustered), 1 GPU only
  box_counts          N > 1 only: the other box counts north_star names (256, 1024, 16384 per image) at this world size -- whole-job boxes/s,
                      timed like the headline (barrier + sync on both sides, MAX over ranks); at N = 1 the keys N256 / N1024 / N16384 hold them
  hip_graph_replay    the same step as a replayed HIP graph of the C-ABI calls (no host cost per step), 1 GPU only
  two_calls, dim3_*, N*, B*_N4096   the other shapes of the path (default line only), each with ms_per_step (eager, host included; the median of
                      three windows of its `steps`, all three in windows_ms), its own
                      roofline brief and device_ms_per_step: the step's time on the device when the host has run ahead (the same
                      eager calls enqueued behind a spin kernel; small shapes are host-bound and hosts differ by 2x between boxes);
                      host-bound one-call shapes also carry c_abi_ms_per_step: the same step as the C-ABI call sequence of a compiled
                      host (forward entry + gnms_backward through ctypes, eager, no autograd engine)
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s
REF_CPU_SURVEY = 459.0      # reference lib/groomed_nms.py + lib/core.py iou, torch CPU, 8 threads, uniform N=4096 (not published)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--boxes", type=int, default=4096, help="boxes per image (BASELINE metric: N=4096/img)")
    ap.add_argument("--batch", type=int, default=8, help="images per GPU")
    ap.add_argument("--kind", default="uniform", choices=["uniform", "clustered"],
                    help="box generator: uniform random boxes (the distribution of the reference figure in BASELINE.md; the headline since round 3) or "
                         "clustered ones (K = N/64 objects x 64 jittered copies: detector-like; the headline of rounds 1-2, now `other_kind`)")
    ap.add_argument("--dim", type=int, default=2, choices=[2, 3], help="2: lib/core.py iou; 3: 0.5*(1+GIoU3D) of the corner AABBs from (x,y,z,w,h,l,ry)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-kind", action="store_true")
    ap.add_argument("--two-calls", action="store_true", help="overlap kernel and gnms_forward as two library calls (matrix-in layer)")
    ap.add_argument("--graph", action="store_true",
                    help="capture one step (IoU + forward + backward through the C ABI, preallocated buffers) in a HIP graph and replay "
                         "it: removes the per-launch host cost that bounds small problems (N <= 1024)")
    ap.add_argument("--sorted-scores", action="store_true",
                    help="feed scores already sorted by descending value, as both reference call sites do (lib/loss/rpn_3d.py:731-737, "
                         "lib/rpn_util.py:1258-1266)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU baseline: keep processing images of the batch for about this long")
    ap.add_argument("--pmc-summary", default=None,
                    help="JSON written by tools/pmc.sh in the same session ({kernel: HBM bytes per launch}); fills roofline.traffic.  Without it "
                         "the default run collects the counters itself (two short rocprofv3 --pmc passes of this script) when rocprofv3 is on PATH")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip what the default 1-GPU line adds by re-running this script: the PMC passes for roofline.traffic, the `two_calls` "
                         "keys (the reference's unchanged call sites in 2D and 3D), the `dim3_*` keys (3D, N = 4096 and 16384) and the other box counts / batches "
                         "north_star names (N256, N1024, N16384, B1_N4096, B4_N4096), timed in this process")
    return ap.parse_args()


def launch_ranks_if_needed(args):
    """`--gpus N` with no launcher around us: become the launcher.  Fails loudly when the node has fewer GPUs."""
    ws = os.environ.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != args.gpus:
            sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (args.gpus, ws))
        return
    if args.gpus <= 1:
        return
    import torch
    from groomed_nms_amd import dist as gdist
    have = torch.cuda.device_count()
    if have < args.gpus and not gdist.share_gpu():          # (GNMS_SHARE_GPU=1: debug mode, ranks fold onto the visible devices over gloo)
        sys.exit("bench.py: --gpus %d requested but this node has %d visible GPU(s); refusing to time fewer GPUs than asked" % (args.gpus, have))
    sys.exit(gdist.relaunch_under_torchrun(args.gpus, os.path.abspath(__file__), sys.argv[1:]))


def _sub_bench(extra, timeout=900):
    """this script once more with other flags (a child process on the same GPU, this one idle meanwhile); its JSON line as a dict"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--no-other-kind", "--no-extras"] + list(extra)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError("rc %d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:]))
    return json.loads(lines[-1])


def _brief(d):
    """what the extra keys keep of a child's line"""
    def roof(r):
        return None if not r else {k: r[k] for k in ("kernel", "kernel_ms", "achieved", "frac", "launches_per_step", "algorithmic_bytes") if k in r}
    out = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "workload": d["config"]["workload"],
           "roofline": roof(d.get("roofline"))}
    if d.get("roofline_iou"):
        out["roofline_iou"] = roof(d["roofline_iou"])
    return out


def _pmc_traffic(workload_args, B, N):
    """HBM bytes per launch from the PMC counters, collected as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE
    rocprofv3 passes with --kernel-trace only, over a short run of this script.  Corrections: the counters are in KiB; FETCH_SIZE
    tallies the 128-byte requests of wide coalesced reads as 64 B on gfx950 -> doubled; WRITE_SIZE is calibrated in the same pass on
    gnms_profile_fill's known store stream (prof_fill_kernel writes exactly 4 B N^2 bytes per launch).  -> ({kernel: bytes}, note)"""
    import collections
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return {}, "rocprofv3 not on PATH"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["TMPDIR"] = "/tmp"
    raw = {}
    with tempfile.TemporaryDirectory(dir="/tmp", prefix="gnms_pmc_") as tmp:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "-d", d, "-o", "pmc", "--output-format", "csv", "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-other-kind", "--no-extras"] + list(workload_args)
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd="/tmp")
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return {}, "rocprofv3 --pmc %s wrote no counter file (rc %d)" % (counter, r.returncode)
            agg = collections.defaultdict(list)
            with open(files[0]) as f:
                for row in csv.DictReader(f):
                    if row.get("Counter_Name") == counter:
                        name = re.sub(r"^void ", "", row["Kernel_Name"])
                        name = re.sub(r"\(anonymous namespace\)::|gnms::|gnms_iou3d::|gnms_iou::", "", name)
                        agg[re.sub(r"[<(].*", "", name)].append(float(row["Counter_Value"]))
            raw[counter] = {k: sum(v) / len(v) for k, v in agg.items()}
    fill = raw["WRITE_SIZE"].get("prof_fill_kernel")
    wcal = (4.0 * B * N * N / 1024.0) / fill if fill else 1.0
    traffic = {}
    for k in set(raw["FETCH_SIZE"]) | set(raw["WRITE_SIZE"]):
        traffic[k] = raw["FETCH_SIZE"].get(k, 0.0) * 1024 * 2 + raw["WRITE_SIZE"].get(k, 0.0) * 1024 * wcal
    return traffic, ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes with --kernel-trace only, over `bench.py --steps 5`; KiB counters, "
                     "FETCH_SIZE x2 (gfx950), WRITE_SIZE x %.4f (calibrated on prof_fill_kernel's known byte count in the same pass)" % wcal)


def main():
    args = parse()
    launch_ranks_if_needed(args)
    import torch
    import groomed_nms_amd as G
    from groomed_nms_amd import overlaps, synthetic, _lib, dist as gdist
    from groomed_nms_amd._lib import GnmsParams, ptr, stream_ptr, check
    world, rank, local_rank = gdist.env_world()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    gdist.init(backend="nccl")                     # RCCL; no-op at world size 1
    lib = _lib.load()
    B, N = args.batch, args.boxes
    P = GnmsParams()
    lib.gnms_default_params(ctypes.byref(P))
    thr = float(P.nms_threshold)

    def make_inputs(kind):
        # every rank owns its own `B` images (weak scaling): rank r draws the images [r*B, (r+1)*B) of the global batch
        if args.dim == 2:
            b_np, s_np = synthetic.batch_2d(1000 + rank, B, N, kind)
        else:
            b_np, s_np = synthetic.batch_3d(1000 + rank, B, N, clustered=(kind == "clustered"))
        if args.sorted_scores:
            o = np.argsort(-s_np, axis=1, kind="stable")
            s_np = np.take_along_axis(s_np, o, axis=1)
            b_np = np.take_along_axis(b_np, o[:, :, None], axis=1)
        return np.ascontiguousarray(b_np), np.ascontiguousarray(s_np)

    if args.no_other_kind:
        args.no_extras = True                               # (every tool that asks for the bare line passes --no-other-kind)
    boxes_np, scores_np = make_inputs(args.kind)
    w_np = np.linspace(-1.0, 2.0, N).astype(np.float32)                 # dL/dprob, the SAME fp32 values on the GPU and in the oracle
    w = torch.from_numpy(np.tile(w_np, (B, 1))).to(dev).contiguous()
    # rotating matrix buffers: the write of step i lands 3+ buffers (> 768 MiB, or 3 buffers) after the last write to the same lines
    mat_bytes = 4 * B * N * N
    n_buf = int(min(max(3, -(-(768 << 20) // max(mat_bytes, 1))), 64))
    iou_bufs = [torch.empty((B, N, N), dtype=torch.float32, device=dev) for _ in range(n_buf)]
    state = {"i": 0}

    def next_buf():
        state["i"] = (state["i"] + 1) % n_buf
        return iou_bufs[state["i"]]

    def make_step(boxes, scores):
        def build_overlaps(out):
            if args.dim == 2:
                return overlaps.iou_batched(boxes, out=out)
            return overlaps.iou3d_batched(boxes, from_params=True, nms_overlap=True, out=out, nms_threshold=thr)

        def step():
            buf = next_buf()
            if args.dim == 2 and not args.two_calls:
                prob = G.differentiable_nms_with_iou2d_batched(scores, boxes, iou_out=buf)[0]       # gnms_forward_with_iou2d
            elif args.dim == 3 and not args.two_calls:
                prob = G.differentiable_nms_with_iou3d_batched(scores, boxes, iou_out=buf)[0]       # gnms_forward_with_iou3d
            else:
                prob = G.differentiable_nms_batched(scores, build_overlaps(buf))[0]
            scores.grad = None
            torch.autograd.backward(prob, w)          # dL/dprob = w
            return prob
        return step, build_overlaps

    boxes = torch.from_numpy(boxes_np).to(dev)
    scores = torch.from_numpy(scores_np).to(dev).requires_grad_(True)
    step, build_overlaps = make_step(boxes, scores)
    eager_step = step

    def graph_steps():
        """The step as the C-ABI call sequence (forward entry + gnms_backward on fixed buffers), eagerly and captured into HIP graphs:
        returns (replay_step, eager_raw_step)."""
        ws_g = torch.empty((lib.gnms_workspace_bytes(B, N, ctypes.byref(P)),), dtype=torch.uint8, device=dev)
        prob_g = torch.empty((B, N), dtype=torch.float32, device=dev)
        grad_g = torch.empty((B, N), dtype=torch.float32, device=dev)
        s_det = scores.detach()

        def raw_step(buf):
            sp = stream_ptr(dev)
            if args.dim == 2 and not args.two_calls:
                check(lib.gnms_forward_with_iou2d(ptr(boxes), ptr(s_det), B, N, N, None, ctypes.byref(P), ptr(buf), ptr(prob_g), None, None,
                                                  None, None, None, ptr(ws_g), ws_g.numel(), sp), "fwd_with_iou2d")
            elif not args.two_calls:
                check(lib.gnms_forward_with_iou3d(ptr(boxes), ptr(s_det), B, N, N, None, ctypes.byref(P), ptr(buf), ptr(prob_g), None, None,
                                                  None, None, None, ptr(ws_g), ws_g.numel(), sp), "fwd_with_iou3d")
            else:
                if args.dim == 2:
                    check(lib.gnms_iou2d(ptr(boxes), ptr(boxes), B, N, N, ptr(buf), N, sp), "iou2d")
                else:
                    check(lib.gnms_nms_overlap3d_from_params(ptr(boxes), B, N, thr, ptr(buf), N, sp), "overlap3d")
                check(lib.gnms_forward(ptr(s_det), ptr(buf), B, N, N, None, ctypes.byref(P), ptr(prob_g), None, None, None, None, None,
                                       ptr(ws_g), ws_g.numel(), sp), "fwd")
            check(lib.gnms_backward(ptr(w), ptr(s_det), ptr(buf), B, N, N, None, ctypes.byref(P), ptr(grad_g), None, ptr(ws_g),
                                    ws_g.numel(), sp), "bwd")

        raw_step(iou_bufs[0])
        torch.cuda.synchronize()
        graphs = []
        for buf in iou_bufs[:3]:                      # one captured step per rotating buffer
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                raw_step(buf)
            graphs.append(g)
        gstate = {"i": 0}

        def replay():                                 # one replay = one full step
            gstate["i"] = (gstate["i"] + 1) % len(graphs)
            graphs[gstate["i"]].replay()
        return replay, (lambda: raw_step(next_buf()))

    if args.graph:
        step, eager_step = graph_steps()              # (eager: the same C-ABI sequence; events cannot be recorded under replay)

    heartbeat = gdist.StepHeartbeat(dev)
    # a short run (the driver's K = 20 is 3 ms of GPU time) would otherwise be timed on clocks that are still ramping and on first-launch
    # work (module load, allocator growth): run untimed steps up to a fixed total before the W warm-up steps the contract asks for
    prewarm = int(os.environ.get("GNMS_BENCH_PREWARM", max(0, 100 - args.warmup)))    # (the env override: tests of the launcher path on big shapes)
    for _ in range(prewarm):
        step()
    torch.cuda.synchronize()
    dt = gdist.timed_steps(step, args.steps, args.warmup, torch.cuda.synchronize, heartbeat)
    # `spread`: the contract's window above is ONE sample of K steps; four more windows of the same K steps right behind it say how far such
    # a sample moves on this box (1 GPU only; `value` is and stays the first window)
    windows = [dt]
    if world == 1 and not args.no_extras:
        for _ in range(4):
            windows.append(gdist.timed_steps(step, args.steps, 0, torch.cuda.synchronize, heartbeat))

    # N > 1: the other box counts north_star names (256, 1024, 16384 per image), every rank its own images, timed like the headline --
    # barrier + device sync on both sides, the MAX over ranks -- so that the scaling run carries the whole grid at every world size (at
    # N = 1 the keys N256 / N1024 / N16384 below are the same shapes with their roofline briefs).  A rank that cannot set a shape up says so
    # BEFORE the timed region's collectives: either every rank times it or none does.
    box_counts = None
    if world > 1 and not args.no_extras and not args.graph and not args.two_calls and args.dim == 2 and not args.sorted_scores:
        box_counts = {}
        for n_, k_ in ((256, 60), (1024, 60), (16384, 6)):
            if n_ == N:
                continue
            one_, err_ = None, ""
            try:
                bx_np, sc_np = synthetic.batch_2d(1000 + rank, B, n_, args.kind)
                bx_ = torch.from_numpy(np.ascontiguousarray(bx_np)).to(dev)
                sc_ = torch.from_numpy(np.ascontiguousarray(sc_np)).to(dev).requires_grad_(True)
                ww_ = torch.from_numpy(np.tile(np.linspace(-1.0, 2.0, n_).astype(np.float32), (B, 1))).to(dev).contiguous()
                nb_ = int(min(max(2, -(-(768 << 20) // max(4 * B * n_ * n_, 1))), 16))
                bufs_ = [torch.empty((B, n_, n_), dtype=torch.float32, device=dev) for _ in range(nb_)]
                st_ = {"i": 0}

                def one_(bx_=bx_, sc_=sc_, ww_=ww_, bufs_=bufs_, st_=st_, nb_=nb_):
                    st_["i"] = (st_["i"] + 1) % nb_
                    prob = G.differentiable_nms_with_iou2d_batched(sc_, bx_, iou_out=bufs_[st_["i"]])[0]
                    sc_.grad = None
                    torch.autograd.backward(prob, ww_)
                for _ in range(6):
                    one_()
                torch.cuda.synchronize()
            except Exception as e:                                  # (e.g. out of memory beside another tenant)
                one_, err_ = None, str(e)[:200]
            if int(round(gdist.sum_over_ranks(1.0 if one_ is not None else 0.0))) == world:
                dt_ = gdist.timed_steps(one_, k_, 2, torch.cuda.synchronize)
                box_counts["N%d" % n_] = {"value": round(world * B * n_ * k_ / dt_, 1), "unit": "boxes/s", "ms_per_step": round(dt_ / k_ * 1e3, 4), "steps": k_,
                                          "workload": "%d images/GPU x %d %s 2D boxes, %d GPUs" % (B, n_, args.kind, world)}
            else:
                box_counts["N%d" % n_] = {"error": err_ or "another rank could not set this shape up"}
            one_ = None
            bx_ = sc_ = ww_ = bufs_ = None
            torch.cuda.empty_cache()

    out = None
    if rank == 0:
        one_call = not args.two_calls
        per_box = 16.0 if args.dim == 2 else 28.0
        alg_write = B * (4.0 * N * N + per_box * N)       # IoU-2D 4N^2 + 16N, IoU-3D (params) 4N^2 + 28N per image
        alg_read = B * (4.0 * N * N + 16.0 * N)           # NMS forward given the matrix: 4N^2 + 16N per image
        if args.sorted_scores:                            # only the columns a leader of the row block can sit in are needed
            alg_read = B * (sum(64 * min(N, 64 * (kb + 1)) for kb in range((N + 63) // 64)) * 4.0 + 16.0 * N)

        # ---- roofline: the SAME step sequence once more, the library bracketing its HBM-bound launches with events ----
        def collect(slot):
            ms, n = ctypes.c_double(0.0), ctypes.c_int(0)
            check(lib.gnms_profile_collect(slot, ctypes.byref(ms), ctypes.byref(n)), "profile_collect")
            return ms.value, n.value

        # (N > 1 ranks: the other ranks sit in the final barrier while rank 0 is in here -- a short event collection, no ceiling streams)
        k_roof = max(50, min(args.steps, 200)) if world == 1 else max(5, min(args.steps, 20))
        for _ in range(3):
            eager_step()
        torch.cuda.synchronize()
        check(lib.gnms_profile_events(1), "profile_events")
        for _ in range(k_roof):
            eager_step()
        torch.cuda.synchronize()
        check(lib.gnms_profile_events(0), "profile_events")
        ms_write, n_write = collect(0)
        ms_read, n_read = collect(1)

        # achievable ceilings: a plain non-temporal store / load stream over the same rotating buffers
        def stream_rate(fn, nbytes):
            # timed like the kernel it is the ceiling of: the dispatch's own begin / end timestamps (slot 2), launch gaps excluded
            for _ in range(2):
                fn(next_buf())
            torch.cuda.synchronize()
            collect(2)
            check(lib.gnms_profile_events(1), "gnms_profile_events")
            for _ in range(max(6, 2 * n_buf)):
                fn(next_buf())
            torch.cuda.synchronize()
            check(lib.gnms_profile_events(0), "gnms_profile_events")
            ms, n = collect(2)
            return nbytes * n / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        sink = torch.zeros(4, dtype=torch.float32, device=dev)
        n_fl = B * N * N // 4 * 4
        fill_gbs = stream_rate(lambda b: check(lib.gnms_profile_fill(ptr(b), n_fl, stream_ptr(dev)), "fill"), 4.0 * n_fl) if (n_fl and world == 1) else 0.0
        fill_what = "plain non-temporal float4 store stream (gnms_profile_fill)"
        for rows_, nt_ in ((8, 1), (16, 1), (8, 0), (16, 0)):   # the same stream in the writers' geometry; the best of the five is the ceiling
            if N % rows_ == 0 and N >= 256 and world == 1:
                tiles_gbs = stream_rate(lambda b: check(lib.gnms_profile_fill_tiles(ptr(b), B, N, N, rows_, nt_, stream_ptr(dev)), "fill_tiles"), 4.0 * B * N * N)
                if tiles_gbs > fill_gbs:
                    fill_gbs = tiles_gbs
                    fill_what = "plain %s store stream, persistent 16-wave workgroups, %d rows x 1 KiB per wave (gnms_profile_fill_tiles)" % (
                        "non-temporal" if nt_ else "16-byte", rows_)
        read_gbs = 0.0
        if n_fl and world == 1:
            for b_ in iou_bufs:
                b_.fill_(0.25)
            read_gbs = stream_rate(lambda b: check(lib.gnms_profile_read(ptr(b), n_fl, ptr(sink), stream_ptr(dev)), "read"), 4.0 * n_fl)

        pmc, pmc_note = {}, None
        workload_args = ["--boxes", str(N), "--batch", str(B), "--kind", args.kind, "--dim", str(args.dim)] + (["--two-calls"] if args.two_calls else []) + (
            ["--sorted-scores"] if args.sorted_scores else [])
        if args.pmc_summary:
            with open(args.pmc_summary) as f:
                pmc = json.load(f).get("traffic_bytes_per_launch", {})
            pmc_note = "tools/pmc.sh, " + args.pmc_summary
        elif world == 1 and not args.no_extras and not args.graph:
            try:
                pmc, pmc_note = _pmc_traffic(workload_args, B, N)
            except Exception as e:                            # the counters are evidence beside the line, never a reason to lose it
                pmc, pmc_note = {}, "PMC pass failed: %s" % (str(e)[:200],)

        def roof(ms_sum, launches, nbytes_per_step, kname, ceiling, what):
            if launches == 0 or ms_sum <= 0:
                return None
            per_step_ms = ms_sum / k_roof
            ach = nbytes_per_step / (per_step_ms * 1e-3) / 1e9
            return {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": (round(pmc[kname]) if kname in pmc else None), "traffic_source": pmc_note, "algorithmic_bytes": round(nbytes_per_step),
                    "kernel": kname, "kernel_ms": round(per_step_ms, 4), "launches_per_step": round(launches / k_roof, 2),
                    "ceiling": ({"what": what, "GB/s": round(ceiling, 1), "frac_of_ceiling": round(ach / ceiling, 4)} if ceiling else None),
                    "measured": "HIP events around the launch inside %d repetitions of the timed step sequence, %d rotating %d-MiB matrix buffers"
                                % (k_roof, n_buf, mat_bytes >> 20)}

        if one_call:
            wname = lib.gnms_profile_write_kernel_name(args.dim, B, N).decode()
        else:
            # gnms_iou2d of a box set with itself: the staged writers as a launch of their own up to N = 4096 (iou2d_self_kernel) when the
            # batch has enough units, the large-matrix writers above (write_staged_kernel)
            self_ok = N <= 4096 and B <= 127 and B * ((N + 7) // 8) >= 8 * 256
            wname = ("iou2d_self_kernel" if self_ok else ("write_staged_kernel" if N > 4096 and N % 4 == 0 else "iou2d_kernel")) if args.dim == 2 else "iou3d_sym_kernel"
        r_write = roof(ms_write, n_write, alg_write, wname, fill_gbs, fill_what)
        r_read = roof(ms_read, n_read, alg_read, "one_launch_kernel" if N <= 1024 and os.environ.get("GNMS_ONE_LAUNCH", "1") != "0" else "bitmask_kernel", read_gbs, "plain non-temporal float4 load stream (gnms_profile_read)")

        total_boxes = world * B * N * args.steps
        value = total_boxes / dt
        out = {
            "metric": "GrooMeD-NMS fwd+bwd boxes/sec (pairwise IoU + differentiable_nms forward + backward wrt scores)",
            "value": round(value, 1),
            "unit": "boxes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%d images/GPU x %d %s %dD boxes/image, nms_threshold 0.4, linear pruning, grouped+masked, group_size 100"
                                   % (B, N, args.kind, args.dim), "boxes_per_image": N, "images_per_gpu": B,
                       "scores_presorted": bool(args.sorted_scores), "hip_graph_replay": bool(args.graph), "untimed_steps_before_the_warmup": prewarm, "parallelism": "images sharded, dp%d" % world,
                       "collective": ("one 4-byte all-reduce (SUM) per step over %s, issued asynchronously like DDP's gradient all-reduce (it "
                                      "overlaps the next step; not a blocking round trip); every step's sum verified == world_size after the "
                                      "timed region" % ("gloo [GNMS_SHARE_GPU debug mode]" if gdist.share_gpu() else "RCCL/xGMI")) if world > 1 else "none (1 GPU)",
                       "matrix_buffers": n_buf},
            "roofline": None,
            "reference_cpu_survey": {"value": REF_CPU_SURVEY, "unit": "boxes/s", "ratio_per_gpu": round(value / world / REF_CPU_SURVEY, 1),
                                     "note": "reference lib/groomed_nms.py + lib/core.py iou on torch CPU, 8 threads, uniform N=4096, measured in the "
                                             "survey container (BASELINE.md section 2); not a published number, hence vs_baseline null"},
        }
        if len(windows) > 1:
            wm = sorted(w_ / args.steps * 1e3 for w_ in windows)
            out["spread"] = {"windows": len(wm), "steps_per_window": args.steps, "min_ms": round(wm[0], 4), "median_ms": round(wm[len(wm) // 2], 4),
                             "max_ms": round(wm[-1], 4), "first_window_ms": round(dt / args.steps * 1e3, 4),
                             "what": "ms per step of %d consecutive windows of --steps steps in this run; `value` is the first" % len(wm)}
        if box_counts:
            out["box_counts"] = box_counts
        if gdist.share_gpu() and world > 1:
            out["debug_shared_gpu"] = "GNMS_SHARE_GPU=1: %d ranks share %d GPU(s), gloo collectives -- launcher-path check, not a scaling number" % (
                world, torch.cuda.device_count())
        if one_call:
            out["roofline"] = r_write
        else:
            out["roofline"], out["roofline_iou"] = r_read, r_write

        if world == 1 and not args.no_other_kind and not args.graph:
            # the same step as a replayed HIP graph of the C-ABI calls: what the GPU alone takes -- the eager line above includes the host's
            # cost of issuing a step through torch.autograd (85-150 us, box dependent), which the GPU time is now within 1.5x of
            replay, _ = graph_steps()
            k_g = max(50, args.steps // 2)
            dtg = gdist.timed_steps(replay, k_g, max(3, args.warmup // 2), torch.cuda.synchronize)
            out["hip_graph_replay"] = {"value": round(B * N * k_g / dtg, 1), "unit": "boxes/s", "ms_per_step": round(dtg / k_g * 1e3, 4), "steps": k_g,
                                       "what": "forward entry + gnms_backward captured once per rotating buffer, replayed"}
        if world == 1 and not args.no_other_kind and not args.graph:
            other = "uniform" if args.kind == "clustered" else "clustered"
            ob_np, os_np = make_inputs(other)
            ob = torch.from_numpy(ob_np).to(dev)
            osc = torch.from_numpy(os_np).to(dev).requires_grad_(True)
            ostep, _ = make_step(ob, osc)
            k_o = max(50, args.steps // 2)
            dto = gdist.timed_steps(ostep, k_o, max(3, args.warmup // 2), torch.cuda.synchronize)
            out["other_kind"] = {"kind": other, "value": round(B * N * k_o / dto, 1), "unit": "boxes/s", "ms_per_step": round(dto / k_o * 1e3, 4),
                                 "steps": k_o}

        if world == 1 and not args.no_extras and not args.graph and not args.two_calls and args.dim == 2 and not args.sorted_scores:
            # driver-timed figures for the other shapes of the path, all on the headline's generator, timed in this
            # process with the same timed_steps: the reference's UNCHANGED call sites -- iou() then differentiable_nms(scores, iou), the
            # overlap kernel and the matrix-in layer as two library calls (lib/loss/rpn_3d.py:772-791) -- in 2D and in 3D (corners ->
            # iou3d_approximate(generalized) -> 0.5 (1 + giou) -> differentiable_nms, rpn_3d.py:776-791), the one-call 3D entry at N = 4096 and
            # C5's N = 16384, the other box counts north_star names (256, 1024, 16384) and the per-GPU batches of C3 / C4 (B = 1, B = 4).
            def shape(dim, b_, n_, two_calls=False, ref_3d=False, steps=20, warmup=3):
                if dim == 2:
                    bx_np, sc_np = synthetic.batch_2d(1000, b_, n_, args.kind)
                else:
                    bx_np, sc_np = synthetic.batch_3d(1000, b_, n_, clustered=(args.kind == "clustered"))
                bx = torch.from_numpy(np.ascontiguousarray(bx_np)).to(dev)
                sc = torch.from_numpy(np.ascontiguousarray(sc_np)).to(dev).requires_grad_(True)
                ww = torch.from_numpy(np.tile(np.linspace(-1.0, 2.0, n_).astype(np.float32), (b_, 1))).to(dev).contiguous()
                mb = 4 * b_ * n_ * n_
                nb_ = int(min(max(3, -(-(768 << 20) // max(mb, 1))), 16))
                bufs = [torch.empty((b_, n_, n_), dtype=torch.float32, device=dev) for _ in range(nb_)]
                st_ = {"i": 0}

                def one():
                    st_["i"] = (st_["i"] + 1) % nb_
                    buf = bufs[st_["i"]]
                    if ref_3d:          # the reference's own sequence of calls, each its own kernel(s)
                        c = overlaps.corners_batched(bx)
                        _, giou = overlaps.iou3d_batched(c, method="generalized", want_bev=True, out=buf)   # both outputs, as the reference computes them
                        ov = giou.add_(1.0).mul_(0.5)                      # 0.5 * (1 + giou), rpn_3d.py:781 (stock torch, in place)
                        prob = G.differentiable_nms_batched(sc, ov)[0]
                    elif two_calls:
                        ov = overlaps.iou_batched(bx, out=buf) if dim == 2 else overlaps.iou3d_batched(bx, from_params=True, nms_overlap=True, out=buf, nms_threshold=thr)
                        prob = G.differentiable_nms_batched(sc, ov)[0]
                    elif dim == 2:
                        prob = G.differentiable_nms_with_iou2d_batched(sc, bx, iou_out=buf)[0]
                    else:
                        prob = G.differentiable_nms_with_iou3d_batched(sc, bx, iou_out=buf)[0]
                    sc.grad = None
                    torch.autograd.backward(prob, ww)
                for _ in range(10):
                    one()
                torch.cuda.synchronize()
                # (three windows of `steps` steps, the MEDIAN reported and all three kept in `windows_ms`: a host hiccup inside a 60-step window of a
                # 0.19-ms step once read 0.307 ms; the headline's own figure stays the contract's single first window)
                wins = [gdist.timed_steps(one, steps, warmup if i == 0 else 0, torch.cuda.synchronize) for i in range(3)]
                dts = sorted(wins)[1]
                # roofline brief of this shape's HBM-bound launches: the same events the headline's roofline is made of (slot 0 = the launch
                # that writes the matrix, slot 1 = bitmask_kernel, the one full read of the matrix-in layer), over kr more steps
                kr = max(5, min(steps, 20))
                collect(0); collect(1)
                check(lib.gnms_profile_events(1), "profile_events")
                # (small shapes are host-bound: on an idle GPU the start marker of a bracket runs when it is enqueued and the launch behind it
                # arrives microseconds later, so the bracket would time the host.  A spin kernel in front lets the host run kr steps ahead:
                # markers and launches then execute back to back)
                # (the spin lasts at least three times what the host needed for kr eager steps, at 2.5 cycles per ns: on the pool's slowest
                # hosts the fixed spin of round 4 ended before the steps were enqueued and the bracket of B = 1 read the host, 0.088 ms for 38.7 us)
                spin = int(4e6) + int(2.5e5) * kr
                if dts / steps < 1e-4:                      # (only the shapes a host can bind: behind 8 ms of spinning the 2-ms steps ran 8 % slower)
                    spin = max(spin, min(int(3 * kr * (dts / steps) * 2.5e9), int(2e7)))
                torch.cuda._sleep(spin)
                for _ in range(kr):
                    one()
                torch.cuda.synchronize()
                check(lib.gnms_profile_events(0), "profile_events")
                (msw, nw_), (msr, nr_) = collect(0), collect(1)
                # the same kr steps once more behind a spin kernel, no per-launch events: what the GPU alone takes per step when the host has
                # run ahead (the eager figure below includes the host's launch cost, which is all of it at the small shapes and differs
                # between boxes by 2x)
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                if dts / steps < 1e-4:                      # (a GPU-bound shape needs no head start, and the spin would cost it clock)
                    torch.cuda._sleep(spin)
                ev0.record()
                for _ in range(kr):
                    one()
                ev1.record()
                torch.cuda.synchronize()
                dev_ms = ev0.elapsed_time(ev1) / kr
                # host-bound shapes: the same step as the C-ABI call sequence a compiled host makes (gnms_forward_with_iou2d/3d + gnms_backward
                # on fixed buffers through ctypes, eager, no autograd engine) -- what of the eager figure is the library's and what PyTorch's
                cabi_ms = None
                if dts / steps < 1e-4 and not (two_calls or ref_3d):
                    ws_c = torch.empty((lib.gnms_workspace_bytes(b_, n_, ctypes.byref(P)),), dtype=torch.uint8, device=dev)
                    prob_c, grad_c, s_c = torch.empty((b_, n_), device=dev), torch.empty((b_, n_), device=dev), sc.detach()
                    entry = lib.gnms_forward_with_iou2d if dim == 2 else lib.gnms_forward_with_iou3d

                    def one_c():
                        st_["i"] = (st_["i"] + 1) % nb_
                        buf = bufs[st_["i"]]
                        sp = stream_ptr(dev)
                        check(entry(ptr(bx), ptr(s_c), b_, n_, n_, None, ctypes.byref(P), ptr(buf), ptr(prob_c), None, None, None, None, None, ptr(ws_c),
                                    ws_c.numel(), sp), "fwd")
                        check(lib.gnms_backward(ptr(ww), ptr(s_c), ptr(buf), b_, n_, n_, None, ctypes.byref(P), ptr(grad_c), None, ptr(ws_c), ws_c.numel(), sp), "bwd")
                    for _ in range(10):
                        one_c()
                    torch.cuda.synchronize()
                    cabi_ms = gdist.timed_steps(one_c, steps, warmup, torch.cuda.synchronize) / steps * 1e3
                    del ws_c
                del bufs
                bytes_w = b_ * (4.0 * n_ * n_ + (16.0 if dim == 2 else 28.0) * n_)
                bytes_r = b_ * (4.0 * n_ * n_ + 16.0 * n_)
                wn = (lib.gnms_profile_write_kernel_name(dim, b_, n_).decode() if not (two_calls or ref_3d) else
                      ("iou2d_self_kernel" if dim == 2 and not ref_3d else "iou3d_kernel"))

                def brief(ms_sum, launches, nbytes, kname):
                    if not launches or ms_sum <= 0:
                        return None
                    per = ms_sum / kr
                    return {"kernel": kname, "kernel_ms": round(per, 4), "launches_per_step": round(launches / kr, 2), "algorithmic_bytes": round(nbytes),
                            "achieved": round(nbytes / (per * 1e-3) / 1e9, 1), "unit": "GB/s", "frac": round(nbytes / (per * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                res = {"value": round(b_ * n_ * steps / dts, 1), "unit": "boxes/s", "ms_per_step": round(dts / steps * 1e3, 4), "steps": steps,
                       "windows_ms": [round(w / steps * 1e3, 4) for w in wins], "device_ms_per_step": round(dev_ms, 4),
                       "workload": "%d images x %d %s %dD boxes%s" % (b_, n_, args.kind, dim, ", reference call sequence" if (two_calls or ref_3d) else ""),
                       "whole_step_frac": round((bytes_w + (bytes_r if (two_calls or ref_3d) else 0.0)) / (dts / steps) / 1e9 / HBM_PEAK_GBS, 4),
                       "roofline": brief(msw, nw_, bytes_w * (2.0 if ref_3d else 1.0), wn)}
                if two_calls or ref_3d:
                    res["roofline_matrix_in"] = brief(msr, nr_, bytes_r, "one_launch_kernel" if n_ <= 1024 and os.environ.get("GNMS_ONE_LAUNCH", "1") != "0" else "bitmask_kernel")
                if cabi_ms is not None:
                    res["c_abi_ms_per_step"] = round(cabi_ms, 4)
                return res

            # (60 steps where a step is a fraction of a millisecond: at 20 the 3D step read 0.203-0.207 ms where 100 steps give 0.185-0.188)
            for key, kw in (("two_calls", dict(dim=2, b_=B, n_=N, two_calls=True, steps=40)),
                            ("two_calls_3d", dict(dim=3, b_=B, n_=N, ref_3d=True)),
                            ("dim3_N4096", dict(dim=3, b_=B, n_=4096, steps=60, warmup=10)),
                            ("dim3_N16384", dict(dim=3, b_=B, n_=16384, steps=10)),
                            ("N256", dict(dim=2, b_=B, n_=256, steps=60, warmup=10)), ("N1024", dict(dim=2, b_=B, n_=1024, steps=60, warmup=10)),
                            ("N16384", dict(dim=2, b_=B, n_=16384, steps=10)),
                            ("B1_N4096", dict(dim=2, b_=1, n_=4096, steps=60, warmup=10)), ("B4_N4096", dict(dim=2, b_=4, n_=4096, steps=60, warmup=10))):
                try:
                    out[key] = shape(**kw)
                except Exception as e:
                    out[key] = {"error": str(e)[:300]}
                torch.cuda.empty_cache()
            out["two_calls_other_kind"] = None
            try:
                keep = args.kind
                args.kind = "uniform" if keep == "clustered" else "clustered"
                out["two_calls_other_kind"] = shape(dim=2, b_=B, n_=N, two_calls=True)
            except Exception as e:
                out["two_calls_other_kind"] = {"error": str(e)[:300]}
            finally:
                args.kind = keep

        if world == 1 and not args.no_extras and not args.graph and not args.two_calls and args.dim == 2 and not args.sorted_scores:
            # (round 6) the reference's OWN operating points, which it never exceeds: differentiable_nms on <= 500 boxes of one image, two images
            # per batch (lib/loss/rpn_3d.py:732,772-793; scripts/config/groumd_nms.py:116), gpu_nms on <= nms_topN_pre = 3000 boxes
            # (lib/rpn_util.py:1285-1334), the ablations' modes at N = 500 (scripts/config/: group_boxes / mask_group_boxes = False), and the
            # soft sort (lib/groomed_nms.py:131-165, test-only) -- the one MFMA kernel of the path -- against torch.matmul on this box
            def timed_us(fn, n, warm):
                for _ in range(warm):
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / n * 1e6

            def ref_call(lazy, **kw):
                bx_np, sc_np = synthetic.batch_2d(1000, 2, 500, args.kind)
                bxs = [torch.from_numpy(bx_np[i]).to(dev) for i in range(2)]
                scs = [torch.from_numpy(sc_np[i]).to(dev).requires_grad_(True) for i in range(2)]
                from groomed_nms_amd import groomed_nms as GN
                keep = GN.LAZY_INDEX_LISTS
                GN.LAZY_INDEX_LISTS = lazy

                def fwd():
                    for i in range(2):                                       # the per-image loop of rpn_3d.py:772-793
                        ov = overlaps.iou(bxs[i], bxs[i])
                        G.differentiable_nms(scs[i], ov, **kw)

                def fwd_bwd():
                    tot = None
                    for i in range(2):
                        ov = overlaps.iou(bxs[i], bxs[i])
                        pr = G.differentiable_nms(scs[i], ov, **kw)[2]
                        tot = pr.sum() if tot is None else tot + pr.sum()
                    scs[0].grad = scs[1].grad = None
                    tot.backward()
                try:
                    return {"us_forward": round(timed_us(fwd, 300, 30), 1), "us_forward_backward": round(timed_us(fwd_bwd, 200, 20), 1)}
                finally:
                    GN.LAZY_INDEX_LISTS = keep
            try:
                out["ref_call_N500_B2"] = {"workload": "per image of 2: iou(boxes, boxes) + differentiable_nms(scores, iou) on 500 %s boxes, GPU tensors "
                                                       "(lib/loss/rpn_3d.py:772-793); host included" % args.kind,
                                           "index_tensors": ref_call(False), "lazy_index_lists": ref_call(True),
                                           "modes_us_forward_backward": {"grouped_unmasked": ref_call(False, mask_group_boxes=False)["us_forward_backward"],
                                                                         "ungrouped": ref_call(False, group_boxes=False)["us_forward_backward"],
                                                                         "sigmoidal": ref_call(False, pruning_method="sigmoidal", temperature=0.1)["us_forward_backward"]}}
            except Exception as e:
                out["ref_call_N500_B2"] = {"error": str(e)[:300]}
            try:
                from groomed_nms_amd.nms import gpu_nms
                bx_np, sc_np = synthetic.batch_2d(1000, 1, 3000, args.kind)
                dets = np.concatenate([bx_np[0], sc_np[0][:, None]], axis=1).astype(np.float32)
                dets = dets[np.argsort(-dets[:, 4], kind="stable")]
                us = timed_us(lambda: gpu_nms(dets, 0.4, device_id=0), 200, 20)
                shuffled = dets[np.random.default_rng(3).permutation(len(dets))]
                us_sh = timed_us(lambda: gpu_nms(shuffled, 0.4, device_id=0), 100, 10)
                out["gpu_nms_N3000"] = {"workload": "gpu_nms(dets[3000, 5] on the host, 0.4) -> keep (lib/rpn_util.py:1285-1334; the C symbol _nms, both PCIe "
                                                    "directions included); boxes sorted by score as the call site delivers them (the wrapper then skips its "
                                                    "argsort + gather), and the same boxes shuffled",
                                        "us_per_call": round(us, 1), "us_per_call_unsorted_input": round(us_sh, 1), "kept": int(len(gpu_nms(dets, 0.4, device_id=0)))}
            except Exception as e:
                out["gpu_nms_N3000"] = {"error": str(e)[:300]}
            try:
                n_ss = 4096
                bx_np, sc_np = synthetic.batch_2d(1000, 1, n_ss, args.kind)
                sc_t = torch.from_numpy(sc_np[0]).to(dev)
                ov_t = overlaps.iou(torch.from_numpy(bx_np[0]).to(dev), torch.from_numpy(bx_np[0]).to(dev))
                ms_ss = timed_us(lambda: G.soft_sort(sc_t, ov_t, temperature=1.0), 20, 3) / 1e3
                a_t = torch.rand((n_ss, n_ss), device=dev)
                ms_mm = timed_us(lambda: torch.matmul(a_t, ov_t), 20, 3) / 1e3
                flops = 2.0 * n_ss ** 3
                peak_tf = 157.3                                              # fp32 matrix peak, MI355X_MICROARCH.md
                out["soft_sort_N4096"] = {"workload": "soft_sort(scores[4096], iou[4096, 4096]): P = softmax rows, P @ iou on v_mfma_f32_32x32x2_f32 "
                                                      "(lib/groomed_nms.py:131-165)", "ms_per_call": round(ms_ss, 4),
                                          "roofline": {"bound": "mfma", "achieved": round(flops / (ms_ss * 1e-3) / 1e12, 1), "peak": peak_tf, "unit": "TFLOP/s",
                                                       "frac": round(flops / (ms_ss * 1e-3) / 1e12 / peak_tf, 4),
                                                       "note": "whole call (row kernels + GEMM) over the GEMM's flops"},
                                          "torch_matmul_same_box": {"ms": round(ms_mm, 4), "TFLOP/s": round(flops / (ms_mm * 1e-3) / 1e12, 1)}}
            except Exception as e:
                out["soft_sort_N4096"] = {"error": str(e)[:300]}

        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle as O
            # the GPU results of the timed entry, for the parity figure BASELINE.json's metric asks for (max |dscore| vs the CPU path)
            s_par = scores.detach().clone().requires_grad_(True)
            if args.two_calls:
                g_out = G.differentiable_nms_batched(s_par, build_overlaps(iou_bufs[0]))
            elif args.dim == 2:
                g_out = G.differentiable_nms_with_iou2d_batched(s_par, boxes)
            else:
                g_out = G.differentiable_nms_with_iou3d_batched(s_par, boxes)
            torch.autograd.backward(g_out[0], w)
            g_prob, g_valid, g_nvalid, g_grad = g_out[0].detach().cpu().numpy(), g_out[2].cpu().numpy(), g_out[4].cpu().numpy(), s_par.grad.cpu().numpy()
            d_prob = d_grad = 0.0
            sets_equal, checked = True, 0
            t0 = time.perf_counter()
            k = 0
            while k < min(4, B) or (time.perf_counter() - t0) < args.cpu_seconds:      # bounded sample: whole images
                b = k % B
                if args.dim == 2:
                    m = O.iou2d(boxes_np[b], boxes_np[b])
                else:
                    c = O.corners_of_cuboid(boxes_np[b])
                    m = (np.float32(0.5) * (np.float32(1.0) + O.iou3d_approximate(c, c, generalized=True)[1])).astype(np.float32)
                res = O.differentiable_nms(scores_np[b], m, grad_prob=w_np)
                if k < B:                                                       # first pass over an image: compare (outside the timed sum)
                    tp = time.perf_counter()
                    d_prob = max(d_prob, float(np.abs(g_prob[b] - res["prob"]).max()))
                    d_grad = max(d_grad, float(np.abs(g_grad[b] - res["grad_scores"]).max()))
                    sets_equal &= set(g_valid[b, :g_nvalid[b]].tolist()) == set(res["valid"].tolist())
                    checked += 1
                    t0 += time.perf_counter() - tp
                k += 1
                if k >= 4096:
                    break
            tc = time.perf_counter() - t0
            out["parity"] = {"max_abs_dscore": d_prob, "max_abs_dgrad_scores": d_grad, "valid_index_sets_equal": bool(sets_equal),
                             "images_checked": checked, "against": "oracle/ (CPU restatement of lib/groomed_nms.py + lib/core.py), tolerance 1e-4"}
            out["cpu_baseline"] = {"value": round(k * N / tc, 1), "unit": "boxes/s", "cores": 1, "kind": "port",
                                   "sample": "%d image passes over rank 0's batch of %d images (N=%d) in %.1f s: oracle/gnms_oracle.c overlap matrix + "
                                             "nms fwd+bwd, single thread" % (k, B, N, tc)}
        print(json.dumps(out), flush=True)
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        t_b = time.perf_counter()
        dist.barrier()
        # (what the ranks other than 0 spend here is rank 0's post-region work: kept short at N > 1, asserted by test_n_rank_launcher_end_to_end)
        print("[bench] rank %d waited %.2f s in the final barrier" % (rank, time.perf_counter() - t_b), file=sys.stderr, flush=True)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
