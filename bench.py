#!/usr/bin/env python3
"""bench.py -- GrooMeD-NMS fwd+bwd boxes/sec on MI355X (BASELINE.json metric).

A step = one pass of the hot path over one batch of synthetic proposals already resident in HBM:
    pairwise IoU matrix  ->  GrooMeD-NMS forward  ->  backward w.r.t. scores
as lib/loss/rpn_3d.py:772-791 runs it.  Default (2D): ONE library call builds the matrix and runs the layer
(gnms_forward_with_iou2d: the matrix is an output, the threshold bits come straight from the boxes), then gnms_backward;
--two-calls keeps gnms_iou2d and the matrix-in layer gnms_forward apart; --dim 3 uses gnms_forward_with_iou3d.
Workload at N GPUs (weak scaling): every rank owns `--batch` images x `--boxes` boxes; images are independent
units, so there is no data-path collective (DESIGN.md "multi-GPU").

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...

Prints ONE JSON line on rank 0 (see the driver contract).  Extra objects:
  roofline            the dominant kernel of the timed step: algorithmic bytes / HIP-event time against the 8 TB/s HBM peak
                      (default: the IoU write kernel; with --two-calls the bit-matrix kernel, the one full read of the matrix)
  roofline_matrix_in  / roofline_iou: the other of the two
  fused_from_boxes    the matrix-free entry (never part of `value`)
  cpu_baseline        the CPU oracle (a C port of the reference algorithm, single thread) timed on this host, N=1 only
  parity              (with cpu_baseline) max |d prob| and max |d grad_scores| of the timed entry against that oracle on every image of
                      rank 0's batch, and whether the valid-index sets agree -- the "max-|dscore| vs ref" half of BASELINE.json's metric
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--boxes", type=int, default=4096, help="boxes per image (BASELINE metric: N=4096/img)")
    ap.add_argument("--batch", type=int, default=8, help="images per GPU")
    ap.add_argument("--kind", default="clustered", choices=["uniform", "clustered"])
    ap.add_argument("--dim", type=int, default=2, choices=[2, 3], help="2: lib/core.py iou; 3: 0.5*(1+GIoU3D) of the corner AABBs from (x,y,z,w,h,l,ry)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--two-calls", action="store_true", help="gnms_iou2d and gnms_forward as two library calls (no fused score sort)")
    ap.add_argument("--graph", action="store_true",
                    help="capture one step (IoU + forward + backward through the C ABI, preallocated buffers) in a HIP graph and replay "
                         "it: removes the per-launch host cost that bounds small problems (N <= 1024)")
    ap.add_argument("--sorted-scores", action="store_true",
                    help="feed scores already sorted by descending value, as both reference call sites do (lib/loss/rpn_3d.py:731-737, "
                         "lib/rpn_util.py:1258-1266): the bit-matrix kernel then reads only the reachable half of the matrix")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU baseline: keep processing images of the batch for about this long")
    return ap.parse_args()


def event_time_ms(fn, iters, stream):
    """Average duration of fn() over `iters` launches, HIP events on the stream the kernels run on."""
    start = torch.cuda.Event(enable_timing=True)
    end = torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    start.record(stream)
    for _ in range(iters):
        fn()
    end.record(stream)
    end.synchronize()
    return start.elapsed_time(end) / iters


def main():
    args = parse()
    import groomed_nms_amd as G
    from groomed_nms_amd import overlaps, synthetic, _lib, dist as gdist
    world, rank, local_rank = gdist.env_world()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    gdist.init(backend="nccl")                     # RCCL; no-op at world size 1
    lib = _lib.load()
    B, N = args.batch, args.boxes

    # every rank owns its own `B` images (weak scaling): rank r draws the images [r*B, (r+1)*B) of the global batch
    if args.dim == 2:
        boxes_np, scores_np = synthetic.batch_2d(1000 + rank, B, N, args.kind)
    else:
        boxes_np, scores_np = synthetic.batch_3d(1000 + rank, B, N, clustered=(args.kind == "clustered"))
    if args.sorted_scores:
        o = np.argsort(-scores_np, axis=1, kind="stable")
        scores_np = np.take_along_axis(scores_np, o, axis=1)
        boxes_np = np.take_along_axis(boxes_np, o[:, :, None], axis=1)
    boxes = torch.from_numpy(np.ascontiguousarray(boxes_np)).to(dev)
    scores = torch.from_numpy(np.ascontiguousarray(scores_np)).to(dev).requires_grad_(True)
    w = torch.linspace(-1.0, 2.0, N, device=dev).repeat(B, 1).contiguous()
    iou_buf = torch.empty((B, N, N), dtype=torch.float32, device=dev)

    def build_overlaps():
        if args.dim == 2:
            return overlaps.iou_batched(boxes, out=iou_buf)
        return overlaps.iou3d_batched(boxes, from_params=True, nms_overlap=True, out=iou_buf)

    def step():
        if args.dim == 2 and not args.two_calls:
            # IoU matrix + forward as ONE library call (gnms_forward_with_iou2d): same kernels' work, the score sort rides in the IoU launch
            prob = G.differentiable_nms_with_iou2d_batched(scores, boxes, iou_out=iou_buf)[0]
        elif args.dim == 3 and not args.two_calls:
            prob = G.differentiable_nms_with_iou3d_batched(scores, boxes, iou_out=iou_buf)[0]      # gnms_forward_with_iou3d
        else:
            prob = G.differentiable_nms_batched(scores, build_overlaps())[0]
        scores.grad = None
        torch.autograd.backward(prob, w)          # dL/dprob = w
        return prob

    if args.graph:
        import ctypes
        from groomed_nms_amd._lib import GnmsParams, ptr, stream_ptr, check
        Pg = GnmsParams()
        lib.gnms_default_params(ctypes.byref(Pg))
        ws_g = torch.empty((lib.gnms_workspace_bytes(B, N, ctypes.byref(Pg)),), dtype=torch.uint8, device=dev)
        prob_g = torch.empty((B, N), dtype=torch.float32, device=dev)
        grad_g = torch.empty((B, N), dtype=torch.float32, device=dev)
        s_det = scores.detach()

        def raw_step():
            sp = stream_ptr(dev)
            if args.dim == 2 and not args.two_calls:
                check(lib.gnms_forward_with_iou2d(ptr(boxes), ptr(s_det), B, N, N, None, ctypes.byref(Pg), ptr(iou_buf), ptr(prob_g), None, None,
                                                  None, None, None, ptr(ws_g), ws_g.numel(), sp), "fwd_with_iou2d")
            elif not args.two_calls:
                check(lib.gnms_forward_with_iou3d(ptr(boxes), ptr(s_det), B, N, N, None, ctypes.byref(Pg), ptr(iou_buf), ptr(prob_g), None, None,
                                                  None, None, None, ptr(ws_g), ws_g.numel(), sp), "fwd_with_iou3d")
            else:
                if args.dim == 2:
                    check(lib.gnms_iou2d(ptr(boxes), ptr(boxes), B, N, N, ptr(iou_buf), N, sp), "iou2d")
                else:
                    check(lib.gnms_iou3d_from_params(ptr(boxes), ptr(boxes), B, N, N, 2, None, ptr(iou_buf), N, sp), "iou3d")
                check(lib.gnms_forward(ptr(s_det), ptr(iou_buf), B, N, N, None, ctypes.byref(Pg), ptr(prob_g), None, None, None, None, None,
                                       ptr(ws_g), ws_g.numel(), sp), "fwd")
            check(lib.gnms_backward(ptr(w), ptr(s_det), ptr(iou_buf), B, N, N, None, ctypes.byref(Pg), ptr(grad_g), None, ptr(ws_g),
                                    ws_g.numel(), sp), "bwd")

        raw_step()
        torch.cuda.synchronize()
        hip_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(hip_graph):
            raw_step()
        step = hip_graph.replay       # noqa: F811  one replay = one full step
    dt = gdist.timed_steps(step, args.steps, args.warmup, torch.cuda.synchronize)

    # ---------------- per-kernel roofline (rank 0), HIP events on the launch stream -----------------
    out = None
    if rank == 0:
        import ctypes
        from groomed_nms_amd._lib import GnmsParams, ptr, stream_ptr, check
        stream = torch.cuda.current_stream(dev)
        per_box = 16.0 if args.dim == 2 else 28.0
        alg_bytes = B * (4.0 * N * N + 16.0 * N)          # NMS forward 4N^2 + 16N per image
        if args.sorted_scores:                            # only the columns a leader of the row block can sit in are needed
            alg_bytes = B * (sum(64 * min(N, 64 * (kb + 1)) for kb in range((N + 63) // 64)) * 4.0 + 16.0 * N)
        alg_bytes_iou = B * (4.0 * N * N + per_box * N)   # IoU-2D 4N^2 + 16N, IoU-3D (params) 4N^2 + 28N
        t_iou = event_time_ms(build_overlaps, 20, stream)
        P = GnmsParams()
        lib.gnms_default_params(ctypes.byref(P))
        ws = torch.empty((lib.gnms_workspace_bytes(B, N, ctypes.byref(P)),), dtype=torch.uint8, device=dev)
        prob = torch.empty((B, N), dtype=torch.float32, device=dev)
        s_det = scores.detach()
        check(lib.gnms_forward(ptr(s_det), ptr(iou_buf), B, N, N, None, ctypes.byref(P), ptr(prob), None, None, None, None, None,
                               ptr(ws), ws.numel(), stream_ptr(dev)), "gnms_forward")
        t_mask = event_time_ms(lambda: check(lib.gnms_profile_bitmask(ptr(iou_buf), B, N, N, None, P.nms_threshold, ptr(ws),
                                                                     ws.numel(), stream_ptr(dev)), "bitmask"), 20, stream)
        t_fwd = event_time_ms(lambda: check(lib.gnms_forward(ptr(s_det), ptr(iou_buf), B, N, N, None, ctypes.byref(P), ptr(prob), None,
                                                             None, None, None, None, ptr(ws), ws.numel(), stream_ptr(dev)), "fwd"), 20, stream)
        gs = torch.empty((B, N), dtype=torch.float32, device=dev)
        t_bwd = event_time_ms(lambda: check(lib.gnms_backward(ptr(w), ptr(s_det), ptr(iou_buf), B, N, N, None, ctypes.byref(P), ptr(gs),
                                                              None, ptr(ws), ws.numel(), stream_ptr(dev)), "bwd"), 20, stream)

        # HBM traffic per launch from the committed PMC passes (profiles/), valid only for the configuration they were taken on
        pmc = {}
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")) as f:
                ps = json.load(f)
            c = ps["config"]
            if (c["images_per_gpu"], c["boxes_per_image"], c["kind"]) == (B, N, args.kind) and args.dim == 2:
                pmc = {k: v["traffic_bytes_per_launch"] for k, v in ps["kernels"].items()}
        except (OSError, KeyError, ValueError):
            pmc = {}

        def roof(t_ms, nbytes, kname):
            ach = nbytes / (t_ms * 1e-3) / 1e9
            return {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": (round(pmc[kname]) if kname in pmc else None), "algorithmic_bytes": round(nbytes), "kernel_ms": round(t_ms, 4)}

        total_boxes = world * B * N * args.steps
        out = {
            "metric": "GrooMeD-NMS fwd+bwd boxes/sec (pairwise IoU + differentiable_nms forward + backward wrt scores)",
            "value": round(total_boxes / dt, 1),
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
                       "scores_presorted": bool(args.sorted_scores), "hip_graph_replay": bool(args.graph), "parallelism": "images sharded, dp%d" % world},
            "roofline": None,
            "phase_ms": {"overlaps": round(t_iou, 4), "nms_forward": round(t_fwd, 4), "nms_backward": round(t_bwd, 4)},
        }
        iou_name = "iou2d_kernel" if args.dim == 2 else "iou3d_nms_fast_kernel"
        one_call = not args.two_calls
        iou_note = (" (one full write of the NxN fp32 matrix; the same tile code runs as iou2d_sort_kernel inside gnms_forward_with_iou2d)"
                    if args.dim == 2 else " (one full write of the NxN fp32 matrix, + the per-box record kernels in front of it)")
        r_iou = dict(roof(t_iou, alg_bytes_iou, "iou2d_sort_kernel" if (one_call and "iou2d_sort_kernel" in pmc) else iou_name),
                     kernel=iou_name + iou_note)
        r_mask = dict(roof(t_mask, alg_bytes, "bitmask_kernel"), kernel="bitmask_kernel (gnms_forward: one full read of the NxN fp32 matrix)")
        if one_call:
            # the timed step hands the boxes over, so the layer never reads the matrix back: the matrix write is the dominant kernel
            out["roofline"], out["roofline_matrix_in"] = r_iou, r_mask
            entry = G.differentiable_nms_with_iou2d_batched if args.dim == 2 else G.differentiable_nms_with_iou3d_batched
            t_one = event_time_ms(lambda: entry(s_det, boxes, iou_out=iou_buf), 20, stream)
            out["phase_ms"] = {"overlaps_plus_nms_forward_one_call": round(t_one, 4), "nms_backward": round(t_bwd, 4),
                               "separately": {"overlaps": round(t_iou, 4), "nms_forward_matrix_in": round(t_fwd, 4)}}
        else:
            out["roofline"], out["roofline_iou"] = r_mask, r_iou
        if args.dim == 2:
            # ---- reported SEPARATELY (never part of `value`): the from-boxes path, same outputs bit for bit, no N x N matrix ----
            def fused_step():
                p = G.differentiable_nms_from_boxes_batched(scores, boxes)[0]
                scores.grad = None
                torch.autograd.backward(p, w)
            k_f = max(5, args.steps // 4)
            for _ in range(3):
                fused_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k_f):
                fused_step()
            torch.cuda.synchronize()
            t_fused = (time.perf_counter() - t0) / k_f
            prob_b = torch.empty((B, N), dtype=torch.float32, device=dev)
            check(lib.gnms_forward_from_boxes(ptr(boxes), ptr(s_det), B, N, None, ctypes.byref(P), ptr(prob_b), None, None, None, None, None,
                                              ptr(ws), ws.numel(), stream_ptr(dev)), "fwd_boxes")      # fills the x-order the kernel needs
            t_mb = event_time_ms(lambda: check(lib.gnms_profile_bitmask_boxes(ptr(boxes), B, N, None, P.nms_threshold, ptr(ws), ws.numel(),
                                                                             stream_ptr(dev)), "bitmask_boxes"), 20, stream)
            pairs = B * N * (N - 1) // 2            # decisions the layer needs: every (leader candidate, lower-ranked box) pair
            out["fused_from_boxes"] = {
                "value": round(B * N / t_fused, 1), "unit": "boxes/s (1 GPU, this rank)", "ms_per_step": round(t_fused * 1e3, 4), "steps": k_f,
                "note": "gnms_forward_from_boxes + gnms_backward_from_boxes: identical outputs (tests/test_gpu_parity.py::"
                        "test_from_boxes_path_is_bit_identical), the NxN matrix is never materialised; not comparable with `value`",
                "bitmask_boxes_kernel": {"bound": "fp32-vector, rows culled against the hull of x-sorted column tiles (data dependent)",
                                         "kernel_ms": round(t_mb, 4), "pair_decisions": pairs,
                                         "decisions_per_s": round(pairs / (t_mb * 1e-3), 1)}}
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle as O
            # the GPU results of the timed entry, for the parity figure BASELINE.json's metric asks for (max |dscore| vs the CPU path)
            entry = (G.differentiable_nms_with_iou2d_batched if args.dim == 2 else G.differentiable_nms_with_iou3d_batched) \
                if not args.two_calls else None
            s_par = scores.detach().clone().requires_grad_(True)
            g_out = entry(s_par, boxes) if entry else G.differentiable_nms_batched(s_par, build_overlaps())
            torch.autograd.backward(g_out[0], w)
            g_prob, g_valid, g_nvalid, g_grad = g_out[0].detach().cpu().numpy(), g_out[2].cpu().numpy(), g_out[4].cpu().numpy(), s_par.grad.cpu().numpy()
            d_prob = d_grad = 0.0
            sets_equal, checked = True, 0
            t0 = time.perf_counter()
            k = 0
            while k < 4 or (time.perf_counter() - t0) < args.cpu_seconds:      # bounded sample: whole images, >= 4 of them
                b = k % B
                if args.dim == 2:
                    m = O.iou2d(boxes_np[b], boxes_np[b])
                else:
                    c = O.corners_of_cuboid(boxes_np[b])
                    m = 0.5 * (1.0 + O.iou3d_approximate(c, c, generalized=True)[1])
                res = O.differentiable_nms(scores_np[b], m, grad_prob=np.linspace(-1, 2, N).astype(np.float32))
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
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
