#!/usr/bin/env python3
"""bench.py -- GrooMeD-NMS fwd+bwd boxes/sec on MI355X (BASELINE.json metric).

A step = one pass of the hot path over one batch of synthetic proposals already resident in HBM:
    pairwise 2D IoU matrix (gnms_iou2d)  ->  GrooMeD-NMS forward (gnms_forward)  ->  backward w.r.t. scores (gnms_backward)
Workload at N GPUs (weak scaling): every rank owns `--batch` images x `--boxes` boxes; images are independent
units, so there is no data-path collective (DESIGN.md "multi-GPU").

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...

Prints ONE JSON line on rank 0 (see the driver contract).  Extra objects:
  roofline      the dominant kernel (threshold bit-matrix kernel = the one full read of the N x N fp32 matrix),
                algorithmic bytes / HIP-event time, against the 8 TB/s HBM peak
  roofline_iou  same for the IoU kernel (the one full write of the matrix)
  cpu_baseline  the CPU oracle (a C port of the reference algorithm, single thread) timed on this host, N=1 only
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
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--boxes", type=int, default=4096, help="boxes per image (BASELINE metric: N=4096/img)")
    ap.add_argument("--batch", type=int, default=8, help="images per GPU")
    ap.add_argument("--kind", default="clustered", choices=["uniform", "clustered"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-images", type=int, default=4, help="images the CPU baseline processes")
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
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import groomed_nms_amd as G
    from groomed_nms_amd import overlaps, synthetic, _lib
    lib = _lib.load()
    B, N = args.batch, args.boxes

    boxes_np, scores_np = synthetic.batch_2d(1000 + rank, B, N, args.kind)
    boxes = torch.from_numpy(boxes_np).to(dev)
    scores = torch.from_numpy(scores_np).to(dev).requires_grad_(True)
    w = torch.linspace(-1.0, 2.0, N, device=dev).repeat(B, 1).contiguous()
    iou_buf = torch.empty((B, N, N), dtype=torch.float32, device=dev)

    def step():
        iou = overlaps.iou_batched(boxes, out=iou_buf)
        prob, order, valid, invalid, nv, ni = G.differentiable_nms_batched(scores, iou)
        scores.grad = None
        torch.autograd.backward(prob, w)          # dL/dprob = w
        return prob

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---------------- per-kernel roofline (rank 0), HIP events on the launch stream -----------------
    out = None
    if rank == 0:
        import ctypes
        from groomed_nms_amd._lib import GnmsParams, ptr, stream_ptr, check
        stream = torch.cuda.current_stream(dev)
        alg_bytes = B * (4.0 * N * N + 16.0 * N)          # 4N^2 + 16N per image, for either kernel
        t_iou = event_time_ms(lambda: overlaps.iou_batched(boxes, out=iou_buf), 20, stream)
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

        def roof(t_ms):
            ach = alg_bytes / (t_ms * 1e-3) / 1e9
            return {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": None, "kernel_ms": round(t_ms, 4)}

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
            "config": {"workload": "%d images/GPU x %d %s 2D boxes/image, nms_threshold 0.4, linear pruning, grouped+masked, group_size 100"
                                   % (B, N, args.kind), "boxes_per_image": N, "images_per_gpu": B, "parallelism": "images sharded, dp%d" % world},
            "roofline": dict(roof(t_mask), kernel="bitmask_kernel (gnms_forward: one full read of the NxN fp32 matrix)"),
            "roofline_iou": dict(roof(t_iou), kernel="iou2d_kernel (gnms_iou2d: one full write of the NxN fp32 matrix)"),
            "phase_ms": {"iou2d": round(t_iou, 4), "nms_forward": round(t_fwd, 4), "nms_backward": round(t_bwd, 4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle as O
            k = min(args.cpu_images, B)
            t0 = time.perf_counter()
            for b in range(k):
                m = O.iou2d(boxes_np[b], boxes_np[b])
                O.differentiable_nms(scores_np[b], m, grad_prob=np.linspace(-1, 2, N).astype(np.float32))
            tc = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": round(k * N / tc, 1), "unit": "boxes/s", "cores": 1, "kind": "port",
                                   "sample": "%d of the %d images of rank 0's batch (N=%d), oracle/gnms_oracle.c: iou2d + nms fwd+bwd" % (k, B, N)}
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
