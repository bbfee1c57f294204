#!/usr/bin/env python3
"""End-to-end harness for BASELINE.json configs C3 / C4 (bench scaffolding, not a deliverable kernel):

    random 3 x 512 x 1760 image  ->  DenseNet-121 RPN (stock torch.nn on ROCm: dilated denseblock4, transition3 pool removed,
    the heads of /root/reference/models/densenet121_3d_dilate_decomp_alpha.py:13-250)  ->  decode (gnms_bbox_transform_inv)
    ->  score top-K (gnms_select_topk)  ->  IoU matrix + GrooMeD-NMS (gnms_forward_with_iou2d)  ->  best box per ground truth
    (gnms_best_targets)  ->  after-NMS AP loss (gnms_aploss)  ->  backward into the backbone  ->  SGD step

    --mode infer   C3: B images, forward only through the NMS layer (default B = 1, K = 4096 proposals into the NMS)
    --mode train   C4: B images per GPU, forward + backward + optimizer step; with --gpus N one process per GPU under
                   torch.distributed (RCCL): DistributedDataParallel all-reduces the ~32 MB of backbone + head gradients over
                   xGMI, bucketed and overlapped with the backward pass.  The NMS layer itself needs no collective (images are
                   independent units).

torchvision is absent from this image and KITTI / pretrained weights are not available, so the network is defined here with the
same layer shapes and randomly initialised, inputs and ground truths are synthetic (`"data": "synthetic"`).
Prints one JSON line on rank 0.  python tools/e2e_bench.py --mode train --gpus 1 --steps 20
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


# ---------------------------------------------------------------- DenseNet-121 (torchvision layout, stock torch.nn) -----------------
class _DenseLayer(nn.Module):
    def __init__(self, cin, growth=32, bn_size=4, dilation=1):
        super().__init__()
        self.norm1, self.conv1 = nn.BatchNorm2d(cin), nn.Conv2d(cin, bn_size * growth, 1, bias=False)
        self.norm2 = nn.BatchNorm2d(bn_size * growth)
        self.conv2 = nn.Conv2d(bn_size * growth, growth, 3, padding=dilation, dilation=dilation, bias=False)

    def forward(self, x):
        y = self.conv1(F.relu(self.norm1(x)))
        return torch.cat([x, self.conv2(F.relu(self.norm2(y)))], 1)


def _block(cin, n, dilation=1):
    layers, c = [], cin
    for _ in range(n):
        layers.append(_DenseLayer(c, dilation=dilation))
        c += 32
    return nn.Sequential(*layers), c


def _transition(cin, pool=True):
    mods = [nn.BatchNorm2d(cin), nn.ReLU(inplace=True), nn.Conv2d(cin, cin // 2, 1, bias=False)]
    if pool:
        mods.append(nn.AvgPool2d(2, 2))
    return nn.Sequential(*mods), cin // 2


class DenseNet121RPN(nn.Module):
    """models/densenet121_3d_dilate_decomp_alpha.py: densenet121.features with transition3.pool deleted (:21) and denseblock4 dilated
    by 2 (:24-39) -> prop_feats 3x3 conv 512 (:46-49) -> 1x1 heads: cls (classes x anchors), 4 2D + 7 3D regression maps (:53-70)."""

    def __init__(self, num_anchors=36, num_classes=4):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2, 1))
        self.b1, c = _block(64, 6)
        self.t1, c = _transition(c)
        self.b2, c = _block(c, 12)
        self.t2, c = _transition(c)
        self.b3, c = _block(c, 24)
        self.t3, c = _transition(c, pool=False)
        self.b4, c = _block(c, 16, dilation=2)
        self.norm5 = nn.BatchNorm2d(c)
        self.prop_feats = nn.Sequential(nn.Conv2d(c, 512, 3, padding=1), nn.ReLU(inplace=True))
        self.num_anchors, self.num_classes = num_anchors, num_classes
        self.cls = nn.Conv2d(512, num_classes * num_anchors, 1)
        self.bbox2d = nn.Conv2d(512, 4 * num_anchors, 1)        # bbox_x, _y, _w, _h as one conv of the same total width
        self.bbox3d = nn.Conv2d(512, 7 * num_anchors, 1)        # x3d y3d z3d w3d h3d l3d + rotation

    def forward(self, x):
        f = self.norm5(self.b4(self.t3(self.b3(self.t2(self.b2(self.t1(self.b1(self.stem(x)))))))))
        f = self.prop_feats(F.relu(f))
        B, _, H, W = f.shape
        A = self.num_anchors
        cls = self.cls(f).view(B, self.num_classes, A * H, W)                                   # softmax over dim 1 (:72-75)
        prob = F.softmax(cls, dim=1).view(B, self.num_classes, A, H, W).permute(0, 3, 4, 2, 1).reshape(B, H * W * A, self.num_classes)
        d2 = self.bbox2d(f).view(B, 4, A, H, W).permute(0, 3, 4, 2, 1).reshape(B, H * W * A, 4)
        d3 = self.bbox3d(f).view(B, 7, A, H, W).permute(0, 3, 4, 2, 1).reshape(B, H * W * A, 7)
        return prob, d2, d3, (H, W)


def make_anchors(H, W, A, stride=16):
    """anchors laid out like lib/rpn_util.py locate_anchors: A template boxes at every feature-map cell, (y, x, a) order."""
    rng = np.random.default_rng(0)
    sizes = np.exp(np.linspace(np.log(24), np.log(384), 12))
    tmpl = np.array([[-w / 2, -w * r / 2, w / 2, w * r / 2] for w in sizes for r in (0.5, 1.0, 1.5)][:A], np.float32)
    ys, xs = np.meshgrid(np.arange(H) * stride + stride / 2, np.arange(W) * stride + stride / 2, indexing="ij")
    ctr = np.stack([xs, ys, xs, ys], -1).reshape(H * W, 1, 4).astype(np.float32)
    del rng
    return torch.from_numpy((ctr + tmpl[None]).reshape(-1, 4))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="train", choices=["train", "infer"])
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (train: 4 = C4's 32 over 8 GPUs; infer: 1)")
    ap.add_argument("--topk", type=int, default=4096, help="proposals per image into the NMS (C3/C4: ~4096; the reference trains with 500)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=1760)
    args = ap.parse_args()
    from groomed_nms_amd import dist as gdist
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        if torch.cuda.device_count() < args.gpus and not gdist.share_gpu():
            sys.exit("e2e_bench.py: --gpus %d requested, %d visible; refusing" % (args.gpus, torch.cuda.device_count()))
        sys.exit(gdist.relaunch_under_torchrun(args.gpus, os.path.abspath(__file__), sys.argv[1:]))
    import groomed_nms_amd as G
    from groomed_nms_amd import proposals as PR
    from groomed_nms_amd.aploss import ap_loss_batched
    world, rank, local_rank = gdist.init(backend="nccl")             # RCCL (GNMS_SHARE_GPU=1, debug: ranks share the visible GPUs over gloo)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    B = args.batch or (4 if args.mode == "train" else 1)
    K, M = args.topk, 8
    torch.manual_seed(1234 + rank)
    net = DenseNet121RPN().to(dev)
    n_params = sum(p.numel() for p in net.parameters())
    model = net
    if torch.distributed.is_initialized():
        model = nn.parallel.DistributedDataParallel(net, device_ids=[local_rank], bucket_cap_mb=25)     # RCCL all-reduce, overlapped with backward
    opt = torch.optim.SGD(net.parameters(), lr=1e-4, momentum=0.9)
    img = torch.randn((B, 3, args.height, args.width), device=dev)
    rng = np.random.default_rng(7 + rank)

    def cuboids(n):
        return np.stack([rng.uniform(-20, 20, n), rng.uniform(0.5, 2.5, n), rng.uniform(6, 50, n), rng.uniform(1.4, 2, n),
                         rng.uniform(1.3, 2, n), rng.uniform(3, 5, n), rng.uniform(-3.1, 3.1, n)], 1).astype(np.float32)
    gt3d = torch.from_numpy(np.stack([cuboids(M) for _ in range(B)])).to(dev)
    p2 = torch.tensor([[721.5, 0, 609.6, 44.9], [0, 721.5, 172.9, 0.22], [0, 0, 1, 0.0027], [0, 0, 0, 1]], device=dev).repeat(B, 1, 1)
    gt2d = PR.projected_boxes_2d(gt3d, p2, 1.0)
    anchors = None
    phase = {"backbone": 0.0, "nms_path": 0.0}

    def step(train):
        nonlocal anchors
        t0 = time.perf_counter()
        prob, d2, d3, (H, W) = model(img)
        if anchors is None:
            anchors = make_anchors(H, W, net.num_anchors).to(dev)
        scores = 1.0 - prob[:, :, 0]                                                    # foreground probability (lib/loss/rpn_3d.py:722-730)
        boxes = PR.bbox_transform_inv(anchors, d2.detach(), means=[0, 0, 0, 0], stds=[0.1, 0.1, 0.2, 0.2])
        A = scores.shape[1]
        # the K best-scoring of ALL anchors (lib/rpn_util.py:1258-1266; the loss ranks the anchors labelled foreground, rpn_3d.py:731)
        idx, num, s_sel, b_sel = PR.select_topk(scores.detach(), K, None, None, boxes)
        s_sel = torch.gather(scores, 1, idx.clamp(min=0)) * (idx >= 0)                   # differentiable gather of the same boxes
        out = G.differentiable_nms_with_iou2d_batched(s_sel, b_sel, counts=num, index_lists=not train)
        if not train:
            return out[4]
        p3 = torch.gather(d3.detach(), 1, idx.clamp(min=0).unsqueeze(-1).expand(-1, -1, 7))
        p3 = torch.cat([p3[..., :2] * 5, p3[..., 2:3].abs() * 20 + 6, p3[..., 3:6].abs() + 1.5, p3[..., 6:]], -1).contiguous()
        targets, _, _ = PR.best_targets(p3, b_sel, gt3d, gt2d, 0.0, num, None)
        # + a stand-in for the box-regression terms of the reference's loss (lib/loss/rpn_3d.py), so that every head receives a gradient
        loss = ap_loss_batched(out[0], targets, counts=num).mean() + 1e-3 * (d2.square().mean() + d3.square().mean())
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        del t0
        return loss.detach()

    train = args.mode == "train"
    ctx = torch.enable_grad() if train else torch.no_grad()
    model.train(train)
    with ctx:
        for _ in range(args.warmup):
            step(train)
        torch.cuda.synchronize()
        gdist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            last = step(train)
        torch.cuda.synchronize()
        gdist.barrier()
        torch.cuda.synchronize()
        dt = gdist.max_over_ranks(time.perf_counter() - t0)
        # the NMS path alone on the same proposals (decode -> top-K -> IoU + layer [-> targets -> AP loss -> backward]), for the split
        with torch.no_grad():
            prob, d2, d3, _ = model(img)
        scores = (1.0 - prob[:, :, 0]).detach().requires_grad_(train)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            boxes = PR.bbox_transform_inv(anchors, d2, means=[0, 0, 0, 0], stds=[0.1, 0.1, 0.2, 0.2])
            idx, num, s_sel, b_sel = PR.select_topk(scores.detach(), K, None, None, boxes)
            s_sel = torch.gather(scores, 1, idx.clamp(min=0)) * (idx >= 0)
            out = G.differentiable_nms_with_iou2d_batched(s_sel, b_sel, counts=num, index_lists=not train)
            if train:
                scores.grad = None
                out[0].sum().backward()
        torch.cuda.synchronize()
        phase["nms_path"] = (time.perf_counter() - t1) / args.steps
        if os.environ.get("GNMS_E2E_DUMP"):                         # (developer: the proposals the layer saw, for offline analysis)
            np.savez(os.environ["GNMS_E2E_DUMP"], scores=s_sel.detach().cpu().numpy(), boxes=b_sel.cpu().numpy(), num=num.cpu().numpy())
    if rank == 0:
        imgs = world * B * args.steps
        print(json.dumps({
            "harness": "C4 training step" if train else "C3 inference",
            "value": round(imgs / dt, 2), "unit": "images/s", "n_gpus": world, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "boxes_into_nms_per_s": round(imgs * K / dt, 1),
            "config": {"model": "DenseNet-121 RPN-3D (random init, %.1f M parameters, fp32)" % (n_params / 1e6), "image": [3, args.height, args.width],
                       "images_per_gpu": B, "anchors_per_image": int(anchors.shape[0]), "proposals_into_nms": K,
                       "parallelism": "dp%d%s" % (world, " (DDP, RCCL all-reduce of %.0f MB fp32 gradients per step)" % (n_params * 4 / 1e6) if world > 1 or torch.distributed.is_initialized() else "")},
            "split_ms": {"whole_step": round(dt / args.steps * 1e3, 3), "decode_topk_iou_nms%s" % ("_bwd" if train else ""): round(phase["nms_path"] * 1e3, 3)},
            "data": "synthetic", "last_loss_or_nvalid": [float(x) for x in last.flatten().tolist()][:4]}))
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
