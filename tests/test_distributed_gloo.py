"""world_size-2 gloo test (CPU) of the multi-GPU path: images are sharded contiguously across ranks, no rank
needs another rank's data, the sharded results concatenate to the unsharded result, and the timing contract
(barrier + MAX over ranks) behaves.  The per-shard arithmetic runs on the CPU oracle here -- the layer under
test on the GPU box is the HIP one; what this test pins is the sharding/collective logic around it."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total_images, n, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from groomed_nms_amd import dist as gdist, synthetic
    from oracle import oracle as O
    w, r, _ = gdist.init(backend="gloo")
    assert (w, r) == (world, rank) and dist.get_backend() == "gloo"
    lo, hi = gdist.shard_range(total_images, rank, world)
    boxes, scores = synthetic.batch_2d(42, total_images, n, "clustered", per=16)     # same global batch on every rank
    probs = []
    calls = {"n": 0}

    def step():
        calls["n"] += 1
        probs.clear()
        for b in range(lo, hi):                                                       # only this rank's images
            m = O.iou2d(boxes[b], boxes[b])
            probs.append(O.differentiable_nms(scores[b], m)["prob"])

    hb = gdist.StepHeartbeat()                                                          # the per-step 4-byte all-reduce
    elapsed = gdist.timed_steps(step, steps=2, warmup=1, sync=lambda: None, heartbeat=hb)
    # timed_steps ran hb.check(): every one of the 3 per-step all-reduces summed to world_size on every rank, then the slots were reset
    assert calls["n"] == 3 and hb.on and hb.steps == 3 and hb.used == 0 and bool((hb.buf == 1).all())
    # ... and that check can fail: a step whose reduction did not see every rank (here: a slot tampered with) is reported
    hb2 = gdist.StepHeartbeat()
    hb2.beat()
    hb2.works[-1].wait()
    assert int(hb2.buf[0]) == world
    hb2.buf[0] = world - 1
    try:
        hb2._verify_slots()
        raise AssertionError("a short per-step reduction went unnoticed")
    except RuntimeError as e:
        assert "did not sum to world_size" in str(e)
    # wrap-around of the slot vector: verified and reused
    hb3 = gdist.StepHeartbeat()
    hb3.CAPACITY = 4
    for _ in range(11):
        hb3.beat()
    hb3.steps = 11
    hb3.check()
    # MAX over ranks: every rank reports the same, largest, time
    gathered = [None] * world
    dist.all_gather_object(gathered, elapsed)
    assert len(set(gathered)) == 1
    assert gdist.max_over_ranks(float(rank)) == float(world - 1)
    assert gdist.sum_over_ranks(float(hi - lo)) == float(total_images)              # shards partition the batch
    np.save(os.path.join(out_dir, "prob_rank%d.npy" % rank), np.stack(probs) if probs else np.zeros((0, n), np.float32))
    gdist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    from groomed_nms_amd.dist import shard_range
    for total in (0, 1, 7, 8, 32, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_sharding(tmp_path):
    from groomed_nms_amd import synthetic
    from oracle import oracle as O
    O.build()
    world, total, n = 2, 5, 96
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total, n, str(tmp_path)), nprocs=world, join=True)
    got = np.concatenate([np.load(os.path.join(tmp_path, "prob_rank%d.npy" % r)) for r in range(world)])
    boxes, scores = synthetic.batch_2d(42, total, n, "clustered", per=16)
    ref = np.stack([O.differentiable_nms(scores[b], O.iou2d(boxes[b], boxes[b]))["prob"] for b in range(total)])
    assert got.shape == ref.shape and np.array_equal(got, ref)
