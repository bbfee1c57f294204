"""Multi-GPU plumbing for the GrooMeD-NMS path: one process per GPU, images sharded across ranks.

The reference is single-process nn.DataParallel (lib/core.py:68) and runs the NMS per image in a Python
loop on the gathered batch (lib/loss/rpn_3d.py:375).  Images are independent units, so here every rank
owns a contiguous slice of the batch and the NMS layer needs NO collective; the only collectives are the
timing barrier / max-reduce below (and, in a training step, DDP's gradient all-reduce of the backbone,
which is stock torch over RCCL).  backend "nccl" is RCCL on ROCm; the CPU tests use "gloo".
"""
import os
import time

import torch
import torch.distributed as dist


def share_gpu():
    """GNMS_SHARE_GPU=1 -- a DEBUG mode that proves the N-rank launch path on a box with fewer GPUs than ranks: every rank uses device
    (local_rank mod visible devices) and the collectives run over gloo (RCCL refuses two ranks on one device).  Timings taken this way
    say nothing about scaling; bench.py marks its line with "debug_shared_gpu"."""
    return os.environ.get("GNMS_SHARE_GPU", "0") == "1"


def env_world():
    """(world, rank, local_rank) from the launcher's environment; under GNMS_SHARE_GPU local_rank is folded onto the visible devices."""
    world, rank, local_rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if share_gpu() and torch.cuda.is_available():
        local_rank %= max(1, torch.cuda.device_count())
    return world, rank, local_rank


def init(backend=None):
    """Initialises torch.distributed from the launcher's env (RANK/WORLD_SIZE/MASTER_*).  Returns (world, rank, local_rank)."""
    world, rank, local_rank = env_world()
    force = os.environ.get("GNMS_FORCE_DIST", "0") == "1"      # exercise the collective path on a single rank (CI on 1-GPU boxes)
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if share_gpu():
            backend = "gloo"
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, **kw)
    return world, rank, local_rank


def shard_range(total, rank, world):
    """Contiguous [lo, hi) slice of `total` images owned by `rank`: sizes differ by at most one, every image
    belongs to exactly one rank (B split contiguously across ranks)."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(seconds, device=None):
    """MAX-reduce of a wall-clock measurement (the driver's timing contract)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(seconds)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


class StepHeartbeat:
    """The one collective of the pure-NMS scaling runs: a 4-byte all-reduce after every step, so that the 1 -> 8 GPU
    curve contains a real RCCL round trip per iteration.  Issued ASYNCHRONOUSLY like DDP's gradient all-reduce (ordered behind the
    step by an event on the process group's stream, overlapping the next step's kernels; not a blocking round trip) with no host
    synchronisation and ONE collective call per step: step i SUM-reduces slot i of a persistent vector of ones, in place -- no refill,
    no copy, no extra launch on a loop that is within 1.5x of being host-bound.
    `check()` (after the timed region) is a real test of the collectives: every used slot must hold exactly world_size (each rank
    contributed its 1 exactly once to exactly that step's reduction), the unused ones still 1, and the per-rank step counts must sum
    to steps x world.  A no-op when torch.distributed is not initialised."""

    CAPACITY = 8192                                   # slots; on wrap-around the used ones are verified and reset

    def __init__(self, device=None):
        self.on = dist.is_available() and dist.is_initialized()
        self.steps = 0
        self.used = 0
        self.works = []
        if self.on:
            if device is None:
                device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
            self.device = device
            self.buf = torch.ones(self.CAPACITY, dtype=torch.int32, device=device)
            self.slots = [self.buf[i:i + 1] for i in range(self.CAPACITY)]      # views made once: nothing but the collective call per beat

    def beat(self):
        self.steps += 1
        if self.on:
            if self.used == self.CAPACITY:
                self._verify_slots()
            # asynchronous, as DDP issues its bucket all-reduces: the collective runs on the process group's own stream behind an event
            # of the compute stream, so its cross-GPU round trip overlaps the next step's kernels instead of stalling them
            self.works.append(dist.all_reduce(self.slots[self.used], op=dist.ReduceOp.SUM, async_op=True))     # 4 bytes over RCCL / xGMI
            self.used += 1

    def _verify_slots(self):
        for w in self.works:
            w.wait()
        self.works = []
        world = dist.get_world_size()
        got = self.buf[:self.used]
        bad = int((got != world).sum().item()) + int((self.buf[self.used:] != 1).sum().item())
        if bad:
            raise RuntimeError("step heartbeat: %d of %d per-step all-reduces did not sum to world_size %d" % (bad, self.used, world))
        self.buf.fill_(1)
        self.used = 0

    def check(self):
        if not self.on:
            return
        self._verify_slots()
        total = torch.tensor([self.steps], dtype=torch.int32, device=self.device)
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
        if int(total.item()) != self.steps * dist.get_world_size():
            raise RuntimeError("step heartbeat: %d steps summed over the ranks, expected %d x %d" % (int(total.item()), self.steps,
                                                                                                        dist.get_world_size()))


def timed_steps(step, steps, warmup, sync, heartbeat=None):
    """warmup untimed steps, then exactly `steps` timed ones bracketed by barrier + device sync on both sides.
    Returns the MAX over ranks of the elapsed seconds.  heartbeat: a StepHeartbeat, beaten after every step."""
    for _ in range(warmup):
        step()
        if heartbeat is not None:
            heartbeat.beat()
    sync()
    barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
        if heartbeat is not None:
            heartbeat.beat()
    sync()
    barrier()
    sync()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    if heartbeat is not None:
        heartbeat.check()
    return elapsed


def relaunch_under_torchrun(gpus, script, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks (one process per GPU) under torch.distributed.run on this node
    and return their exit code.  The driver may also launch the ranks itself (WORLD_SIZE is then set and this is not called)."""
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC only on this host driver (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), script] + list(argv)
    return subprocess.call(cmd, env=env)
