"""Data parallelism for the training step: one process per GPU, full replica per
rank, batch sharded `rank::world`, ONE exchange per step — a sum all-reduce of the flat fp32
gradient buffer (RCCL over xGMI; backend "nccl" is RCCL on ROCm, "gloo" for the CPU tests).
The mean over ranks is folded into the loss scale (1/world) so no extra scaling pass is needed.
BatchNorm statistics stay per replica, as with the reference's only multi-device mechanisms
(nn.DataParallel integration.py:1058-1060; tf.distribute default strategy nb4 c1:63-75).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized()


def world_size() -> int:
    return dist.get_world_size() if is_distributed() else 1


def rank() -> int:
    return dist.get_rank() if is_distributed() else 0


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """Initialise torch.distributed from RANK/WORLD_SIZE/LOCAL_RANK/MASTER_* (torchrun)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rk = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not is_distributed():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rk, world_size=world)
    return rk, world, local


def shard_batch(x, y, rk: int | None = None, world: int | None = None):
    """Per-rank shard of a global batch: samples rk, rk+world, ...."""
    rk = rank() if rk is None else rk
    world = world_size() if world is None else world
    return x[rk::world], y[rk::world]


def allreduce_sum_(flat: torch.Tensor) -> torch.Tensor:
    """In-place sum over ranks of the flat gradient bucket."""
    if world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def overlap_enabled() -> bool:
    """ISHARA_OVERLAP_ALLREDUCE=1: reduce the gradient in buckets on a side stream while the backward pass is still running
    (allreduce_buckets_).  Off by default: the single flat all-reduce after the backward pass is the path the 1-GPU box can
    rehearse end to end; the bucketed path is checked against it bit for bit with two gloo ranks (tests/test_bench_gpu.py)."""
    return os.environ.get("ISHARA_OVERLAP_ALLREDUCE", "0") == "1"


_side_streams = {}


def allreduce_ranges_(flat: torch.Tensor, ranges, before_range=None) -> None:
    """Sum over ranks of `flat[off:off+cnt]` for every (off, cnt) in `ranges`, one asynchronous collective per non-empty
    range, issued in the given order and all waited for before returning.  `before_range(i)` runs right before range i's
    collective is enqueued (the GPU path makes the side stream wait for that range's completion event there).  Device
    agnostic: the ordering logic is what tests/test_parallel_gloo.py checks against the flat all-reduce with 3 gloo ranks."""
    works = []
    for i, (off, cnt) in enumerate(ranges):
        if cnt <= 0:
            continue
        if before_range is not None:
            before_range(i)
        works.append(dist.all_reduce(flat[off:off + cnt], op=dist.ReduceOp.SUM, async_op=True))
    for w in works:
        w.wait()


def allreduce_buckets_(model) -> None:
    """Sum over ranks of model.grads[:n_train], one collective per gradient bucket.  The backward pass is already enqueued on
    the current stream when this is called; bucket i's collective is enqueued on a side stream that waits only for the event the
    library recorded when that range became final, so it runs while the rest of the backward pass executes.  The current
    stream waits for all of them at the end (before the optimizer step).  NOT yet validated over RCCL on a multi-GPU node
    (only over gloo): off by default, see overlap_enabled()."""
    dev = model.grads.device
    side = _side_streams.get(dev)
    if side is None:
        side = _side_streams[dev] = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        allreduce_ranges_(model.grads, model.grad_buckets(), before_range=lambda i: model.wait_grad_bucket(i, side))   # w.wait(): side <- collective
    torch.cuda.current_stream(dev).wait_stream(side)


def reduce_max(value: float, device=None) -> float:
    """Max over ranks of a host scalar (step-time aggregation in bench.py)."""
    if world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if world_size() > 1:
        dist.barrier()
