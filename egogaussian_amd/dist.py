"""Multi-GPU layout of the render path: one process per GPU, frames sharded, scalars reduced.

A render is a pure function of (Gaussians, one camera) and the reference's loops pick one camera per step
(/root/reference/trainers/fine_all.py:74-101), so frames are independent units: every rank holds the Gaussians,
takes frames rank, rank + world, ... and exchanges nothing on the data path.  The only collective is one all-reduce of
a few scalars (loss / PSNR sums, counts, timing) over RCCL (backend "nccl" on ROCm) -- or gloo on CPU in the tests.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, world_size, local_rank) from the torchrun environment; (0, 1, 0) when launched plainly."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None, device=None):
    rank, world, _ = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if (device is not None and torch.device(device).type == "cuda") else "gloo"
        kw = {"device_id": torch.device(device)} if backend == "nccl" and device is not None else {}
        dist.init_process_group(backend, **kw)
    return rank, world


def shard_frames(n_frames, rank, world):
    """Round-robin frame ids of this rank: disjoint across ranks, union = range(n_frames)."""
    return list(range(rank, n_frames, world))


def reduce_scalars(values, device="cpu", op="sum"):
    """All-reduce a short list of python floats; returns python floats.  Identity when not distributed."""
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "gloo":
        device = "cpu"
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op={"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}[op])
    return [float(x) for x in t.tolist()]


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def shutdown():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
