"""Multi-GPU layout of the render path: one process per GPU, frames sharded, scalars reduced.

A render is a pure function of (Gaussians, one camera) and the reference's loops pick one camera per step
(/root/reference/trainers/fine_all.py:74-101), so frames are independent units: every rank holds the Gaussians,
takes frames rank, rank + world, ... and exchanges nothing on the data path.  The only collective is one all-reduce of
a few scalars (loss / PSNR sums, counts, timing) over RCCL (backend "nccl" on ROCm) -- or gloo on CPU in the tests.

Environment of a multi-process GPU run (the one place it is stated and checked):
  HSA_ENABLE_IPC_MODE_LEGACY=0   the host driver of the MI355X boxes supports dmabuf IPC only; without it RCCL's device-buffer
                                 exchange fails with `hipIpcGetMemHandle: invalid argument`.  It is read when the HSA runtime
                                 loads, so it must be in the environment BEFORE the first HIP call: `require_env()` sets it when
                                 nothing has touched the GPU yet and refuses to continue otherwise.
  MASTER_ADDR=127.0.0.1          single-node rendezvous (the container hostname may not resolve); MASTER_PORT, RANK, LOCAL_RANK,
                                 WORLD_SIZE come from torchrun, or from `init(force=True)` for a one-rank group.
"""
import os
import socket

import torch
import torch.distributed as dist

REQUIRED_ENV = {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}
_pg_device = None                  # device of the RCCL group (barrier / all_reduce run there)


def env_world():
    """(rank, world_size, local_rank) from the torchrun environment; (0, 1, 0) when launched plainly."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def require_env(backend):
    """What an RCCL group needs from the environment on hosts whose driver only supports dmabuf IPC: set while no HIP context exists,
    a warning when it is unset and the context is already up (a trainer that touched the GPU first; hosts with legacy IPC work
    without it, and an RCCL failure names the variable itself), an error only for an explicitly conflicting value."""
    if backend != "nccl":
        return
    for k, v in REQUIRED_ENV.items():
        have = os.environ.get(k)
        if have == v:
            continue
        if have is None and not torch.cuda.is_initialized():
            os.environ[k] = v
            continue
        if have is None:
            import warnings
            warnings.warn(f"{k} is unset and a HIP context already exists: on hosts that only support dmabuf IPC multi-process RCCL "
                          f"fails with 'hipIpcGetMemHandle: invalid argument' -- export {k}={v} in the launching shell", RuntimeWarning)
            continue
        raise RuntimeError(f"RCCL over xGMI needs {k}={v} in the environment before the first HIP call (found {have!r}); "
                           f"export it in the launching shell (egogaussian_amd/dist.py)")


def init(backend=None, device=None, force=False):
    """Create the process group of this run.  world == 1 creates none unless `force` (a one-rank group: the same RCCL
    initialisation, communicator and collectives as the multi-GPU run, on one device)."""
    global _pg_device
    rank, world, _ = env_world()
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if (device is not None and torch.device(device).type == "cuda") else "gloo"
        require_env(backend)
        if world == 1 and "MASTER_PORT" not in os.environ:
            s = socket.socket(); s.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(s.getsockname()[1]); s.close()
        os.environ.setdefault("RANK", str(rank)); os.environ.setdefault("WORLD_SIZE", str(world))
        kw = {"device_id": torch.device(device)} if backend == "nccl" and device is not None else {}
        _pg_device = torch.device(device) if backend == "nccl" and device is not None else None
        # RCCL prints a version banner with printf() when its communicator is created.  A caller whose stdout is a protocol (bench.py:
        # ONE JSON line) must not find it there: file descriptor 1 points at stderr while the group and its communicator come up.
        with _stdout_to_stderr(backend == "nccl"):
            dist.init_process_group(backend, rank=rank, world_size=world, **kw)
            if _pg_device is not None:
                t = torch.zeros(1, device=_pg_device)
                dist.all_reduce(t)                                # the first collective creates the communicator
                torch.cuda.synchronize(_pg_device)
    return rank, world


class _stdout_to_stderr:
    """fd 1 -> fd 2 for the duration of the block, C stdio buffers flushed on both edges (printf from native libraries lands on stderr)."""

    def __init__(self, on=True):
        self.on, self.saved = on, None

    def __enter__(self):
        if self.on:
            import ctypes, sys
            sys.stdout.flush()
            self.libc = ctypes.CDLL(None)
            self.libc.fflush(None)
            self.saved = os.dup(1)
            os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        if self.on and self.saved is not None:
            self.libc.fflush(None)
            os.dup2(self.saved, 1)
            os.close(self.saved)
        return False


def collective_name():
    """What carries the scalars of this run: "rccl" (backend nccl on ROCm), "gloo", or "none" (no process group)."""
    if not (dist.is_available() and dist.is_initialized()):
        return "none"
    return {"nccl": "rccl"}.get(dist.get_backend(), dist.get_backend())


def shard_frames(n_frames, rank, world):
    """Round-robin frame ids of this rank: disjoint across ranks, union = range(n_frames)."""
    return list(range(rank, n_frames, world))


def reduce_scalars(values, device="cpu", op="sum"):
    """All-reduce a short list of python floats; returns python floats.  Identity when there is no process group; with one
    (also a one-rank group) the collective really runs -- on the device for RCCL, on the host for gloo."""
    on = dist.is_available() and dist.is_initialized()
    if on and dist.get_backend() == "gloo":
        device = "cpu"
    elif on and _pg_device is not None:
        device = _pg_device
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    if on:
        dist.all_reduce(t, op={"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}[op])
    return [float(x) for x in t.tolist()]


def world_seen(device="cpu"):
    """How many ranks the collective itself saw: an all-reduce (sum) of a one on every rank, on a DEVICE tensor for RCCL -- the proof in a
    result line that N ranks really exchanged data over the group, not only that N processes ran.  0 when there is no process group."""
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    dev = "cpu" if dist.get_backend() == "gloo" else (_pg_device if _pg_device is not None else device)
    t = torch.ones(1, dtype=torch.float32, device=dev)
    dist.all_reduce(t)
    return int(round(float(t.item())))


def barrier():
    if dist.is_available() and dist.is_initialized():
        if _pg_device is not None:
            dist.barrier(device_ids=[_pg_device.index if _pg_device.index is not None else torch.cuda.current_device()])
        else:
            dist.barrier()


def shutdown():
    global _pg_device
    if dist.is_available() and dist.is_initialized():
        with _stdout_to_stderr(dist.get_backend() == "nccl"):
            dist.destroy_process_group()
    _pg_device = None
