"""CPU, world_size 2 over gloo: the N > 1 layout of the path -- frames sharded round-robin, no data-path collective,
scalars all-reduced (egogaussian_amd/dist.py)."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from egogaussian_amd import dist as d
    r, w = d.init("gloo")
    assert (r, w) == (rank, world) and d.env_world() == (rank, world, rank)
    frames = d.shard_frames(300, r, w)
    # every rank "renders" its frames: a per-frame scalar stands in for the per-frame loss / PSNR
    local = [float(sum(k * 0.5 for k in frames)), float(len(frames)), 1.0 + rank]
    tot = d.reduce_scalars(local, "cpu", "sum")
    tmax = d.reduce_scalars([10.0 + rank], "cpu", "max")[0]
    d.barrier()
    assert d.world_seen() == world                              # the all-reduce of ones bench.py prints as collective_world_seen
    out[rank] = (frames, tot, tmax)
    d.shutdown()


def test_frame_sharding_and_scalar_allreduce_world2():
    world = 2
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    f0, t0, m0 = out[0]; f1, t1, m1 = out[1]
    assert sorted(f0 + f1) == list(range(300)) and not set(f0) & set(f1) and abs(len(f0) - len(f1)) <= 1
    assert t0 == t1 and m0 == m1 == 11.0
    assert abs(t0[0] - 0.5 * sum(range(300))) < 1e-9 and t0[1] == 300.0 and t0[2] == 3.0


def test_single_process_is_identity():
    from egogaussian_amd import dist as d
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    assert d.env_world() == (0, 1, 0) and d.shard_frames(7, 0, 1) == list(range(7))
    assert d.reduce_scalars([1.5, 2.0]) == [1.5, 2.0]
    d.barrier()
    assert d.world_seen() == 0                                  # no process group: nothing was exchanged


def test_bench_plain_launch_command():
    """`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run, one rank per GPU, rendezvous
    on 127.0.0.1 (the container's hostname may not resolve)."""
    import importlib.util, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    cmd = bench.spawn_command(4, ["--gpus", "4", "--steps", "5"], port=29517)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29517"
    assert cmd[-5:] == [os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "5"]
    port = int(bench.spawn_command(2, [])[bench.spawn_command(2, []).index("--master-port") + 1])
    assert 1024 < port < 65536


def test_counter_files_are_tied_to_the_kernel_sources():
    """bench.py reports profiles/*.json numbers only when they were collected on the present kernel sources."""
    import importlib.util, json, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    from egogaussian_amd import lib
    h = lib.kernel_source_hash()
    assert len(h) == 16 and h == lib.kernel_source_hash()
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "c.json")
        json.dump({"_source_hash": h, "500000@960x540": {"render_backward": {"hbm_bytes_per_launch": 5}}}, open(p, "w"))
        ent, why = bench.stamped(p, "500000@960x540", h)
        assert why is None and ent["render_backward"]["hbm_bytes_per_launch"] == 5
        ent, why = bench.stamped(p, "500000@960x540", "0" * 16)
        assert ent is None and "collected at kernel-source hash" in why
        ent, why = bench.stamped(p, "1@1x1", h)
        assert ent is None and "no entry" in why
        assert bench.stamped(os.path.join(d, "absent.json"), "k", h) == (None, "absent.json absent")


def test_rccl_environment_is_stated_and_checked(monkeypatch):
    """The one place the multi-process GPU environment is stated: dist.REQUIRED_ENV; a wrong value is refused, a missing one is set
    while no HIP context exists (nothing has read it yet)."""
    from egogaussian_amd import dist as d
    import pytest
    assert d.REQUIRED_ENV == {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "1")
    with pytest.raises(RuntimeError, match="HSA_ENABLE_IPC_MODE_LEGACY=0"):
        d.require_env("nccl")
    d.require_env("gloo")                                         # gloo needs nothing
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY")
    d.require_env("nccl")
    assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert d.collective_name() == "none"
