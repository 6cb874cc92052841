"""GPU: the first RCCL call happens HERE, before any multi-GPU run.  A one-rank process group with backend "nccl" (= RCCL on ROCm)
created with device_id= exactly as egogaussian_amd/dist.py does for N ranks, then the collectives the path uses -- the scalar
all-reduce (sum / max), a barrier, shutdown -- on device tensors, in a child process so that the group and its environment do not
leak into the test session.  (/root/reference/trainers/fine_all.py:74-101 is the loop whose frames are sharded)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys
sys.path.insert(0, %r)
import torch
from egogaussian_amd import dist as d
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
rank, world = d.init("nccl", dev, force=True)
import torch.distributed as td
out = {"rank": rank, "world": world, "backend": td.get_backend(), "collective": d.collective_name(), "initialized": td.is_initialized()}
out["sum"] = d.reduce_scalars([1.5, 2.0, -3.25], dev, "sum")
out["max"] = d.reduce_scalars([10.0], dev, "max")
d.barrier()
t = torch.arange(16, device=dev, dtype=torch.float32)          # a 64-byte device vector, the size SURVEY 8e names
td.all_reduce(t)
torch.cuda.synchronize()
out["vec_ok"] = bool(torch.equal(t.cpu(), torch.arange(16, dtype=torch.float32)))
out["env"] = {k: os.environ.get(k) for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "MASTER_ADDR", "WORLD_SIZE", "RANK")}
d.shutdown()
out["after_shutdown"] = td.is_initialized()
print("RESULT " + json.dumps(out))
""" % ROOT


def test_rccl_one_rank_group_collectives():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    out = json.loads(line[len("RESULT "):])
    assert out["backend"] == "nccl" and out["collective"] == "rccl" and out["initialized"] and not out["after_shutdown"]
    assert (out["rank"], out["world"]) == (0, 1)
    assert out["sum"] == [1.5, 2.0, -3.25] and out["max"] == [10.0] and out["vec_ok"]
    assert out["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and out["env"]["MASTER_ADDR"] == "127.0.0.1"


def test_bench_single_gpu_line_says_rccl():
    """The default single-GPU bench run creates the one-rank RCCL group: its barriers and scalar reductions are the N > 1 code path."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "5", "--gaussians", "20000", "--height", "135",
                        "--width", "240", "--no-cpu-baseline", "--no-sh3-leg", "--no-fine-all-leg", "--no-config-legs", "--spinup-ms", "0"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["collective"] == "rccl" and "collective_error" not in j and j["n_gpus"] == 1 and j["value"] > 0
    assert j["rccl_world_seen"] == 1


def test_bench_two_ranks_over_rccl_on_two_devices():
    """The first multi-rank RCCL initialisation happens here when the box has two devices (the reference's `fine_all` loop sharded over
    GPUs, /root/reference/trainers/fine_all.py:74-101): `python bench.py --gpus 2 --verify-ranks` with the default `nccl` backend, one
    rank per device, disjoint frames, equal replicas at the start, scalars reduced over RCCL."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip(f"needs two HIP devices for a two-rank RCCL group (this box has {torch.cuda.device_count()}); "
                    "tests/test_gpu_bench_mode.py::test_bench_two_ranks_plain_launch covers two ranks on one device over gloo")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "EGS_BENCH_SHARE_DEVICE0",
                                                            "EGS_BENCH_BACKEND")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "5", "--gaussians", "60000", "--height", "270",
           "--width", "480", "--no-cpu-baseline", "--no-sh3-leg", "--verify-ranks"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert j["collective"] == "rccl" and "collective_error" not in j
    assert j["n_gpus"] == 2 and j["steps"] == 10 and j["scaling"] == "weak" and j["value"] > 0
    assert j["rccl_world_seen"] == 2                             # RCCL itself saw both ranks (an all-reduce of ones on device tensors)
    r0, r1 = sorted(j["ranks"], key=lambda x: x["rank"])
    assert (r0["rank"], r1["rank"]) == (0, 1) and (r0["device"], r1["device"]) == ("cuda:0", "cuda:1")
    assert r0["frames"] == list(range(0, 30, 2)) and r1["frames"] == list(range(1, 30, 2))
    assert r0["param_checksum_start"] == r1["param_checksum_start"]
    assert r0["graph"] and r1["graph"] and r0["overflow"]["ok"] and r1["overflow"]["ok"]
    assert r0["loss_sum"] > 0 and r1["loss_sum"] > 0 and r0["loss_sum"] != r1["loss_sum"]
    assert abs(j["mean_loss"] - (r0["loss_sum"] + r1["loss_sum"]) / 20) < 1e-6
    assert j["fine_all_shape"]["n_gpus"] == 2 and j["fine_all_shape"]["value"] > 0
