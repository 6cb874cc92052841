#!/usr/bin/env python
"""Writes bench_data/trained_scene.npz: the model examples/train_synth.py ends with after the reference's full schedule (bench.trained_scene),
run ONCE on an MI355X; bench.py's `trained_scene_op_only` leg and tests/test_gpu_trained_scene.py load the committed arrays so that every run
sees the same densified model.  Usage (GPU box):  python tools/make_trained_scene.py [--iters 30000] [--out gpurun_out/trained_scene.npz]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30000)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "trained_scene.npz"))
    ap.add_argument("--measure", action="store_true", help="also print the per-stage op-only leg on the new arrays")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    scene, how = bench.trained_scene(dev, a.iters, regenerate=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    np.savez_compressed(a.out, train_iters=np.int64(a.iters), **{k: np.ascontiguousarray(v, dtype=np.float32) for k, v in scene.items()})
    print(f"wrote {a.out}: {scene['xyz'].shape[0]} Gaussians, {os.path.getsize(a.out) / 1e6:.1f} MB", file=sys.stderr)
    if a.measure:
        leg = bench.config_leg(dev, scene["xyz"].shape[0], 540, 960, False, iters=20, scene=scene, what=how)
        print(json.dumps(leg))


if __name__ == "__main__":
    main()
