#!/usr/bin/env python
"""Times the device-side densification bookkeeping (egogaussian_amd/densify.py) at config C size:
per-iteration statistics kernel, and one densify_and_prune call (plan + gathers + optimizer surgery), wall clock with
synchronisation, against the same steps written with PyTorch boolean-mask indexing the way the reference does them."""
import math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egogaussian_amd import densify
from egogaussian_amd.scene_synth import make_scene, SynthGaussians

N, H, W = 500_000, 540, 960
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(0)


def fresh():
    pc = SynthGaussians(make_scene(N, H, W, 0), device=dev)
    pc.training_setup()
    for p in pc.optimizer.param_groups:
        p["params"][0].grad = torch.zeros_like(p["params"][0])
    pc.optimizer.step(); pc.optimizer.zero_grad(set_to_none=True)
    pc.xyz_gradient_accum = torch.rand(N, 1, device=dev, generator=gen) * 1e-3
    pc.denom = torch.randint(0, 4, (N, 1), device=dev, generator=gen).float()
    return pc


pc = fresh()
vs = torch.zeros(N, 3, device=dev, requires_grad=True)
vs.grad = torch.randn(N, 3, device=dev, generator=gen) * 1e-4
radii = (torch.rand(N, device=dev, generator=gen) * 30).int() * (torch.rand(N, device=dev, generator=gen) < 0.6)
vis = radii > 0


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


def torch_stats():                      # trainers/train_static.py:125-127 + gaussian_model.py:735-740
    pc.max_radii2D[vis] = torch.max(pc.max_radii2D[vis], radii[vis].float())
    pc.xyz_gradient_accum[vis] += torch.norm(vs.grad[vis, :2], dim=-1, keepdim=True)
    pc.denom[vis] += 1


print(f"per-iteration statistics, N={N}:  HIP kernel {timeit(lambda: densify.add_densification_stats(pc, vs, vis, radii=radii)):.1f} us   "
      f"torch masked indexing {timeit(torch_stats):.1f} us")
ts = []
for rep in range(5):
    pc = fresh()
    torch.cuda.synchronize(); t = time.perf_counter()
    n0, n1 = densify.densify_and_prune(pc, 2e-4, 0.005, 10.0, 20)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
print(f"densify_and_prune {n0} -> {n1} Gaussians: {min(ts) * 1e3:.2f} ms (best of 5, incl. the one host read and the optimizer surgery)")
