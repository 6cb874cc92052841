#!/usr/bin/env python
"""Run the fused loss forward + backward a few times at 3x540x960 (for rocprofv3 A/B runs of library variants)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egogaussian_amd.fused import l1_ssim_loss
H, W = int(os.environ.get("H", 540)), int(os.environ.get("W", 960))
g = torch.Generator().manual_seed(0)
a = torch.rand(3, H, W, generator=g).cuda().requires_grad_(True)
b = torch.rand(3, H, W, generator=g).cuda()
for _ in range(30):
    l1_ssim_loss(a, b, 0.2).backward()
    a.grad = None
torch.cuda.synchronize()
