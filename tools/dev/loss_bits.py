"""Dump the fused loss's value, gradient and (through the graph-replay path's forward_ex) nothing else for a seeded image pair, so that two builds
of the library can be compared bit for bit:  EGS_RASTER_LIB=a.so python tools/dev/loss_bits.py /tmp/a.pt;  EGS_RASTER_LIB=b.so ... /tmp/b.pt;
python tools/dev/loss_bits.py /tmp/a.pt /tmp/b.pt"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) == 3:
    a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
    for k in a:
        same = torch.equal(a[k], b[k])
        print(k, "bit-identical" if same else f"DIFFERENT: max |d| {float((a[k].double() - b[k].double()).abs().max()):.3e}")
    sys.exit(0 if all(torch.equal(a[k], b[k]) for k in a) else 1)
from egogaussian_amd.fused import l1_ssim_loss
out = {}
for (C, H, W) in ((3, 540, 960), (3, 37, 50), (1, 16, 16), (3, 71, 129)):
    g = torch.Generator().manual_seed(H * W)
    x = torch.rand(C, H, W, generator=g).cuda().requires_grad_(True)
    y = (x.detach() + 0.1 * torch.randn(C, H, W, generator=g).cuda()).clamp(0, 1)
    l = l1_ssim_loss(x, y, 0.2)
    l.backward()
    out[f"loss_{C}x{H}x{W}"] = l.detach().cpu(); out[f"grad_{C}x{H}x{W}"] = x.grad.cpu()
torch.save(out, sys.argv[1])
print("saved", sys.argv[1])
