"""Image losses and PSNR of the reference's training step (PyTorch ops, run on whatever device the images are on).

l1_loss / l2_loss / ssim follow /root/reference/utils/loss_utils.py:57-107 (11x11 Gaussian window, sigma 1.5,
zero padding 5, C1 = 0.01^2, C2 = 0.03^2, per-channel depthwise); psnr follows /root/reference/utils/image_utils.py:14-19
(20 log10(1/sqrt(mse)) per image).  The window is applied as its two 1-D factors (the 2-D window is their outer
product), which is the same filter at a fifth of the work.
"""
import math

import torch
import torch.nn.functional as F


def l1_loss(a, b):
    return (a - b).abs().mean()


def l2_loss(a, b):
    return ((a - b) ** 2).mean()


def _gauss1d(size, sigma, device, dtype):
    g = torch.tensor([math.exp(-(x - size // 2) ** 2 / (2.0 * sigma ** 2)) for x in range(size)], device=device, dtype=dtype)
    return g / g.sum()


def _blur(x, g, C):
    k = g.numel()
    x = F.conv2d(x, g.view(1, 1, k, 1).expand(C, 1, k, 1), padding=(k // 2, 0), groups=C)
    return F.conv2d(x, g.view(1, 1, 1, k).expand(C, 1, 1, k), padding=(0, k // 2), groups=C)


def ssim(img1, img2, window_size=11, size_average=True):
    squeeze = img1.dim() == 3
    if squeeze:
        img1, img2 = img1.unsqueeze(0), img2.unsqueeze(0)
    C = img1.shape[1]
    g = _gauss1d(window_size, 1.5, img1.device, img1.dtype)
    mu1, mu2 = _blur(img1, g, C), _blur(img2, g, C)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = _blur(img1 * img1, g, C) - mu1_sq
    s2 = _blur(img2 * img2, g, C) - mu2_sq
    s12 = _blur(img1 * img2, g, C) - mu12
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu12 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))
    return m.mean() if size_average else m.mean(1).mean(1).mean(1)


def psnr(img1, img2):
    mse = ((img1 - img2) ** 2).reshape(img1.shape[0], -1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))


def training_loss(image, gt, lambda_dssim=0.2):
    """(1 - lambda) L1 + lambda (1 - SSIM), /root/reference/trainers/train_static.py:92-95, arguments/__init__.py:83."""
    return (1.0 - lambda_dssim) * l1_loss(image, gt) + lambda_dssim * (1.0 - ssim(image, gt))
