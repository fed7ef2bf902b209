"""Image-quality metrics of the reference's evaluation scripts, restated.

* ``batch_psnr``  -- DN_Gray/utils.py:18-24 (``skimage`` compare_psnr per image, averaged)
* ``quantize`` / ``calc_psnr`` -- DN_Gray/utility.py:129-150 (8-bit quantisation, border shave, luma for RGB benchmarks)
* ``ssim``        -- Demosaic/pytorch_ssim/__init__.py:20-76 (11x11 Gaussian window, sigma 1.5)
All on whatever device the inputs live on; no host round trip.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def batch_psnr(img: torch.Tensor, clean: torch.Tensor, data_range: float = 1.0) -> float:
    """Mean over the batch of 10*log10(range^2 / mse_i)."""
    d = (img.double() - clean.double()).flatten(1)
    mse = (d * d).mean(dim=1)
    return float((10.0 * torch.log10(data_range ** 2 / mse)).mean())


def quantize(img: torch.Tensor, rgb_range: float) -> torch.Tensor:
    pixel_range = 255.0 / rgb_range
    return img.mul(pixel_range).clamp(0, 255).round().div(pixel_range)


def calc_psnr(sr: torch.Tensor, hr: torch.Tensor, scale: int, rgb_range: float, benchmark: bool = False) -> float:
    diff = (sr - hr) / rgb_range
    if benchmark:
        shave = scale
        if diff.size(1) > 1:
            convert = diff.new_tensor([65.738, 129.057, 25.064]).view(1, 3, 1, 1)
            diff = (diff * convert / 256).sum(dim=1, keepdim=True)
    else:
        shave = scale + 6
    valid = diff[:, :, shave:-shave, shave:-shave]
    return -10.0 * math.log10(float(valid.pow(2).mean()))


def _window(size: int, channel: int, like: torch.Tensor) -> torch.Tensor:
    g = torch.tensor([math.exp(-(x - size // 2) ** 2 / (2 * 1.5 ** 2)) for x in range(size)], dtype=torch.float32)
    g = (g / g.sum()).unsqueeze(1)
    return (g @ g.t()).expand(channel, 1, size, size).contiguous().to(like)


def ssim(img1: torch.Tensor, img2: torch.Tensor, window_size: int = 11, size_average: bool = True):
    c = img1.size(1)
    w = _window(window_size, c, img1)
    pad = window_size // 2
    mu1, mu2 = F.conv2d(img1, w, padding=pad, groups=c), F.conv2d(img2, w, padding=pad, groups=c)
    s11 = F.conv2d(img1 * img1, w, padding=pad, groups=c) - mu1 * mu1
    s22 = F.conv2d(img2 * img2, w, padding=pad, groups=c) - mu2 * mu2
    s12 = F.conv2d(img1 * img2, w, padding=pad, groups=c) - mu1 * mu2
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s11 + s22 + c2))
    return m.mean() if size_average else m.mean(dim=(1, 2, 3))
