"""Inference path (tester.py:58-67): `G.eval()`, `torch.no_grad()`, one `G(x)` per image; plus the reference's
`denorm` (utils.py:128-130) and PSNR (metrics/CalcPSNR.py:85-92, border crop :24,56) restated for the
inference-parity configuration."""
import math

import torch


def denorm(x):
    """utils.py:128-130"""
    out = (x + 1) / 2.0
    return out.clamp_(0, 1)


@torch.no_grad()
def enhance(G, x):
    """tester.py:58-67 inner loop body: eval-mode generator forward."""
    G.eval()
    return G(x)


def to_uint8_image(x):
    """What torchvision.utils.save_image does to a [0,1] tensor before PNG encoding (tester.py:70-71):
    mul(255).add_(0.5).clamp_(0,255) -> uint8, CHW -> HWC."""
    return denorm(x.detach().clone()).mul(255).add_(0.5).clamp_(0, 255).permute(0, 2, 3, 1).to(torch.uint8)


def calculate_psnr(img1, img2, crop_border=4):
    """metrics/CalcPSNR.py:85-92 on uint8 HWC images with the 4-pixel border crop of :24,56."""
    a = img1.double()
    b = img2.double()
    if crop_border:
        a = a[crop_border:-crop_border, crop_border:-crop_border]
        b = b[crop_border:-crop_border, crop_border:-crop_border]
    mse = torch.mean((a - b) ** 2).item()
    if mse == 0:
        return float("inf")
    return 20 * math.log10(255.0 / math.sqrt(mse))
