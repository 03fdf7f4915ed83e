"""Evaluation metrics of the inference configuration: the 8-bit quantisation between G and the metric
(tester.py:70-71), PSNR (metrics/CalcPSNR.py:85-92) and skimage-default SSIM (metrics/CalcSSIM.py:63) with the 4-pixel border
crop -- device kernels (uegan_quantize_u8, uegan_image_metrics_u8) against the numpy/scipy oracle and closed-form answers."""
import math

import numpy as np
import pytest
import torch

from helpers import BACKENDS, use_backend
from oracle import uegan_oracle as O
from uegan_amd import tester


def _images(seed, B=2, H=40, W=52):
    g = torch.Generator().manual_seed(seed)
    # smooth + noise, partly outside [-1, 1] so that denorm's clamp and the 8-bit rounding are exercised
    base = torch.nn.functional.interpolate(torch.rand(B, 3, 5, 7, generator=g), size=(H, W), mode="bilinear", align_corners=True) * 2.4 - 1.2
    a = base + 0.02 * torch.randn(B, 3, H, W, generator=g)
    b = base + 0.05 * torch.randn(B, 3, H, W, generator=g)
    return a, b


@pytest.mark.parametrize("backend", BACKENDS)
def test_quantize_psnr_ssim_match_oracle(backend):
    dev = use_backend(backend)
    a, b = _images(3)
    qa, qb = tester.to_uint8_image(a.to(dev)), tester.to_uint8_image(b.to(dev))
    assert torch.equal(qa.cpu(), O.to_uint8_image(a)) and torch.equal(qb.cpu(), O.to_uint8_image(b))     # bit-exact 8-bit images
    psnr = tester.calculate_psnr(qa, qb)
    ssim = tester.calculate_ssim(qa, qb)
    for i in range(a.shape[0]):
        ra, rb = O.to_uint8_image(a)[i].numpy(), O.to_uint8_image(b)[i].numpy()
        assert abs(psnr[i] - O.psnr_u8(ra, rb)) < 1e-9
        assert abs(ssim[i] - O.ssim_u8_skimage(ra, rb)) < 1e-9
        assert abs(tester.calculate_psnr(qa[i], qb[i]) - psnr[i]) < 1e-12      # HWC form
    assert abs(tester.mean_metric(psnr) - sum(psnr) / len(psnr)) < 1e-12


@pytest.mark.parametrize("backend", BACKENDS)
def test_metric_known_answers(backend):
    dev = use_backend(backend)
    H, W = 24, 30
    x = torch.full((1, H, W, 3), 100, dtype=torch.uint8, device=dev)
    y = torch.full((1, H, W, 3), 110, dtype=torch.uint8, device=dev)
    assert tester.calculate_psnr(x, x) == [float("inf")]
    assert abs(tester.calculate_ssim(x, x)[0] - 1.0) < 1e-12
    # constant images: mse = 100 -> 10 log10(65025/100); SSIM = (2ab + C1)/(a^2 + b^2 + C1) (all variances zero)
    assert abs(tester.calculate_psnr(x, y)[0] - 10 * math.log10(255.0 ** 2 / 100.0)) < 1e-9
    C1 = (0.01 * 255) ** 2
    assert abs(tester.calculate_ssim(x, y)[0] - (2 * 100 * 110 + C1) / (100 ** 2 + 110 ** 2 + C1)) < 1e-12
    # the border crop really is excluded: a difference confined to the 4-pixel frame changes nothing
    z = x.clone()
    z[:, :4] = 0
    z[:, :, :4] = 255
    assert tester.calculate_psnr(x, z) == [float("inf")]
    with pytest.raises(ValueError):
        tester.calculate_psnr(x, y[:, :-1])


def test_oracle_metric_restatement_is_self_consistent():
    """the oracle's SSIM against a direct per-window evaluation of the same published formula (no filtering library)"""
    a, b = _images(5, B=1, H=20, W=22)
    ra, rb = O.to_uint8_image(a)[0].numpy().astype(np.float64), O.to_uint8_image(b)[0].numpy().astype(np.float64)
    ca, cb = ra[4:-4, 4:-4], rb[4:-4, 4:-4]
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    vals = []
    for c in range(3):
        for y in range(ca.shape[0] - 6):
            for x in range(ca.shape[1] - 6):
                p, q = ca[y:y + 7, x:x + 7, c].ravel(), cb[y:y + 7, x:x + 7, c].ravel()
                cov = np.cov(p, q, ddof=1)
                vals.append(((2 * p.mean() * q.mean() + C1) * (2 * cov[0, 1] + C2)) / ((p.mean() ** 2 + q.mean() ** 2 + C1) * (cov[0, 0] + cov[1, 1] + C2)))
    assert abs(np.mean(vals) - O.ssim_u8_skimage(ra, rb)) < 1e-9
