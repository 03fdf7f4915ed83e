"""Inference path (tester.py:58-71) and the evaluation metrics of the inference configuration, on the device.

    enhance(G, x)                     tester.py:58-67: `G.eval()`, `torch.no_grad()`, one `G(x)` per image
    GraphedGenerator(G, shape)        the same forward captured once into a hipGraph and replayed (batch-1 inference is
                                      ~60 dependent launches: launch latency, not arithmetic, sets its time)
    to_uint8_image(x)                 what tester.py:70-71 writes to a PNG: denorm (utils.py:128-130) + torchvision save_image's
                                      mul(255).add(0.5).clamp(0,255).to(uint8), NHWC
    calculate_psnr / calculate_ssim   metrics/CalcPSNR.py:85-92 and metrics/CalcSSIM.py:63 (skimage defaults) with the 4-pixel
                                      border crop both scripts apply (:24,56), computed by libuegan_hip.so kernels
    mean_metric(values)               the TRUE mean; the reference's directory averages divide by N-1 (CalcPSNR.py:77, CalcSSIM.py:75)
    run_test(G, loader, ...)          the loop of Tester.test (tester.py:40-105): enhance every batch of a test loader, write the
                                      PNGs `save_image` would write, PSNR / SSIM against the labels on the device
"""
import math

import torch

from . import _lib as L
from . import ops

CROP_BORDER = 4          # CalcPSNR.py:24 / CalcSSIM.py:24


def denorm(x):
    """utils.py:128-130"""
    out = (x + 1) / 2.0
    return out.clamp_(0, 1)


@torch.no_grad()
def enhance(G, x):
    """tester.py:58-67 inner loop body: eval-mode generator forward."""
    G.eval()
    return G(x)


class GraphedGenerator:
    """`enhance` for a fixed input shape as one hipGraph launch: the eval-mode forward is captured once on a side stream
    (torch.cuda.CUDAGraph = hipGraph on ROCm; every kernel of the forward is enqueued on the capturing stream by the C ABI and
    nothing in it allocates or synchronises) and replayed per image.  Weights are read at replay time, but their PACKED copies
    are made at capture time: re-capture (`.capture()`) after the weights change."""

    def __init__(self, G, shape, device=None):
        self.G = G
        dev = device if device is not None else next(G.parameters()).device
        self.x = torch.zeros(shape, dtype=torch.float32, device=dev)
        self.graph = None
        self.y = None
        self.capture()

    @torch.no_grad()
    def capture(self):
        self.G.eval()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):                       # warm-up: packs the weights, fills the allocator pool
                self.G(self.x)
        torch.cuda.current_stream().wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.y = self.G(self.x)

    @torch.no_grad()
    def __call__(self, x):
        self.x.copy_(x)
        self.graph.replay()
        return self.y


def to_uint8_image(x):
    """[-1,1] NCHW fp32 -> uint8 NHWC as tester.py:70-71 + torchvision.utils.save_image produce it (uegan_quantize_u8)."""
    x = x.detach().contiguous()
    if x.dtype != torch.float32 or x.dim() != 4:
        raise TypeError("to_uint8_image expects a float32 [B,C,H,W] tensor")
    B, C, H, W = x.shape
    y = torch.empty((B, H, W, C), dtype=torch.uint8, device=x.device)
    ops._chk(x, y)
    L.check(ops.lib().uegan_quantize_u8(x.data_ptr(), y.data_ptr(), B, C, H, W, ops._stream()))
    return y


def _as_stack(img):
    if img.dtype != torch.uint8:
        raise TypeError("metrics take uint8 HWC / BHWC images (to_uint8_image)")
    return (img.unsqueeze(0) if img.dim() == 3 else img).contiguous()


def _metrics(img1, img2, crop_border, want_sq, want_ssim):
    a, b = _as_stack(img1), _as_stack(img2)
    if a.shape != b.shape:
        raise ValueError("Input images must have the same dimensions.")
    B, H, W, C = a.shape
    sq = torch.empty((B,), dtype=torch.float64, device=a.device) if want_sq else None
    ss = torch.empty((B,), dtype=torch.float64, device=a.device) if want_ssim else None
    ops._chk(a, b)
    L.check(ops.lib().uegan_image_metrics_u8(a.data_ptr(), b.data_ptr(), ops._p(sq), ops._p(ss), B, H, W, C, crop_border, ops._stream()))
    h, w = H - 2 * crop_border, W - 2 * crop_border
    return sq, ss, h * w * C, (h - 6) * (w - 6) * C


def calculate_psnr(img1, img2, crop_border=CROP_BORDER):
    """metrics/CalcPSNR.py:85-92 on uint8 HWC images (or a BHWC stack -> list) after the border crop of :24,56."""
    sq, _, n, _ = _metrics(img1, img2, crop_border, True, False)
    out = []
    for v in sq.tolist():
        mse = v / n
        out.append(float("inf") if mse == 0 else 10 * math.log10(255.0 ** 2 / mse))
    return out[0] if img1.dim() == 3 else out


def calculate_ssim(img1, img2, crop_border=CROP_BORDER):
    """metrics/CalcSSIM.py:63: skimage structural_similarity(multichannel=True, data_range=255) with its defaults (7x7 uniform
    window, K1 0.01, K2 0.03, sample covariance) on the border-cropped uint8 images."""
    _, ss, _, n = _metrics(img1, img2, crop_border, False, True)
    out = [v / n for v in ss.tolist()]
    return out[0] if img1.dim() == 3 else out


def mean_metric(values):
    """True mean over a test set.  (The reference's directory loops return total / i with i = N - 1: CalcPSNR.py:77, CalcSSIM.py:75.)"""
    values = list(values)
    return sum(values) / len(values)


def run_test(G, loader, save_dir=None, tag="0.00", metrics=True):
    """Tester.test (tester.py:40-105) over a `uegan_amd.data` test loader: `G.eval()` forward per batch (:64-67), the enhanced image of
    every sample as `<name>_<tag>_testFakeExp.png` in `save_dir` (:69-71: the 8-bit image torchvision's save_image writes; None: no
    files), and -- what calc_psnr / calc_ssim then compute from those files against the label images (:96-103) -- PSNR and SSIM of
    each enhanced image against `img_exp`, here straight from the device tensors.  Returns {"names", "psnr", "ssim", "mean_psnr",
    "mean_ssim"} (true means).

    Restriction: the label here is the loader's `img_exp` -- the label FILE resized to the test size by the loader's transform
    (data_loader.py:95-99) and re-quantised to 8 bits -- whereas calc_psnr / calc_ssim read the ORIGINAL files of test_label_dir.  The
    numbers agree with the reference's when the label files already have the test size (the reference itself needs equal shapes:
    CalcPSNR.py:87 raises otherwise); for labels of another size decode them yourself and call calculate_psnr / calculate_ssim.  The
    test_compare montage images (tester.py:73-90) are not written."""
    import os
    names, psnr, ssim = [], [], []
    if save_dir is not None:
        os.makedirs(save_dir, exist_ok=True)
    for batch in loader:
        fake = enhance(G, batch.img_raw)
        q = to_uint8_image(fake)
        if metrics:
            ref = to_uint8_image(batch.img_exp)
            psnr += calculate_psnr(q, ref)
            ssim += calculate_ssim(q, ref)
        names += list(batch.img_name)
        if save_dir is not None:
            from PIL import Image
            host = q.cpu().numpy()
            for i, name in enumerate(batch.img_name):
                Image.fromarray(host[i], "RGB").save(os.path.join(save_dir, "%s_%s_testFakeExp.png" % (name, tag)))
    out = {"names": names, "psnr": psnr, "ssim": ssim}
    if metrics and names:
        out["mean_psnr"], out["mean_ssim"] = mean_metric(psnr), mean_metric(ssim)
    return out
