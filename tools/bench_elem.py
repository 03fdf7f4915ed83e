#!/usr/bin/env python3
"""Bandwidth of the elementwise / normalisation kernels at full-resolution shapes (bf16, 16 x 512 x 512 x 32 = 268 MB tensors).
Prints effective GB/s = algorithmic bytes (reads + writes) / time.  Usage: python tools/bench_elem.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import uegan_amd  # noqa: E402
from uegan_amd import ops  # noqa: E402

uegan_amd.set_compute_dtype(torch.bfloat16)
dev = torch.device("cuda:0")
B, H, W, C = 16, 512, 512, 32
dt = torch.bfloat16


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


x = torch.randn(B, H, W, C, device=dev).to(dt)
y = torch.randn(B, H, W, C, device=dev).to(dt)
nb = x.numel() * 2
lib = ops.lib()
st = torch.cuda.current_stream().cuda_stream
p = ops._p
out = torch.empty_like(x)
out2 = torch.empty_like(x)
from uegan_amd import _lib as L  # noqa: E402
rows = []
rows.append(("act_bwd (2r+1w)", 3 * nb, lambda: L.check(lib.uegan_act_bwd(1, 1, p(x), p(y), p(out), x.numel(), st))))
rows.append(("mul_fwd (2r+1w)", 3 * nb, lambda: L.check(lib.uegan_mul_fwd(1, p(x), p(y), p(out), x.numel(), st))))
rows.append(("mul_bwd (3r+2w)", 5 * nb, lambda: L.check(lib.uegan_mul_bwd(1, p(x), p(y), p(out), p(out), p(out2), x.numel(), st))))
rows.append(("torch add bf16 (2r+1w)", 3 * nb, lambda: torch.add(x, y, out=out)))
rows.append(("torch copy (1r+1w)", 2 * nb, lambda: out.copy_(x)))
xs = x.clone().requires_grad_(True)
rows.append(("instnorm fwd (2r+1w)", 3 * nb, lambda: ops.instnorm(xs.detach())))
yn = ops.instnorm(xs)
rows.append(("instnorm bwd (3r+1w)", 4 * nb, lambda: torch.autograd.grad(yn, xs, y, retain_graph=True)))
xh = torch.randn(B, H // 2, W // 2, C * 2, device=dev).to(dt).requires_grad_(True)
rows.append(("upsample2x fwd (1r+4w) 64ch @256", xh.numel() * 2 * 5, lambda: ops.upsample2x(xh.detach())))
yu = ops.upsample2x(xh)
gu = torch.randn_like(yu)
rows.append(("upsample2x bwd (4r+1w)", xh.numel() * 2 * 5, lambda: torch.autograd.grad(yu, xh, gu, retain_graph=True)))
xv = torch.randn(B, H, W, 64, device=dev).to(dt).requires_grad_(True)
rows.append(("maxpool fwd (4r+1w) 64ch @512", xv.numel() * 2 * 1.25, lambda: ops.maxpool2x2(xv.detach())))
ym = ops.maxpool2x2(xv)
gm = torch.randn_like(ym)
rows.append(("maxpool bwd (4r+1r+4w)", xv.numel() * 2 * 2.25, lambda: torch.autograd.grad(ym, xv, gm, retain_graph=True)))
img = torch.randn(B, 3, H, W, device=dev)
rows.append(("to_nhwc 3->8ch (fp32 in, bf16 out)", img.numel() * 4 + B * H * W * 8 * 2, lambda: ops.to_nhwc(img, dt)))
for name, byts, fn in rows:
    t = timeit(fn)
    print("%-40s %8.3f ms  %7.1f GB/s" % (name, t, byts / t / 1e6))
