#!/usr/bin/env python3
"""Timing sweeps of one 3x3 stride-1 zero-padded bf16 convolution (the wide-tile kernel's regime): input channels x batch, to separate the
fixed per-block cost (prologue, epilogue) from the per-chunk cost of the main loop.  Usage: python tools/bench_wide.py [--n 512] [--hw 64]"""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from uegan_amd import _lib as L, ops
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=512)
ap.add_argument("--hw", type=int, default=64)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--cs", default="64,128,256,512,1024")
ap.add_argument("--bs", default="16,32")
ap.add_argument("--zero", action="store_true", help="zero-filled operands (DVFS check)")
args = ap.parse_args()
dev = torch.device("cuda:0"); lib = L.load(); dt = torch.bfloat16
def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
print("%6s %4s %10s %10s" % ("C", "B", "ms", "TFLOP/s"))
for B in [int(v) for v in args.bs.split(",")]:
    for Cin in [int(v) for v in args.cs.split(",")]:
        H = args.hw
        x = (torch.zeros if args.zero else torch.randn)(B, H, H, Cin, device=dev).to(dt)
        w = (torch.zeros if args.zero else torch.randn)(args.n, Cin, 3, 3, device=dev) * 0.05
        b = torch.zeros(args.n, device=dev)
        cfg = ops.ConvCfg(1, ops.PAD_ZERO, 2)
        d = ops._desc(x, None, w, cfg)
        ohwi, ihwo = cfg.packed.get(w, dt, d.C1 + d.C2, d.Cout)
        y = torch.empty(B, d.Ho, d.Wo, d.Cout, device=dev, dtype=dt)
        st = torch.cuda.current_stream().cuda_stream
        p = ops._p
        t = timeit(lambda: L.check(lib.uegan_conv2d_fwd(C.byref(d), p(x), None, p(ohwi), p(b), None, p(y), st)), args.iters)
        fl = 2.0 * B * H * H * args.n * 9 * Cin
        print("%6d %4d %10.4f %10.1f" % (Cin, B, t, fl / t / 1e9))
