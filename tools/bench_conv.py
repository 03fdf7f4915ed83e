#!/usr/bin/env python3
"""Per-layer micro-benchmark of the convolution kernels (forward / dgrad / wgrad) on the real layer shapes of
G, D and VGG19 at 512x512 batch 16.  Prints algorithmic TFLOP/s per layer and the time-weighted total.
Usage: python tools/bench_conv.py [--dtype bf16|f32] [--iters 5] [--filter substr|substr]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from uegan_amd import _lib as L, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--filter", default="")
ap.add_argument("--impl", type=int, default=0, help="0/1 MFMA+glds, 3 MFMA+register staging, 2 direct")
ap.add_argument("--abl", type=int, default=0, help="timing ablation bits of the streaming / weight-gradient kernels (tools build: tools/build_tools.sh)")
ap.add_argument("--wide-abl", type=int, default=0, help="timing ablation variant of conv_wide_kernel (tools build)")
ap.add_argument("--tune", default="", help="knob=value[,knob=value]: uegan_set_tuning by index (include/uegan_hip.h), e.g. 4=-1 switches conv_tall_kernel off")
ap.add_argument("--kernels", action="store_true", help="also list the kernels (library profiler) each of fwd / dgrad / wgrad launches")
args = ap.parse_args()
dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
dev = torch.device("cuda:0")
if args.abl or args.wide_abl:      # the ablation kernels exist only in the tools build of the library
    lib = L.load(os.path.join(ROOT, "tools", "_build", "libuegan_hip_tools.so"))
    lib.uegan_tools_set_ablation(args.abl, args.wide_abl)
else:
    lib = L.load()
lib.uegan_set_conv_impl(args.impl)
for kv in filter(None, args.tune.split(",")):
    L.check(lib.uegan_set_tuning(int(kv.split("=")[0]), int(kv.split("=")[1]), None))
B, S = args.batch, args.size

# name, H(in), C1, C2, Cout, k, stride, pad_mode, act, count of (fwd, dgrad, wgrad) per train step
LAYERS = []
def add(name, H, C1, C2, Co, k, s, pm, act, nf, nd, nw):
    LAYERS.append((name, H, C1, C2, Co, k, s, pm, act, nf, nd, nw))
R, Z = ops.PAD_REFLECT, ops.PAD_ZERO
# Generator (2 fwd, 2 bwd per step); enc1 has no dgrad (image input)
add("G.enc1 7x7 3->32", S, 3, 0, 32, 7, 1, R, 1, 2, 0, 2)
add("G.enc2 3x3s2 32->64", S, 32, 0, 64, 3, 2, R, 1, 2, 2, 2)
add("G.enc3 3x3s2 64->128", S // 2, 64, 0, 128, 3, 2, R, 1, 2, 2, 2)
add("G.enc4 3x3s2 128->256", S // 4, 128, 0, 256, 3, 2, R, 1, 2, 2, 2)
add("G.enc5 3x3s2 256->512", S // 8, 256, 0, 512, 3, 2, R, 1, 2, 2, 2)
for i, (c, h) in enumerate(((512, S // 16), (256, S // 8), (128, S // 4), (64, S // 2), (32, S))):
    add("G.ga%d 1x1 %d->%d" % (5 - i, c, c), h, c, 0, c, 1, 1, R, 0, 2, 2, 2)
for i, (c, h) in enumerate(((512, S // 16), (256, S // 8), (128, S // 4), (64, S // 2))):
    add("G.up%d 1x1 %d->%d" % (i + 1, c, c // 2), h, c, 0, c // 2, 1, 1, R, 0, 2, 2, 2)
    add("G.dec%d 3x3 %d+%d->%d" % (i + 1, c // 2, c // 2, c // 2), h * 2, c // 2, c // 2, c // 2, 3, 1, R, 1, 2, 2, 2)
add("G.dec5.0 3x3 32->32", S, 32, 0, 32, 3, 1, R, 0, 2, 2, 2)
add("G.dec5.1 7x7 32->3", S, 32, 0, 3, 7, 1, R, 3, 2, 2, 2)
# Discriminator: 5 fwd; bwd: 3 full (D step) + 1 dgrad-only (G step, fake branch)
cin, h = 3, S
for i, (k, m) in enumerate(zip((7, 7, 7, 5, 5), (1, 2, 4, 8, 16))):
    co = 32 * m
    add("D.d%d %dx%ds2 %d->%d" % (i + 1, k, k, cin, co), h, cin, 0, co, k, 2, R, 1, 5, 4 if i else 1, 3)
    add("D.d%d_pred %dx%d %d->1" % (i + 1, k, k, co), h // 2, co, 0, 1, k, 1, R, 3, 5, 4, 3)
    cin, h = co, h // 2
# VGG19 through relu5_1: 2 fwd + 1 dgrad, no wgrad
cfg = [(3, 64, S), (64, 64, S), (64, 128, S // 2), (128, 128, S // 2), (128, 256, S // 4), (256, 256, S // 4), (256, 256, S // 4), (256, 256, S // 4),
       (256, 512, S // 8), (512, 512, S // 8), (512, 512, S // 8), (512, 512, S // 8), (512, 512, S // 16)]
for i, (ci, co, hh) in enumerate(cfg):
    add("VGG.conv%d 3x3 %d->%d" % (i, ci, co), hh, ci, 0, co, 3, 1, Z, 2, 2, 1 if i else 0, 0)


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


tot_ms = tot_fl = 0.0
print("%-28s %9s %9s %9s   %8s %8s %8s  ms/step" % ("layer", "fwd ms", "dgrad ms", "wgrad ms", "fwd TF", "dgrad TF", "wgrad TF"))
for (name, H, C1, C2, Co, k, s, pm, act, nf, nd, nw) in LAYERS:
    if args.filter and not any(f in name for f in args.filter.split("|")):
        continue
    C1p = ops.cpad(C1, dt)
    x1 = torch.randn(B, H, H, C1p, device=dev).to(dt)
    x2 = torch.randn(B, H, H, C2, device=dev).to(dt) if C2 else None
    w = (torch.randn(Co, C1 + C2, k, k, device=dev) * 0.05)
    b = torch.zeros(Co, device=dev)
    cfg_ = ops.ConvCfg(s, pm, act)
    d = ops._desc(x1, x2, w, cfg_)
    ohwi, ihwo = cfg_.packed.get(w, dt, d.C1 + d.C2, d.Cout)
    y = torch.empty(B, d.Ho, d.Wo, d.Cout, device=dev, dtype=dt)
    dz = torch.randn(B, d.Ho, d.Wo, d.Cout, device=dev).to(dt)
    dx1 = torch.empty_like(x1)
    dx2 = torch.empty_like(x2) if C2 else None
    wsb = lib.uegan_conv2d_wgrad_workspace_bytes(C.byref(d))
    ws = torch.empty(max(wsb // 4, 1), device=dev)
    dw = torch.empty_like(w)
    db = torch.empty(Co, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    p = ops._p
    flops = 2.0 * B * d.Ho * d.Wo * Co * k * k * (C1 + C2)
    tf = timeit(lambda: L.check(lib.uegan_conv2d_fwd(C.byref(d), p(x1), p(x2), p(ohwi), p(b), None, p(y), st)), args.iters)
    dwsb = lib.uegan_conv2d_dgrad_workspace_bytes(C.byref(d))
    dws = torch.empty(dwsb // 4 + 1, dtype=torch.float32, device=dev)
    td = timeit(lambda: L.check(lib.uegan_conv2d_dgrad_ws(C.byref(d), p(dz), p(ihwo), None, p(dx1), p(dx2), p(dws), dwsb, st)), args.iters) if nd else 0.0
    tw = timeit(lambda: L.check(lib.uegan_conv2d_wgrad(C.byref(d), p(x1), p(x2), p(dz), None, p(dw), p(db), p(ws), wsb, st)), args.iters) if nw else 0.0
    twk = 0.0
    if nw:      # kernel-only time of the main wgrad kernel (library profiler: HIP events around that launch)
        L.check(lib.uegan_profile_begin(16))
        for _ in range(3):
            L.check(lib.uegan_conv2d_wgrad(C.byref(d), p(x1), p(x2), p(dz), None, p(dw), p(db), p(ws), wsb, st))
        torch.cuda.synchronize()
        ents = (L.ProfileEntry * 8)()
        nn = C.c_int(0)
        L.check(lib.uegan_profile_end(ents, 8, C.byref(nn)))
        if nn.value:
            twk = ents[0].total_ms / max(ents[0].launches, 1)
    if args.kernels:
        for tag, fn in (("fwd", lambda: lib.uegan_conv2d_fwd(C.byref(d), p(x1), p(x2), p(ohwi), p(b), None, p(y), st)),
                        ("dgrad", (lambda: lib.uegan_conv2d_dgrad_ws(C.byref(d), p(dz), p(ihwo), None, p(dx1), p(dx2), p(dws), dwsb, st)) if nd else None),
                        ("wgrad", (lambda: lib.uegan_conv2d_wgrad(C.byref(d), p(x1), p(x2), p(dz), None, p(dw), p(db), p(ws), wsb, st)) if nw else None)):
            if fn is None:
                continue
            L.check(lib.uegan_profile_begin(32))
            for _ in range(3):
                L.check(fn())
            torch.cuda.synchronize()
            ents = (L.ProfileEntry * 16)()
            nn = C.c_int(0)
            L.check(lib.uegan_profile_end(ents, 16, C.byref(nn)))
            print("    %-6s " % tag + "; ".join("%s x%d %.3f ms" % (ents[i].name.decode(), ents[i].launches // 3, ents[i].total_ms / max(ents[i].launches, 1)) for i in range(nn.value)))
    ms = nf * tf + nd * td + nw * tw
    tot_ms += ms
    tot_fl += flops * (nf + nd + nw)
    g = lambda t: flops / (t * 1e-3) / 1e12 if t else 0.0
    print("%-28s %9.3f %9.3f %9.3f   %8.1f %8.1f %8.1f  %7.2f   wgrad kernel only %.3f" % (name, tf, td, tw, g(tf), g(td), g(tw), ms, twk))
print("TOTAL conv ms/step %.2f  algorithmic %.2f TFLOP/step -> %.1f TFLOP/s" % (tot_ms, tot_fl / 1e12, tot_fl / (tot_ms * 1e-3) / 1e12))
