#!/usr/bin/env python3
"""Which stored tensors / packed weights of the generator must carry more than 11 significant bits for the fp16 storage mode to put the
enhanced pixels inside north_star's 1e-3 (max-norm, pixels in [-1, 1])?  CPU emulation of the product's arithmetic on the oracle: every
tensor the kernels STORE (including the two the r4 tool left out: the GAM's pre-norm conv result and the low-resolution 1x1 result in front of
the bilinear up-sampling) is rounded to fp16 ('h'), to a hi + lo fp16 PAIR ('hl': hi = fp16(v), lo = fp16(v - hi), ~22 bits -- what the
`precise` mode stores for the full-resolution tensors) or kept in fp32 ('f'); packed weights likewise per layer.  fp32 accumulation everywhere.
Usage: python tools/diag_g_hilo.py [size=512] [batch=2] [smooth|noise]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from oracle import uegan_oracle as O


def r16(t):
    return t.to(torch.float16).to(torch.float32)


def rnd(t, mode):
    if mode == "f":
        return t
    hi = r16(t)
    if mode == "h":
        return hi
    return hi + r16(t - hi)


FULLRES = ["input", "x1", "ga1pre", "ga1", "up4", "y4", "prod", "d50", "res"]
DEEP = ["x2", "x3", "x4", "x5", "ga2pre", "ga2", "ga3pre", "ga3", "ga4pre", "ga4", "ga5pre", "ga5", "up1pre", "up1", "up2pre", "up2", "up3pre", "up3",
        "up4pre", "y1", "y2", "y3"]
THINW = ["enc1", "ga1", "upsample4", "dec4", "dec5"]


def forward(P, x, tm, wm):
    """tm: {tensor: mode} (default 'h'); wm: {layer prefix: mode} (default 'h')"""
    def W(k):
        v = P[k]
        return rnd(v, wm.get(k.split(".")[0], wm.get("*", "h"))) if v.dim() == 4 else v
    r = lambda name, t: rnd(t, tm.get(name, tm.get("*", "h")))

    def cb(prefix, t, stride):
        w, b = W(prefix + ".main.1.weight"), P[prefix + ".main.1.bias"]
        return F.leaky_relu(O.reflect_conv(t, w, b, stride), 0.2)

    def gam(n, t):
        C = t.shape[1]
        w = W(n + ".fuse.0.weight")[:, :C]
        y = r(n + "pre", F.conv2d(t, w))
        return r(n, F.instance_norm(y, eps=O.IN_EPS))

    def up(n, key, t):
        y = r(key + "pre", F.conv2d(t, W(n + ".1.main.1.weight"), P[n + ".1.main.1.bias"]))
        return r(key, F.interpolate(y, scale_factor=2, mode="bilinear", align_corners=True))

    xin = r("input", x)
    x1 = r("x1", cb("enc1", xin, 1)); x2 = r("x2", cb("enc2", x1, 2)); x3 = r("x3", cb("enc3", x2, 2))
    x4 = r("x4", cb("enc4", x3, 2)); x5 = r("x5", cb("enc5", x4, 2))
    x5 = gam("ga5", x5)
    y1 = r("y1", cb("dec1", torch.cat([up("upsample1", "up1", x5), gam("ga4", x4)], 1), 1))
    y2 = r("y2", cb("dec2", torch.cat([up("upsample2", "up2", y1), gam("ga3", x3)], 1), 1))
    y3 = r("y3", cb("dec3", torch.cat([up("upsample3", "up3", y2), gam("ga2", x2)], 1), 1))
    y4 = r("y4", cb("dec4", torch.cat([up("upsample4", "up4", y3), gam("ga1", x1)], 1), 1))
    prod = r("prod", y4 * x1)
    d50 = r("d50", O.reflect_conv(prod, W("dec5.0.main.1.weight"), P["dec5.0.main.1.bias"], 1))
    res = r("res", torch.tanh(O.reflect_conv(d50, W("dec5.1.main.1.weight"), P["dec5.1.main.1.bias"], 1)))
    return torch.clamp(res + x, -1.0, 1.0)


def images(B, S, seed, kind):
    g = torch.Generator().manual_seed(seed)
    if kind == "noise":
        return torch.rand(B, 3, S, S, generator=g) * 2 - 1
    lo = torch.rand(B, 3, S // 32, S // 32, generator=g)            # tests/test_oracle_at_size.py::_images
    x = F.interpolate(lo, size=(S, S), mode="bicubic", align_corners=False) + 0.05 * torch.randn(B, 3, S, S, generator=g)
    return (x.clamp(0, 1) * 2 - 1).contiguous()


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    kind = sys.argv[3] if len(sys.argv) > 3 else "smooth"
    P = O.init_params(O.generator_param_shapes(32), 41, "default")
    x = images(B, S, 41, kind)
    torch.set_num_threads(8)
    with torch.no_grad():
        ref = O.generator_forward(P, x)

        def report(tag, tm, wm):
            out = forward(P, x, tm, wm)
            d = (out - ref)
            fl = 1e-2
            er = float((d.abs() / ref.abs().clamp_min(fl)).max())
            print("%-86s max|err| %.2e  rms %.2e  elem-rel(floor 1e-2) %.2e" % (tag, float(d.abs().max()), float(d.pow(2).mean().sqrt()), er), flush=True)

        print("== %s %d x %dx%d, fp16 storage; error of the enhanced pixels against the fp32 oracle" % (kind, B, S, S))
        report("everything fp16 (today's fp16 mode)", {}, {})
        report("everything fp32 (emulation noise floor)", {"*": "f"}, {"*": "f"})
        full = {k: "hl" for k in FULLRES}
        thin = {k: "hl" for k in THINW}
        report("A. full-resolution tensors hi/lo, thin weights hi/lo", full, thin)
        report("A'. A with y4, up4 plain and the weights of upsample4 plain (what the precise mode stores)", dict(full, y4="f", up4="h"), dict(thin, upsample4="h"))
        if os.environ.get("QUICK"):
            return
        for drop in FULLRES:
            tm = dict(full); tm[drop] = "h"
            report("   A but %-7s plain fp16" % drop, tm, thin)
        for drop in THINW:
            w2 = dict(thin); w2[drop] = "h"
            report("   A but weights of %-9s plain fp16" % drop, full, w2)
        report("B. A + up4pre, x2, ga2pre, ga2, up3, up3pre, y3 (the 256^2 tensors) hi/lo", dict(full, up4pre="hl", x2="hl", ga2pre="hl", ga2="hl", up3="hl", up3pre="hl", y3="hl"), thin)
        report("C. A + all weights hi/lo", full, {"*": "hl"})
        report("D. every activation hi/lo, thin weights hi/lo", {"*": "hl"}, thin)
        report("E. every activation hi/lo, all weights hi/lo", {"*": "hl"}, {"*": "hl"})
        report("F. only the full-resolution tensors hi/lo (weights fp16)", full, {})
        report("G. only thin weights hi/lo", {}, thin)


main()
