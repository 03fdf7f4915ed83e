#!/usr/bin/env python3
"""Which stored tensors of the generator cost the bf16 mode its output accuracy, and what would it take to reach north_star's 1e-3 / 60 dB?
CPU emulation of the product's arithmetic: the oracle's generator with chosen weights and stored activations rounded to bf16 (or to a hi+lo
bf16 pair = 16 mantissa bits) where the kernels store them, fp32 accumulation everywhere.  Prints, against the fp32 oracle:
  (1) the error POWER (mean squared error x 1e6) each stored tensor contributes on its own -- they add up to the total;
  (2) cumulative remedies, cheapest first, with PSNR after 8-bit quantisation (tester.py) and the max / rms pixel error.
Usage: python tools/diag_g_bf16.py [size=256] [noise|smooth]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import uegan_oracle as O

def rb(t): return t.to(torch.bfloat16).to(torch.float32)
def rb2(t):                      # hi + lo bf16 pair
    hi = rb(t)
    return hi + rb(t - hi)

TENSORS = ["input", "x1", "x2-5", "ga1", "ga2-5", "up1-3", "up4", "y1-3", "y4", "prod", "d50", "res"]
WGROUPS = {"thin": ("enc1", "dec4", "dec5", "ga1", "upsample4"), "rest": None}

def forward(P, x, fp32_t=(), w_mode=None):
    """fp32_t: tensors kept in fp32 (everything else rounded to bf16); w_mode: {layer prefix: 'f32' | 'hilo'} (default bf16)"""
    w_mode = w_mode or {}
    def wq(k, v):
        if v.dim() != 4: return v
        m = w_mode.get(k.split(".")[0], w_mode.get("*", "bf16"))
        return v if m == "f32" else (rb2(v) if m == "hilo" else rb(v))
    Pw = {k: wq(k, v) for k, v in P.items()}
    r = lambda name, t: t if name in fp32_t else rb(t)
    xin = r("input", x)
    x1 = r("x1", O.conv_block(Pw, "enc1", xin, 1)); x2 = r("x2-5", O.conv_block(Pw, "enc2", x1, 2)); x3 = r("x2-5", O.conv_block(Pw, "enc3", x2, 2))
    x4 = r("x2-5", O.conv_block(Pw, "enc4", x3, 2)); x5 = r("x2-5", O.conv_block(Pw, "enc5", x4, 2))
    g = lambda n, t: r("ga1" if n == "ga1" else "ga2-5", O.gam(Pw, n, t))
    u = lambda n, t: r("up4" if n == "upsample4" else "up1-3", O.upsample_conv(Pw, n, t))
    x5 = g("ga5", x5)
    y1 = r("y1-3", O.conv_block(Pw, "dec1", torch.cat([u("upsample1", x5), g("ga4", x4)], 1), 1))
    y2 = r("y1-3", O.conv_block(Pw, "dec2", torch.cat([u("upsample2", y1), g("ga3", x3)], 1), 1))
    y3 = r("y1-3", O.conv_block(Pw, "dec3", torch.cat([u("upsample3", y2), g("ga2", x2)], 1), 1))
    y4 = r("y4", O.conv_block(Pw, "dec4", torch.cat([u("upsample4", y3), g("ga1", x1)], 1), 1))
    prod = r("prod", y4 * x1)
    d50 = r("d50", O.sn_conv(Pw, "dec5.0", prod))
    res = r("res", torch.tanh(O.sn_conv(Pw, "dec5.1", d50)))
    return torch.clamp(res + x, -1.0, 1.0)

def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    kind = sys.argv[2] if len(sys.argv) > 2 else "smooth"
    P = O.init_params(O.generator_param_shapes(32), 41, "default")
    g = torch.Generator().manual_seed(1990)
    if kind == "noise":
        x = torch.rand(1, 3, S, S, generator=g) * 2 - 1
    else:                                        # FiveK-shaped: low-pass filtered noise (tests/test_parity_full.py::_smooth_images)
        lo = torch.rand(1, 3, S // 32, S // 32, generator=g)
        x = torch.nn.functional.interpolate(lo, size=(S, S), mode="bicubic", align_corners=False) + 0.03 * torch.randn(1, 3, S, S, generator=g)
        x = (x.clamp(0, 1) * 2 - 1).contiguous()
    ALL = set(TENSORS)
    F32W = {"*": "f32"}
    with torch.no_grad():
        ref = O.generator_forward(P, x)
        q = lambda t: O.to_uint8_image(t)[0].numpy()
        def stat(out):
            d = out - ref
            return O.psnr_u8(q(out), q(ref)), float(d.abs().max()), float(d.pow(2).mean().sqrt()), float(d.pow(2).mean()) * 1e6
        def report(tag, out):
            p, mx, rms, pw = stat(out)
            print("%-64s PSNR %6.2f dB  max|err| %.4f  rms %.5f  power %.3f" % (tag, p, mx, rms, pw))
            return pw
        print("== input: %s %dx%d; error against the fp32 oracle (enhanced pixels in [-1, 1])" % (kind, S, S))
        tot = report("everything bf16 (the product's bf16 mode)", forward(P, x))
        report("bf16 weights only (all activations fp32)", forward(P, x, ALL))
        report("bf16 activations only (all weights fp32)", forward(P, x, (), F32W))
        print("-- (1) error power of ONE rounded tensor group, everything else fp32 (x 1e-6; additive)")
        acc = 0.0
        for t in TENSORS:
            acc += report("  only %-6s rounded" % t, forward(P, x, ALL - {t}, F32W))
        for name in ("thin", "rest"):
            wm = {"*": "f32"}
            if name == "thin":
                wm.update({k: "bf16" for k in WGROUPS["thin"]})
            else:
                wm = {"*": "bf16"}
                wm.update({k: "f32" for k in WGROUPS["thin"]})
            acc += report("  only the weights of %s rounded" % ("enc1/dec4/dec5/ga1/upsample4" if name == "thin" else "every other layer"), forward(P, x, ALL, wm))
        print("  sum of the parts %.3f vs everything rounded %.3f" % (acc, tot))
        print("-- (2) cumulative remedies")
        thin_hilo = {k: "hilo" for k in WGROUPS["thin"]}
        report("a. + hi/lo input (free: the 5 padding channels of enc1's 8-channel rows)", forward(P, x, {"input"}))
        report("b. + hi/lo weights on the thin full-resolution layers (2 MFMAs per fragment)", forward(P, x, {"input"}, thin_hilo))
        report("c. + res kept in fp32 (residual + clamp in dec5.1's epilogue)", forward(P, x, {"input", "res"}, thin_hilo))
        report("d. + prod = y4*x1 from the fp32 accumulator (mul in dec4's epilogue)", forward(P, x, {"input", "res", "y4"}, thin_hilo))
        report("e. + d50 stored as fp32 / hi+lo (2x bytes of one 32-ch tensor)", forward(P, x, {"input", "res", "y4", "d50"}, thin_hilo))
        report("f. + prod stored as fp32 / hi+lo", forward(P, x, {"input", "res", "y4", "d50", "prod"}, thin_hilo))
        report("g. + x1 stored as fp32 / hi+lo", forward(P, x, {"input", "res", "y4", "d50", "prod", "x1"}, thin_hilo))
        report("h. + ga1, up4 stored as fp32 / hi+lo (= every full-resolution tensor)", forward(P, x, {"input", "res", "y4", "d50", "prod", "x1", "ga1", "up4"}, thin_hilo))
        report("i. h with hi/lo weights everywhere", forward(P, x, {"input", "res", "y4", "d50", "prod", "x1", "ga1", "up4"}, {"*": "hilo"}))
        report("   fp16 storage of everything instead (11 mantissa bits, same bytes as bf16)",
               torch.clamp(_fp16_forward(P, x), -1, 1))

def _fp16_forward(P, x):
    h = lambda t: t.to(torch.float16).to(torch.float32)
    Pw = {k: (h(v) if v.dim() == 4 else v) for k, v in P.items()}
    xin = h(x)
    x1 = h(O.conv_block(Pw, "enc1", xin, 1)); x2 = h(O.conv_block(Pw, "enc2", x1, 2)); x3 = h(O.conv_block(Pw, "enc3", x2, 2))
    x4 = h(O.conv_block(Pw, "enc4", x3, 2)); x5 = h(O.conv_block(Pw, "enc5", x4, 2))
    g = lambda n, t: h(O.gam(Pw, n, t)); u = lambda n, t: h(O.upsample_conv(Pw, n, t))
    x5 = g("ga5", x5)
    y1 = h(O.conv_block(Pw, "dec1", torch.cat([u("upsample1", x5), g("ga4", x4)], 1), 1))
    y2 = h(O.conv_block(Pw, "dec2", torch.cat([u("upsample2", y1), g("ga3", x3)], 1), 1))
    y3 = h(O.conv_block(Pw, "dec3", torch.cat([u("upsample3", y2), g("ga2", x2)], 1), 1))
    y4 = h(O.conv_block(Pw, "dec4", torch.cat([u("upsample4", y3), g("ga1", x1)], 1), 1))
    d50 = h(O.sn_conv(Pw, "dec5.0", h(y4 * x1)))
    return h(torch.tanh(O.sn_conv(Pw, "dec5.1", d50))) + x

main()
