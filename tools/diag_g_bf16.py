#!/usr/bin/env python3
"""Which stored tensors of the generator cost the bf16 mode its output accuracy?  CPU emulation: the oracle's generator with every weight rounded
to bf16 and a chosen set of activations rounded to bf16 where the product stores them (fp32 accumulation everywhere, as in the kernels).
Prints PSNR (8-bit images) / max abs error against the fp32 oracle for: all tensors rounded, and all-but-one-group kept in fp32."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from oracle import uegan_oracle as O

def rb(t): return t.to(torch.bfloat16).to(torch.float32)

GROUPS = ["input", "enc", "gam", "up", "dec123", "y4", "prod", "d50", "res", "weights"]

def forward(P, x, keep):
    r = lambda name, t: t if name in keep else rb(t)
    Pw = P if "weights" in keep else {k: (rb(v) if v.dim() == 4 else v) for k, v in P.items()}
    xin = r("input", x)
    x1 = r("enc", O.conv_block(Pw, "enc1", xin, 1)); x2 = r("enc", O.conv_block(Pw, "enc2", x1, 2)); x3 = r("enc", O.conv_block(Pw, "enc3", x2, 2))
    x4 = r("enc", O.conv_block(Pw, "enc4", x3, 2)); x5 = r("enc", O.conv_block(Pw, "enc5", x4, 2))
    g = lambda n, t: r("gam", O.gam(Pw, n, t))
    x5 = g("ga5", x5)
    u = lambda n, t: r("up", O.upsample_conv(Pw, n, t))
    y1 = r("dec123", O.conv_block(Pw, "dec1", torch.cat([u("upsample1", x5), g("ga4", x4)], 1), 1))
    y2 = r("dec123", O.conv_block(Pw, "dec2", torch.cat([u("upsample2", y1), g("ga3", x3)], 1), 1))
    y3 = r("dec123", O.conv_block(Pw, "dec3", torch.cat([u("upsample3", y2), g("ga2", x2)], 1), 1))
    y4 = r("y4", O.conv_block(Pw, "dec4", torch.cat([u("upsample4", y3), g("ga1", x1)], 1), 1))
    prod = r("prod", y4 * x1)
    d50 = r("d50", O.sn_conv(Pw, "dec5.0", prod))
    res = r("res", torch.tanh(O.sn_conv(Pw, "dec5.1", d50)))
    return torch.clamp(res + x, -1.0, 1.0)

def main():
    torch.manual_seed(0)
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    P = O.init_params(O.generator_param_shapes(32), 41, "default")
    g = torch.Generator().manual_seed(1990)
    x = torch.rand(1, 3, S, S, generator=g) * 2 - 1
    with torch.no_grad():
        ref = O.generator_forward(P, x)
        q = lambda t: O.to_uint8_image(t)[0].numpy()
        def report(tag, out):
            print("%-28s PSNR %.2f dB   max|err| %.4f   rms %.5f" % (tag, O.psnr_u8(q(out), q(ref)), float((out - ref).abs().max()), float((out - ref).pow(2).mean().sqrt())))
        report("all rounded", forward(P, x, set()))
        for k in GROUPS:
            report("fp32: " + k, forward(P, x, {k}))
        report("fp32: prod+d50+res", forward(P, x, {"prod", "d50", "res"}))
        report("fp32: y4+prod+d50+res", forward(P, x, {"y4", "prod", "d50", "res"}))
        report("fp32: weights+d50+res", forward(P, x, {"weights", "d50", "res"}))
        report("only weights rounded", forward(P, x, set(GROUPS) - {"weights"}))

main()


def per_layer_weights():
    """all activations fp32; weights of ONE layer group rounded at a time, and all-but-one"""
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    P = O.init_params(O.generator_param_shapes(32), 41, "default")
    g = torch.Generator().manual_seed(1990)
    x = torch.rand(1, 3, S, S, generator=g) * 2 - 1
    names = sorted({k.split(".")[0] for k, v in P.items() if v.dim() == 4})
    with torch.no_grad():
        ref = O.generator_forward(P, x)
        q = lambda t: O.to_uint8_image(t)[0].numpy()
        for n in names:
            Pw = {k: (rb(v) if (v.dim() == 4 and k.split(".")[0] == n) else v) for k, v in P.items()}
            out = O.generator_forward(Pw, x)
            print("weights of %-10s rounded: PSNR %.2f dB  rms %.5f" % (n, O.psnr_u8(q(out), q(ref)), float((out - ref).pow(2).mean().sqrt())))

if len(sys.argv) > 2:
    per_layer_weights()
