#!/usr/bin/env python3
"""Diagnostic (GPU): where does the bf16 VGG fidelity gradient lose agreement with fp32?  Per tap: forward error, and the
gradient of a single-tap loss w.r.t. the image (cosine vs the fp32 HIP path)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from uegan_amd import losses, ops

dev = torch.device("cuda:0")
z = np.load(os.path.join(ROOT, "tests", "golden", "percep_full.npz"))
for B, S, src in ((1, 64, "fixture"), (2, 128, "rand"), (4, 256, "rand")):
    if src == "fixture":
        x0, y0 = torch.from_numpy(z["x"]).to(dev), torch.from_numpy(z["y"]).to(dev)
    else:
        g = torch.Generator().manual_seed(5)
        x0, y0 = torch.rand(B, 3, S, S, generator=g).to(dev), torch.rand(B, 3, S, S, generator=g).to(dev)
    res = {}
    for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        ops.set_compute_dtype(dt)
        P = losses.PerceptualLoss(vgg_weights="seeded").to(dev)
        taps = [t.float() for t in P._taps(x0, 1.0, 0.0)]
        grads = []
        for i in range(5):
            w = [0.0] * 5
            w[i] = 1.0
            P.weights = w
            x = x0.clone().requires_grad_(True)
            l = P(x, y0)
            l.backward()
            grads.append((float(l), x.grad.clone()))
        P.weights = [1.0 / 64, 1.0 / 64, 1.0 / 32, 1.0 / 32, 1.0]
        x = x0.clone().requires_grad_(True)
        l = P(x, y0)
        l.backward()
        res[name] = (taps, grads, float(l), x.grad.clone())
    print("== B=%d S=%d (%s): total loss f32 %.6f bf16 %.6f  grad cos %.5f norm ratio %.4f" % (
        B, S, src, res["f32"][2], res["bf16"][2],
        float((res["f32"][3] * res["bf16"][3]).sum() / (res["f32"][3].norm() * res["bf16"][3].norm())),
        float(res["bf16"][3].norm() / res["f32"][3].norm())))
    for i in range(5):
        ta, tb = res["f32"][0][i], res["bf16"][0][i]
        ga, gb = res["f32"][1][i][1], res["bf16"][1][i][1]
        print("  tap %d %s: fwd rel-rms %.4f | single-tap loss f32 %.6f bf16 %.6f | dx cos %.5f norm ratio %.4f" % (
            i, tuple(ta.shape), float((ta - tb).norm() / ta.norm()), res["f32"][1][i][0], res["bf16"][1][i][0],
            float((ga * gb).sum() / (ga.norm() * gb.norm() + 1e-30)), float(gb.norm() / (ga.norm() + 1e-30))))
