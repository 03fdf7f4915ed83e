#!/usr/bin/env python3
"""uegan_sn_act_bwd_p with plain vs padded-grid gradients at the discriminator's map sizes (batch 48 = 3 groups x 16)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import uegan_amd
from uegan_amd import ops, _lib
dev = torch.device("cuda:0")
uegan_amd.set_compute_dtype(torch.bfloat16)
lib = _lib.load()
P = ops._p
for (H, C, pg, pg2) in [(256, 32, 3, 0), (256, 32, 0, 0), (128, 64, 3, 3), (128, 64, 0, 0), (64, 128, 3, 2), (64, 128, 0, 0), (32, 256, 2, 2), (32, 256, 0, 0), (16, 512, 2, 0), (16, 512, 0, 0)]:
    nb, ng = 16, 3
    B = nb * ng
    y = torch.randn(B, H, H, C, device=dev).to(torch.bfloat16)
    g = torch.randn(B, H + 2 * pg, H + 2 * pg, C, device=dev).to(torch.bfloat16)
    g2 = torch.randn(B, H + 2 * pg2, H + 2 * pg2, C, device=dev).to(torch.bfloat16)
    dz = torch.empty_like(y)
    bias = torch.zeros(C, device=dev)
    inv = torch.ones(ng, device=dev)
    ws = torch.empty(lib.uegan_sn_act_bwd_workspace_floats(ng, C), device=dev)
    def run():
        r = lib.uegan_sn_act_bwd_p(1, 1, P(g), pg, P(g2), pg2, P(y), P(bias), C, P(inv), P(dz), P(ws), nb * H * H, H, H, C, ng, None)
        assert r > 0, r
    run(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20): run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 20
    mb = (g.numel() + g2.numel() + 2 * y.numel()) * 2 / 1e6
    print("H %4d C %4d pad_g %d pad_g2 %d: %.1f us  %.0f MB  %.2f TB/s" % (H, C, pg, pg2, dt * 1e6, mb, mb / dt / 1e6))
