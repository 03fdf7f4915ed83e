#!/usr/bin/env python3
"""The discriminator's share of a training step in isolation: the batched D-update pass (3 image groups, forward + backward + Adam) and the
adversarial pass of the G update (2 groups, frozen D, data gradient of the fake group only), as Trainer.train_step issues them.
Usage: python tools/bench_d.py [--iters 5]   (run under rocprofv3 --kernel-trace --stats for the per-kernel table)"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import uegan_amd
from uegan_amd import fused, models, ops, trainer
ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--size", type=int, default=512)
args = ap.parse_args()
dev = torch.device("cuda:0")
uegan_amd.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
D = models.Discriminator(32, "none", "LeakyReLU", True, "rahinge").to(dev)
opt = ops.FusedAdamL2(D.parameters(), 4e-4, (0.5, 0.999), 1e-8, 1e-4)
B, S = args.batch, args.size
imgs = [(torch.rand(B, 3, S, S, device=dev) * 2 - 1) for _ in range(3)]
fake = (torch.rand(B, 3, S, S, device=dev) * 2 - 1).requires_grad_(True)
def d_update():
    opt.zero_grad()
    loss = fused.discriminator_loss(D, imgs, [(0, 1), (0, 2)], True)
    loss.backward()
    opt.step(1.0)
def g_adv():
    with trainer._Frozen(D):
        adv = fused.discriminator_loss(D, [imgs[0], fake], [(0, 1)], False)
    adv.backward()
    fake.grad = None
for name, fn in (("d_update", d_update), ("g_adv", g_adv)):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(args.iters): fn()
    torch.cuda.synchronize()
    print("%-10s %.3f ms" % (name, (time.perf_counter() - t) / args.iters * 1e3))
