#!/usr/bin/env python3
"""Step time of the benchmark configuration under the Trainer's stream schedules (overlap / early_taps), on one box."""
import os, sys, random, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import uegan_amd
from uegan_amd import losses, models, trainer
dev = torch.device("cuda:0")
uegan_amd.set_compute_dtype(torch.bfloat16)
g = torch.Generator().manual_seed(1990)
raw = (torch.rand(16, 3, 512, 512, generator=g) * 2 - 1).to(dev)
exp = (torch.rand(16, 3, 512, 512, generator=g) * 2 - 1).to(dev)
P = losses.PerceptualLoss(vgg_weights="seeded").to(dev)
for rep in range(2):
    for kw in (dict(), dict(overlap=False), dict(early_taps=True)):
        torch.manual_seed(1990)
        G = models.Generator(32, "none", "LeakyReLU", False).to(dev)
        D = models.Discriminator(32, "none", "LeakyReLU", True, "rahinge").to(dev)
        T = trainer.Trainer(G, D, P, pool_size=50, rng=random.Random(1990), **kw)
        for _ in range(3): T.train_step(raw, exp)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): T.train_step(raw, exp)
        torch.cuda.synchronize()
        print("[%d] %-22s %.3f ms/step" % (rep, kw or "default (overlap)", (time.perf_counter() - t) / 10 * 1e3), flush=True)
        del T, G, D
