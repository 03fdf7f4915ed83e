#!/usr/bin/env python3
"""Step time of the benchmark configuration under different launch-variant thresholds (uegan_set_tuning), on one box.
Usage: python tools/sweep_tuning.py "0=256,4=192" "0=128" ...   (knob=value pairs per setting; knob indices from include/uegan_hip.h)"""
import os, sys, random, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import uegan_amd
from uegan_amd import _lib, losses, models, trainer
dev = torch.device("cuda:0")
uegan_amd.set_compute_dtype(torch.bfloat16)
lib = _lib.load()
torch.manual_seed(1990)
G = models.Generator(32, "none", "LeakyReLU", False).to(dev)
D = models.Discriminator(32, "none", "LeakyReLU", True, "rahinge").to(dev)
T = trainer.Trainer(G, D, losses.PerceptualLoss(vgg_weights="seeded").to(dev), pool_size=50, rng=random.Random(1990))
g = torch.Generator().manual_seed(1990)
raw = (torch.rand(16, 3, 512, 512, generator=g) * 2 - 1).to(dev)
exp = (torch.rand(16, 3, 512, 512, generator=g) * 2 - 1).to(dev)
DEFAULTS = {0: 256, 1: -1, 2: 0, 3: 192, 4: 192, 5: 0}
def run(n):
    for _ in range(3): T.train_step(raw, exp)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): T.train_step(raw, exp)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
settings = sys.argv[1:] or [""]
for rep in range(2):
    for s in [""] + settings:
        for k, v in DEFAULTS.items(): _lib.check(lib.uegan_set_tuning(k, v, None))
        for kv in filter(None, s.split(",")):
            _lib.check(lib.uegan_set_tuning(int(kv.split("=")[0]), int(kv.split("=")[1]), None))
        print("[%d] %-24s %.3f ms/step" % (rep, s or "defaults", run(10)), flush=True)
