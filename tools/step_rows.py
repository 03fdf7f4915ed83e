#!/usr/bin/env python3
"""One instrumented training step (bf16, 16 x 512 x 512): every convolution-kernel instantiation with its launches, time, algorithmic
TFLOP/s and GB/s (uegan_profile_begin / _end).  Usage: python tools/step_rows.py"""
import sys, os, ctypes, random
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, uegan_amd
from uegan_amd import _lib, losses, models, trainer
uegan_amd.set_compute_dtype(torch.bfloat16)
lib = _lib.load(); dev = torch.device("cuda:0")
torch.manual_seed(1990)
G = models.Generator(32, "none", "LeakyReLU", False).to(dev)
D = models.Discriminator(32, "none", "LeakyReLU", True, "rahinge").to(dev)
P = losses.PerceptualLoss(vgg_weights="seeded").to(dev)
T = trainer.Trainer(G, D, P, pool_size=50, rng=random.Random(1990))
g = torch.Generator().manual_seed(1990)
raw = (torch.rand(16, 3, 512, 512, generator=g) * 2 - 1).to(dev); exp = (torch.rand(16, 3, 512, 512, generator=g) * 2 - 1).to(dev)
for _ in range(2): T.train_step(raw, exp)
torch.cuda.synchronize()
_lib.check(lib.uegan_profile_begin(2000))
T.train_step(raw, exp)
torch.cuda.synchronize()
ents = (_lib.ProfileEntry * 128)(); n = ctypes.c_int(0)
_lib.check(lib.uegan_profile_end(ents, 128, ctypes.byref(n)))
rows = sorted([(ents[i].total_ms, ents[i].name.decode(), ents[i].launches, ents[i].total_flops, ents[i].total_bytes) for i in range(n.value)], reverse=True)
for ms, name, nl, fl, by in rows:
    print("%-62s %3d launches %7.3f ms  %7.1f TF/s %7.0f GB/s  %8.2f GF/launch" % (name, nl, ms, fl / ms / 1e9, by / ms / 1e6, fl / nl / 1e9))
