#!/usr/bin/env python3
"""single-image generator inference (tester.py:58-67) in a loop, for rocprofv3 --kernel-trace (tools/gpu_kstat.sh)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import uegan_amd
from uegan_amd import models, tester
dev = torch.device("cuda:0")
uegan_amd.set_compute_dtype(torch.bfloat16)
torch.manual_seed(1990)
G = models.Generator(32, "none", "LeakyReLU", False).to(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
x = (torch.rand(B, 3, 512, 512) * 2 - 1).to(dev)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
    tester.enhance(G, x)
torch.cuda.synchronize()
