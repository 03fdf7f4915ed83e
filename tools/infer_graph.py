#!/usr/bin/env python3
"""hipGraph replays of the batch-1 eval-mode generator (tester.GraphedGenerator), for rocprofv3 --kernel-trace (tools/infer_trace.py).
usage: infer_graph.py [bf16|f16|f16p] [replays] [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import uegan_amd
from uegan_amd import models, tester
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda:0")
uegan_amd.set_compute_dtype(torch.bfloat16 if mode == "bf16" else torch.float16)
uegan_amd.set_precise(mode == "f16p")
torch.manual_seed(1990)
G = models.Generator(32, "none", "LeakyReLU", False).to(dev)
x = (torch.rand(int(sys.argv[3]) if len(sys.argv) > 3 else 1, 3, 512, 512) * 2 - 1).to(dev)
GG = tester.GraphedGenerator(G, x.shape)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 20):
    GG(x)
torch.cuda.synchronize()
