#!/usr/bin/env python3
"""RCCL sanity on the GPU box: the Trainer's distributed path (broadcast + two bucket all-reduces per step) at whatever world
size torch.distributed.run gives (1 on the single-GPU test boxes).  Usage:
  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 --master-port 29511 tools/dist_smoke.py"""
import os
import random
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

rank, world, lr = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(lr)
dev = torch.device("cuda", lr)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
import uegan_amd  # noqa: E402
from uegan_amd import losses, models, trainer  # noqa: E402

uegan_amd.set_compute_dtype(torch.bfloat16)
torch.manual_seed(1990)
G = models.Generator(32, "none", "LeakyReLU", False).to(dev)
D = models.Discriminator(32, "none", "LeakyReLU", True, "rahinge").to(dev)
P = losses.PerceptualLoss(vgg_weights="seeded").to(dev)
T = trainer.Trainer(G, D, P, pool_size=50, rng=random.Random(1990 + rank))
g = torch.Generator().manual_seed(7 + rank)
for _ in range(2):
    raw = (torch.rand(2, 3, 128, 128, generator=g) * 2 - 1).to(dev)
    exp = (torch.rand(2, 3, 128, 128, generator=g) * 2 - 1).to(dev)
    T.train_step(raw, exp)
items = T.loss_items()
# every rank must hold identical weights after the all-reduced updates
w = torch.cat([p.detach().flatten()[:64] for p in G.parameters()][:8]).float()
ref = w.clone()
dist.broadcast(ref, 0)
assert torch.equal(w, ref), "weights diverged across ranks"
if rank == 0:
    print("dist smoke ok: world", world, {k: round(v, 5) for k, v in items.items()})
dist.destroy_process_group()
