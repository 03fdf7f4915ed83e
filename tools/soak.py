#!/usr/bin/env python3
"""End-to-end soak on one MI355X: a PNG tree on disk -> uegan_amd.data.DeviceLoader (decode threads, pinned ring, side-stream H2D +
device transform) -> Trainer.train_step (bf16), N steps; prints the five losses, checks they stay finite, and reports images/s with
the loader against the same steps on resident tensors.  Usage: python tools/soak.py [--steps 40] [--batch 16] [--img 512] [--resize 256]"""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import uegan_amd  # noqa: E402
from uegan_amd import data, losses, models, trainer  # noqa: E402

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warm", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--img", type=int, default=512)
    ap.add_argument("--resize", type=int, default=256)
    ap.add_argument("--files", type=int, default=48)
    ap.add_argument("--workers", type=int, default=16)
    ap.add_argument("--mode", default="thread", help="decode workers: thread | process")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--precise", action="store_true", help="uegan_amd.set_precise(True) (with --dtype f16)")
    ap.add_argument("--loss-scale", default=None, help="Trainer(loss_scale=...): a number or 'dynamic' (default: the Trainer's choice for the dtype)")
    args = ap.parse_args()

    from PIL import Image
    tmp = tempfile.mkdtemp(prefix="uegan_soak_")
    rng = np.random.default_rng(0)
    for d in ("exp", "raw"):
        os.makedirs(os.path.join(tmp, d))
        for i in range(args.files):
            h, w = args.img + int(rng.integers(0, 120)), args.img + int(rng.integers(0, 160))
            low = rng.integers(0, 256, size=(h // 32 + 2, w // 32 + 2, 3)).astype(np.uint8)
            im = Image.fromarray(low, "RGB").resize((w, h), Image.BICUBIC)
            im.save(os.path.join(tmp, d, "im%03d.png" % i), compress_level=1)

    dev = torch.device("cuda:0")
    uegan_amd.set_compute_dtype({"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype])
    uegan_amd.set_precise(args.precise)
    torch.manual_seed(1990)
    G = models.Generator(32, "none", "LeakyReLU", False).to(dev)
    D = models.Discriminator(32, "none", "LeakyReLU", True, "rahinge").to(dev)
    P = losses.PerceptualLoss(vgg_weights="seeded").to(dev)
    ls = args.loss_scale if args.loss_scale in (None, "dynamic") else float(args.loss_scale)
    T = trainer.Trainer(G, D, P, pool_size=50, loss_scale=ls)
    loader = data.get_train_loader(tmp, img_size=args.img, resize_size=args.resize, batch_size=args.batch, num_workers=args.workers,
                                   generator=torch.Generator().manual_seed(7), workers=args.mode)
    fetch = data.InputFetcher(loader)
    for _ in range(args.warm):           # (also lets a spawned worker pool finish importing)
        b = next(fetch)
        T.train_step(b.img_raw, b.img_exp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        b = next(fetch)
        T.train_step(b.img_raw, b.img_exp)
        if (i + 1) % 10 == 0:
            it = T.loss_items()
            assert all(np.isfinite(v) for v in it.values()), it
            print("step %3d  " % (i + 1) + "  ".join("%s %.4f" % kv for kv in it.items()), flush=True)
    torch.cuda.synchronize()
    dt_loader = time.perf_counter() - t0
    raw, exp = b.img_raw.clone(), b.img_exp.clone()
    for _ in range(2):
        T.train_step(raw, exp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        T.train_step(raw, exp)
    torch.cuda.synchronize()
    dt_res = time.perf_counter() - t0
    assert float(b.img_raw.min()) >= -1.0 and float(b.img_raw.max()) <= 1.0
    print("mode: %s%s, loss scale %s, skipped steps %d" % (args.dtype, " precise" if args.precise else "", T.loss_scale, T.skipped_steps))
    print("with loader: %.1f img/s (%.2f ms/step); resident inputs: %.1f img/s (%.2f ms/step); %dx%d crops -> %dx%d, batch %d, %d decode %s workers"
          % (args.steps * args.batch / dt_loader, dt_loader / args.steps * 1e3, args.steps * args.batch / dt_res, dt_res / args.steps * 1e3,
             args.img, args.img, args.resize, args.resize, args.batch, args.workers, args.mode))
    loader.close()


if __name__ == "__main__":      # (process workers are spawned: they re-import this file)
    main()
