#!/usr/bin/env python3
"""bench.py -- UEGAN training throughput on MI355X (BASELINE.json: "train imgs/sec @512px bs=16 on 1/2/4/8 MI355X").

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one full training iteration of the reference (trainer.py:77-119): G fwd x2, D fwd x5, VGG19 fwd x2,
rahinge + VGG-fidelity + multiscale-L1 losses, both backward sweeps, two Adam updates -- on a synthetic FiveK-shaped
batch (16 x 3 x 512 x 512 per GPU, uniform(-1,1), seed 1990+rank) already resident in HBM, random-init G/D of the
reference architecture (conv_dim 32) and the seeded stand-in VGG19 (no network: pretrained weights unavailable).
Prints ONE JSON line (rank 0).  Extra objects:
  roofline     -- the dominant kernel (the MFMA implicit-GEMM convolution instantiation with the most time), measured
                  live with HIP events on the launch stream during the timed steps: algorithmic FLOP/s vs dense MFMA peak.
  cpu_baseline -- the CPU oracle (plain PyTorch-CPU restatement, kind "port") timed on this box's host cores on a
                  bounded sample (batch 2 @512^2, 16 threads, 1 warm-up + 2 timed steps), rank 0 at N=1 only.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# dense peaks from /opt/skills/guides/MI355X_MICROARCH.md
PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU (BASELINE config: 16)")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--dtype", choices=["bf16", "f32"], default="bf16")
    ap.add_argument("--conv-dim", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=2)
    ap.add_argument("--no-profile", action="store_true", help="skip the per-launch HIP-event timing")
    ap.add_argument("--no-infer", dest="infer", action="store_false", help="skip the single-image G inference timing (tester.py:58-67)")
    ap.set_defaults(infer=True)
    return ap.parse_args()


def cpu_baseline(args):
    """Oracle (port) train step on the host cores: bounded sample of the same workload."""
    import random
    from oracle import uegan_oracle as O
    # measured on the MI355X host (EPYC 9575F, 256 hw threads): 8 thr 6.0 s/step, 16 thr 5.3, 32 thr 5.6, 64 thr 9.3,
    # 256 thr 291 s (oneDNN oversubscription) -> use 16 worker threads
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    cores = torch.get_num_threads()
    B, S, cd = args.cpu_batch, args.size, args.conv_dim
    PG = O.init_params(O.generator_param_shapes(cd), 41, "default")
    PD = O.init_params(O.discriminator_param_shapes(cd), 42, "default")
    V = O.make_vgg_weights(seed=1234, width_div=1)
    St = O.TrainState(PG, PD, V, pool_size=50, rng=random.Random(1990))
    g = torch.Generator().manual_seed(1990)
    times = []
    for it in range(3):
        raw = torch.rand(B, 3, S, S, generator=g) * 2 - 1
        exp = torch.rand(B, 3, S, S, generator=g) * 2 - 1
        t = time.time()
        O.train_step(St, raw, exp)
        times.append(time.time() - t)
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    t_med = sorted(times[1:])[0]
    return {"value": round(B / t_med, 4), "unit": "imgs/sec", "cores": cores, "kind": "port",
            "sample": "oracle/uegan_oracle.py train_step (plain PyTorch-CPU fp32), batch %d @%dx%d, 1 warm-up + 2 timed steps, best %.2f s/step"
                      % (B, S, S, t_med),
            "cpu_model": model}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import random
    import uegan_amd
    from uegan_amd import _lib, losses, models, trainer

    uegan_amd.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    lib = _lib.load()
    torch.manual_seed(1990)            # same init on every rank (also broadcast from rank 0 by the Trainer)
    G = models.Generator(args.conv_dim, "none", "LeakyReLU", False).to(dev)
    D = models.Discriminator(args.conv_dim, "none", "LeakyReLU", True, "rahinge").to(dev)
    P = losses.PerceptualLoss(vgg_weights="seeded").to(dev)      # explicit opt-in: no network, pretrained weights unavailable
    T = trainer.Trainer(G, D, P, pool_size=50, rng=random.Random(1990 + rank))

    B, S = args.batch, args.size
    g = torch.Generator().manual_seed(1990 + rank)
    nb = 2                                   # two resident synthetic batches, alternated
    raws = [(torch.rand(B, 3, S, S, generator=g) * 2 - 1).to(dev) for _ in range(nb)]
    exps = [(torch.rand(B, 3, S, S, generator=g) * 2 - 1).to(dev) for _ in range(nb)]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        T.train_step(raws[i % nb], exps[i % nb])
    sync()
    prof = (not args.no_profile)
    if prof:
        _lib.check(lib.uegan_profile_begin(400 * args.steps + 64))
    t0 = time.perf_counter()
    for i in range(args.steps):
        T.train_step(raws[i % nb], exps[i % nb])
    sync()
    dt = time.perf_counter() - t0
    items = T.loss_items()
    roof = None
    if prof:
        ents = (_lib.ProfileEntry * 96)()
        n = ctypes.c_int(0)
        _lib.check(lib.uegan_profile_end(ents, 96, ctypes.byref(n)))
        rows = [dict(name=ents[i].name.decode(), launches=int(ents[i].launches), total_ms=float(ents[i].total_ms),
                     total_flops=float(ents[i].total_flops)) for i in range(n.value)]
        rows.sort(key=lambda r: -r["total_ms"])
        if rows:
            top = rows[0]
            peak = PEAK_TFLOPS[args.dtype]
            ach = top["total_flops"] / (top["total_ms"] * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": top["name"], "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": None, "launches_per_step": top["launches"] / args.steps,
                    "avg_launch_ms": round(top["total_ms"] / top["launches"], 5),
                    "gflop_per_launch": round(top["total_flops"] / top["launches"] / 1e9, 3),
                    "kernel_ms_per_step": round(top["total_ms"] / args.steps, 3),
                    "all_mfma_kernels_ms_per_step": round(sum(r["total_ms"] for r in rows) / args.steps, 3),
                    "all_mfma_kernels_tflops": round(sum(r["total_flops"] for r in rows) / (sum(r["total_ms"] for r in rows) * 1e-3) / 1e12, 2),
                    "top5": [{"kernel": r["name"], "ms_per_step": round(r["total_ms"] / args.steps, 2),
                              "tflops": round(r["total_flops"] / (r["total_ms"] * 1e-3) / 1e12, 1)} for r in rows[:5]]}
    if roof is not None:
        # HBM traffic of that kernel comes from separate rocprofv3 --pmc passes (profiles/pmc_traffic.json, committed with the
        # round's profiles): PMC collection cannot run inside this process
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(roof["kernel"])
            if pmc:
                roof["traffic"] = pmc["traffic_bytes"]
                roof["traffic_unit"] = "bytes/launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, %s)" % pmc.get("round", "")
        except (OSError, ValueError):
            pass
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    infer_ms = None
    if args.infer and rank == 0:
        from uegan_amd import tester
        x1 = raws[0][:1].contiguous()
        for _ in range(3):
            tester.enhance(G, x1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(20):
            tester.enhance(G, x1)
        torch.cuda.synchronize()
        infer_ms = (time.perf_counter() - t1) / 20 * 1e3

    if rank == 0:
        out = {
            "metric": "train imgs/sec @512px bs=16 on 1/2/4/8 MI355X; infer ms/img",
            "value": round(world * B * args.steps / dt, 3), "unit": "imgs/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "FiveK-shaped %dx%d batch=%d/GPU, full G/D/VGG train step (trainer.py:77-119), conv_dim=%d, "
                                   "seeded stand-in VGG19" % (S, S, B, args.conv_dim),
                       "global_batch": world * B, "parallelism": "dp%d" % world, "pool_size": 50},
            "losses_last_step": {k: round(v, 6) for k, v in items.items()},
        }
        if roof is not None:
            out["roofline"] = roof
        if infer_ms is not None:
            out["infer_ms_per_img"] = round(infer_ms, 3)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
