"""Data-parallel path on CPU: world_size 2 over gloo.  Each rank runs the real bucket / fused-Adam code
(kernel sources on the CPU emulator): flat fp32 gradient bucket aliased by the parameters' .grad, ONE all-reduce per
optimizer step, 1/world folded into the Adam kernel.  Expected result: identical weights on every rank, equal to a
single-process torch.optim.Adam step on the rank-averaged gradients (= the average of N independent reference steps'
gradients)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import EMU_LIB, build_emu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["UEGAN_EMU_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uegan_amd import _lib, ops, trainer
    _lib._inject_for_tests(EMU_LIB)
    torch.manual_seed(100 + rank)                      # deliberately different init per rank
    params = [torch.nn.Parameter(torch.randn(7, 5)), torch.nn.Parameter(torch.randn(300)), torch.nn.Parameter(torch.randn(3, 2, 2, 2))]
    for p in params:                                   # what Trainer(broadcast_init=True) does
        dist.broadcast(p.data, src=0)
    opt = ops.FusedAdamL2(params, 1e-2, (0.5, 0.999), 1e-8, 1e-4)
    bucket = trainer.GradBucket(opt.flat_grad)
    assert bucket.world == world
    grads_log = []
    for step in range(3):
        opt.zero_grad()
        g = torch.Generator().manual_seed(1000 * step + rank)
        gs = [torch.randn(p.shape, generator=g) for p in params]
        for p, gr in zip(params, gs):
            p.grad.add_(gr)                            # autograd accumulates in place into the bucket views
        grads_log.append(gs)
        bucket.start()
        opt.step(bucket.finish())
    out[rank] = ([p.detach().clone() for p in params], grads_log)
    dist.destroy_process_group()


def test_two_rank_bucket_allreduce_matches_averaged_adam():
    build_emu()
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        res = {r: out[r] for r in range(world)}
    p0, p1 = res[0][0], res[1][0]
    for a, b in zip(p0, p1):
        assert torch.equal(a, b)                       # replicas stay bit-identical
    # single-process reference: same init (rank 0's), Adam on the average of the two ranks' gradients
    torch.manual_seed(100)
    ref = [torch.nn.Parameter(torch.randn(7, 5)), torch.nn.Parameter(torch.randn(300)), torch.nn.Parameter(torch.randn(3, 2, 2, 2))]
    topt = torch.optim.Adam(ref, lr=1e-2, betas=(0.5, 0.999), weight_decay=1e-4)
    for step in range(3):
        topt.zero_grad()
        for i, q in enumerate(ref):
            q.grad = 0.5 * (res[0][1][step][i] + res[1][1][step][i])
        topt.step()
    for a, q in zip(p0, ref):
        assert float((a - q).abs().max()) < 1e-6


def test_single_process_bucket_is_identity():
    from uegan_amd import trainer
    b = trainer.GradBucket(torch.ones(4))
    b.start()
    assert b.finish() == 1.0 and b.world == 1


def test_lr_lambda_rule_matches_reference():
    from uegan_amd import trainer
    # trainer.py:347-349: 1 - max(0, epoch + 1 - 50) / 50
    assert trainer.lambda_rule(0) == 1.0 and trainer.lambda_rule(49) == 1.0
    assert abs(trainer.lambda_rule(50) - 0.98) < 1e-12 and abs(trainer.lambda_rule(99) - 0.0) < 1e-12
