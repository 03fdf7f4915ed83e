"""Uninitialised-memory hunt: run one training step normally and with every torch.empty / empty_like / new_empty of the uegan_amd modules
filled with NaN (float) / 0x7f (ints); any kernel that reads a buffer before writing it shows up as a NaN or a changed result."""
import os, sys, random
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import uegan_amd
from uegan_amd import ops, fused, losses, models, trainer, variants
from oracle import uegan_oracle as O
dev = torch.device("cuda:0")
B, S = int(sys.argv[1]), int(sys.argv[2])
_empty, _empty_like = torch.empty, torch.empty_like
POISON = [False]
def _fill(t):
    if POISON[0] and t.is_cuda:
        if t.dtype.is_floating_point: t.fill_(float("nan"))
        else: t.fill_(0x7f)
    return t
class TorchProxy:
    def __getattr__(self, n): return getattr(torch, n)
    def empty(self, *a, **k): return _fill(_empty(*a, **k))
    def empty_like(self, *a, **k): return _fill(_empty_like(*a, **k))
proxy = TorchProxy()
for m in (ops, fused, losses, models, trainer, variants):
    m.torch = proxy
def images(B, S, seed):
    g = torch.Generator().manual_seed(seed)
    lo = torch.rand(B, 3, S // 32, S // 32, generator=g)
    x = torch.nn.functional.interpolate(lo, size=(S, S), mode="bicubic", align_corners=False) + 0.03 * torch.randn(B, 3, S, S, generator=g)
    return (x.clamp(0, 1) * 2 - 1).contiguous()
PG = O.init_params(O.generator_param_shapes(32), 41, "default")
PD = O.init_params(O.discriminator_param_shapes(32), 42, "default")
raw, exp = images(B, S, 1990).to(dev), images(B, S, 1991).to(dev)
for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
    res = []
    for poison in (False, True, False):
        ops.set_compute_dtype(dt)
        POISON[0] = False
        G = models.Generator(32, "none", "LeakyReLU", False); D = models.Discriminator(32, "none", "LeakyReLU", True, "rahinge")
        G.load_state_dict(PG); D.load_state_dict(PD)
        T = trainer.Trainer(G.to(dev), D.to(dev), losses.PerceptualLoss(vgg_weights="seeded").to(dev), pool_size=50, rng=random.Random(1990))
        POISON[0] = poison
        for step in range(2):
            T.train_step(raw, exp)
        torch.cuda.synchronize()
        POISON[0] = False
        res.append((T.loss_items(), T.g_optimizer.flat_grad.clone(), T.d_optimizer.flat_grad.clone(), T.fake_exp.clone()))
        del T, G, D
        torch.cuda.empty_cache()
    for i, tag in ((1, "poisoned"), (2, "repeat")):
        a, b = res[0], res[i]
        print(name, tag, "losses", {k: (a[0][k], b[0][k]) for k in a[0] if a[0][k] != b[0][k]} or "identical",
              "gG equal", bool(torch.equal(a[1], b[1])), "gD equal", bool(torch.equal(a[2], b[2])), "fake equal", bool(torch.equal(a[3], b[3])),
              "nan gG %d gD %d" % (int(torch.isnan(b[1]).sum()), int(torch.isnan(b[2]).sum())))
