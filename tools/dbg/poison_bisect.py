"""Bisect which torch.empty* allocation of one training step is read before it is written: baseline = every allocation zero-filled (what a
fresh process sees), trial = allocations [lo, hi) NaN-filled."""
import os, sys, random, traceback
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
from uegan_amd import ops, fused, losses, models, trainer, variants
from oracle import uegan_oracle as O
dev = torch.device("cuda:0")
B, S = int(sys.argv[1]), int(sys.argv[2])
dt = torch.float32 if len(sys.argv) < 4 or sys.argv[3] == "f32" else torch.bfloat16
_empty, _empty_like = torch.empty, torch.empty_like
state = {"n": 0, "lo": -1, "hi": -1, "sites": {}}
def _fill(t):
    if t.is_cuda:
        i = state["n"]; state["n"] += 1
        if state["lo"] <= i < state["hi"]:
            t.fill_(float("nan")) if t.dtype.is_floating_point else t.fill_(0x7f)
            if state["hi"] - state["lo"] <= 2:
                state["sites"][i] = "".join(traceback.format_stack(limit=7)[:-2]) + " shape=%s dtype=%s" % (tuple(t.shape), t.dtype)
        else:
            t.zero_()
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
ops.set_compute_dtype(dt)
P = losses.PerceptualLoss(vgg_weights="seeded").to(dev)
def run(lo, hi):
    G = models.Generator(32, "none", "LeakyReLU", False); D = models.Discriminator(32, "none", "LeakyReLU", True, "rahinge")
    G.load_state_dict(PG); D.load_state_dict(PD)
    T = trainer.Trainer(G.to(dev), D.to(dev), P, pool_size=50, rng=random.Random(1990))
    state.update(n=0, lo=lo, hi=hi)
    T.train_step(raw, exp)
    torch.cuda.synchronize()
    n = state["n"]
    out = (T.loss_items(), T.g_optimizer.flat_grad.clone(), T.d_optimizer.flat_grad.clone(), n)
    state.update(lo=-1, hi=-1)
    return out
base = run(-1, -1)
N = base[3]
print("allocations per step:", N, base[0])
def bad(lo, hi):
    r = run(lo, hi)
    return not (torch.equal(r[1], base[1]) and torch.equal(r[2], base[2]) and r[0] == base[0])
print("repeat clean differs:", bad(-1, -1))
found = []
def search(lo, hi):
    if not bad(lo, hi): return
    if hi - lo == 1:
        found.append(lo); return
    mid = (lo + hi) // 2
    search(lo, mid); search(mid, hi)
search(0, N)
print("culprits:", found)
for i in found[:8]:
    run(i, i + 1)
    print("=== allocation", i); print(state["sites"].get(i))
