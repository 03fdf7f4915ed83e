"""The whole-network functions of uegan_amd/fused.py against the per-layer module API they restructure (same kernels,
different sequencing: batched passes, sub-batch backward, fused loss) -- both are additionally pinned to the reference-generated
fixtures by tests/test_train_step.py / tests/test_parity_full.py, which run the Trainer with fused_passes=True."""
import random

import pytest
import torch

from helpers import BACKENDS, golden, tens, use_backend
from oracle import uegan_oracle as O
from uegan_amd import fused, losses, models, ops, trainer


def _vgg8():
    zl = golden("losses.npz")
    return {k[len("vgg8/"):]: tens(zl, k) for k in zl.files if k.startswith("vgg8/")}


def _relmax(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("backend", BACKENDS)
def test_fused_fidelity_loss_equals_two_pass_autograd(backend):
    """losses.py:22-36: one VGG pass over [x; y] with a backward over the x half == two passes with per-layer autograd"""
    dev = use_backend(backend)
    ops.set_compute_dtype(torch.float32)
    P = losses.PerceptualLoss(vgg_weights=_vgg8(), width_div=8).to(dev)
    g = torch.Generator().manual_seed(0)
    x0, y0 = torch.rand(2, 3, 32, 48, generator=g).to(dev), torch.rand(2, 3, 32, 48, generator=g).to(dev)
    res = []
    for fz in (False, True):
        P.fused = fz
        x = x0.clone().requires_grad_(True)
        l = P(x, y0) * 3.0
        l.backward()
        res.append((float(l.detach()), x.grad.clone()))
    assert abs(res[0][0] - res[1][0]) <= 1e-6 * abs(res[0][0])
    assert _relmax(res[1][1], res[0][1]) < 1e-5
    # and against the oracle (reference arithmetic)
    xo = x0.cpu().clone().requires_grad_(True)
    lo = O.perceptual_loss(_vgg8(), xo, y0.cpu()) * 3.0
    lo.backward()
    assert abs(res[1][0] - float(lo.detach())) <= 1e-4 * abs(float(lo.detach())) and _relmax(res[1][1].cpu(), xo.grad) < 1e-3
    with torch.no_grad():
        assert abs(float(P(x0, y0)) * 3.0 - res[1][0]) <= 1e-6 * abs(res[1][0])          # no-grad path (nothing saved)


@pytest.mark.parametrize("backend", BACKENDS)
def test_generator_forward_pair_equals_two_passes(backend):
    """trainer.py:85 + :112 as one generator pass over the concatenated batch"""
    dev = use_backend(backend)
    ops.set_compute_dtype(torch.float32)
    z = golden("g_cd8_default.npz")
    G = models.Generator(8, "none", "LeakyReLU", False)
    G.load_state_dict({k[len("param/"):]: tens(z, k) for k in z.files if k.startswith("param/")})
    G = G.to(dev)
    g = torch.Generator().manual_seed(1)
    xa, xb = (torch.rand(1, 3, 32, 48, generator=g) * 2 - 1).to(dev), (torch.rand(2, 3, 32, 48, generator=g) * 2 - 1).to(dev)
    wa, wb = torch.randn(1, 3, 32, 48, generator=g).to(dev), torch.randn(2, 3, 32, 48, generator=g).to(dev)

    def run(pair, use_b=True):
        G.zero_grad()
        oa, ob = G.forward_pair(xa, xb) if pair else (G(xa), G(xb))
        l = (oa * wa).sum() + ((ob * wb).sum() if use_b else 0.0)
        l.backward()
        return oa.detach(), ob.detach(), {k: p.grad.clone() for k, p in G.named_parameters()}

    a, b = run(False), run(True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])                               # per-sample ops: bit-identical images
    for k in a[2]:
        assert _relmax(b[2][k], a[2][k]) < 1e-5, k
    a, b = run(False, use_b=False), run(True, use_b=False)                                   # one output unused: its half gets a zero gradient
    for k in a[2]:
        assert _relmax(b[2][k], a[2][k]) < 1e-5, k
    with pytest.raises(RuntimeError):
        G.forward_pair(xa, xb[:, :, :16])


@pytest.mark.parametrize("backend", BACKENDS)
def test_fused_discriminator_loss_equals_module_passes(backend):
    """trainer.py:90-95 (three D passes, two GANLoss terms) and :102-104 (two passes, D frozen or not) as single batched passes with a
    per-group spectral-norm sigma: loss, every parameter gradient, the image gradient and the advanced u / v must agree"""
    dev = use_backend(backend)
    ops.set_compute_dtype(torch.float32)
    PD = O.init_params(O.discriminator_param_shapes(8), 42, "default")
    g = torch.Generator().manual_seed(0)
    shape = (1, 3, 80, 80) if backend == "emu" else (2, 3, 80, 96)       # (the emulator runs every GPU thread as a fiber: keep it small)
    xs = [(torch.rand(*shape, generator=g) * 2 - 1).to(dev) for _ in range(3)]
    A = losses.GANLoss("rahinge")

    def fresh():
        D = models.Discriminator(8, "none", "LeakyReLU", True, "rahinge")
        D.load_state_dict(PD)
        return D.to(dev).train()

    D1, D2 = fresh(), fresh()
    pe, pf, pr = D1(xs[0]), D1(xs[1]), D1(xs[2])
    l1 = A(pe, pf, None, None, for_discriminator=True) + A(pe, pr, None, None, for_discriminator=True)
    l1.backward()
    l2 = fused.discriminator_loss(D2, xs, [(0, 1), (0, 2)], True)
    l2.backward()
    assert l2.shape == (1,) and abs(float(l1.detach()) - float(l2.detach())) <= 1e-6 * abs(float(l1.detach()))
    for (k, p), q in zip(D1.named_parameters(), D2.parameters()):
        assert _relmax(q.grad, p.grad) < 2e-5, k
    for (k, a), b in zip(D1.state_dict().items(), D2.state_dict().values()):
        if k.endswith(("weight_u", "weight_v")):
            assert float((a - b).abs().max()) < 1e-6, k                                      # three power iterations each
    # oracle (reference arithmetic) for the same step
    Dp = O._with_grad({k: v.clone() for k, v in PD.items()})
    xc = [x.cpu() for x in xs]
    oe, of, orr = O.discriminator_forward(Dp, xc[0], True), O.discriminator_forward(Dp, xc[1], True), O.discriminator_forward(Dp, xc[2], True)
    lo = O.rahinge_loss(oe, of, True) + O.rahinge_loss(oe, orr, True)
    tr = O.trainable(Dp)
    go = dict(zip(tr.keys(), torch.autograd.grad(lo.sum(), list(tr.values()))))
    assert abs(float(lo.detach()) - float(l2.detach())) <= 1e-4 * abs(float(lo.detach()))
    for k, p in D2.named_parameters():
        assert _relmax(p.grad.cpu(), go[k]) < 1e-3, k

    for frozen in (True, False):
        D1, D2 = fresh(), fresh()
        if frozen:
            for p in list(D1.parameters()) + list(D2.parameters()):
                p.requires_grad_(False)
        xf1, xf2 = xs[1].clone().requires_grad_(True), xs[1].clone().requires_grad_(True)
        l1 = A(D1(xs[0]), D1(xf1), None, None, for_discriminator=False)
        l1.backward()
        l2 = fused.discriminator_loss(D2, [xs[0], xf2], [(0, 1)], False)
        l2.backward()
        assert abs(float(l1.detach()) - float(l2.detach())) <= 1e-6 * abs(float(l1.detach()))
        assert _relmax(xf2.grad, xf1.grad) < 2e-5
        if not frozen:
            for (k, p), q in zip(D1.named_parameters(), D2.parameters()):
                assert _relmax(q.grad, p.grad) < 2e-5, k
    D2.eval()                                                                                 # eval mode: u / v do not advance
    before = {k: v.clone() for k, v in D2.state_dict().items()}
    with torch.no_grad():
        fused.discriminator_loss(D2, xs[:2], [(0, 1)], False)
    assert all(torch.equal(before[k], v) for k, v in D2.state_dict().items())


@pytest.mark.parametrize("backend", [pytest.param("gpu", id="gpu", marks=pytest.mark.gpu)])
def test_trainer_fused_passes_equal_per_line_module_calls(backend):
    """the full step (trainer.py:77-119) with the batched passes vs one module call per reference line: same losses, images, weights"""
    dev = use_backend(backend)
    ops.set_compute_dtype(torch.float32)
    z = golden("train_cd8_default.npz")
    PG = {k[len("G_init/"):]: tens(z, k) for k in z.files if k.startswith("G_init/")}
    PD = {k[len("D_init/"):]: tens(z, k) for k in z.files if k.startswith("D_init/")}
    out = []
    for fz in (False, True):
        G = models.Generator(8, "none", "LeakyReLU", False)
        D = models.Discriminator(8, "none", "LeakyReLU", True, "rahinge")
        G.load_state_dict(PG)
        D.load_state_dict(PD)
        T = trainer.Trainer(G.to(dev), D.to(dev), losses.PerceptualLoss(vgg_weights=_vgg8(), width_div=8).to(dev), pool_size=3,
                            rng=random.Random(1990), fused_passes=fz)
        logs = []
        for step in range(3):
            T.train_step(tens(z, "raw%d" % step, dev), tens(z, "exp%d" % step, dev))
            logs.append(T.loss_items())
        out.append((logs, T.fake_exp.cpu(), {k: v.cpu() for k, v in list(G.state_dict().items()) + list(D.state_dict().items())}))
    for a, b in zip(out[0][0], out[1][0]):
        for k in a:
            assert abs(a[k] - b[k]) <= 1e-4 * abs(a[k]) + 1e-7, (k, a[k], b[k])
    assert float((out[0][1] - out[1][1]).abs().max()) < 1e-4
    for k, v in out[0][2].items():
        assert float((v - out[1][2][k]).abs().max()) <= 2e-4 + 1e-3 * float(v.abs().max()), k
