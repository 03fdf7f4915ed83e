"""Non-default flags: norm_fun / act_fun / use_sn variants of Generator and Discriminator against fixtures
generated from the REFERENCE's models.py (tools/make_golden_variants.py), every GANLoss mode against fixtures from the reference's
losses.py (and the oracle restatement against the same fixtures), RMSprop against torch.optim.RMSprop, and a training step
through Trainer with non-default flags."""
import numpy as np
import pytest
import torch

from helpers import BACKENDS, golden, rel, tens, use_backend
from oracle import uegan_oracle as O
from uegan_amd import losses, models, ops, variants
from uegan_amd.trainer import Trainer

TOL = 2e-4
G_CONFIGS = {"g_bn_swish_sn": ("BatchNorm", "Swish", True), "g_in_selu": ("InstanceNorm", "SELU", False),
             "g_none_relu_sn": ("none", "ReLU", True), "g_none_none": ("none", "none", False)}
D_CONFIGS = {"d_in_selu_rals": ("InstanceNorm", "SELU", False, "rals"), "d_bn_relu_sn_ls": ("BatchNorm", "ReLU", True, "ls"),
             "d_none_swish_sn": ("none", "Swish", True, "rahinge")}
DEAD = ("conv.0.weight", "conv.2.weight", "fuse.0.bias")


def _check_network(net, z, dev, n_out):
    sd0 = {k[4:]: tens(z, k) for k in z.files if k.startswith("sd0.")}
    assert set(net.state_dict().keys()) == set(sd0.keys())            # the reference's state-dict keys, norm layers and u / v included
    net.load_state_dict(sd0)
    net = net.to(dev)
    ops.invalidate_weight_caches()
    net.train()
    x = tens(z, "x", dev).requires_grad_(True)
    out = net(x)
    outs = out if isinstance(out, list) else [out]
    assert len(outs) == n_out
    for i, o in enumerate(outs):
        assert rel(o, tens(z, "out%d" % i)) < TOL, "out%d" % i
    sum((o * tens(z, "w%d" % i, dev)).sum() for i, o in enumerate(outs)).backward()
    assert rel(x.grad, tens(z, "gx")) < TOL
    for k, p in net.named_parameters():
        ref = tens(z, "grad." + k)
        if k.endswith(DEAD) and "fuse.0" not in k:
            assert p.grad is not None and float(p.grad.abs().max()) == 0.0, k        # forward-dead GAM branch
            continue
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        scale = float(ref.abs().max())
        err = float((g.cpu() - ref).abs().max())
        assert err <= TOL * scale + 2e-4, (k, err, scale)          # (absolute floor: biases in front of a norm layer have true gradient 0)
    for k, b in net.named_buffers():                                     # running statistics, batch counters, power-iteration vectors
        ref = tens(z, "buf1." + k)
        if ref.dtype in (torch.int64, torch.int32):
            assert int(b) == int(ref), k
        else:
            assert rel(b, ref) < TOL, k
    net.eval()
    with torch.no_grad():
        oe = net(x.detach())
    for i, o in enumerate(oe if isinstance(oe, list) else [oe]):
        assert rel(o, tens(z, "eval%d" % i)) < TOL, "eval%d" % i


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name", sorted(G_CONFIGS))
def test_generator_variants_match_reference(backend, name):
    dev = use_backend(backend)
    ops.set_compute_dtype(torch.float32)
    norm, act, sn = G_CONFIGS[name]
    _check_network(models.Generator(8, norm, act, sn), golden("variants_%s.npz" % name), dev, 1)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name", sorted(D_CONFIGS))
def test_discriminator_variants_match_reference(backend, name):
    dev = use_backend(backend)
    ops.set_compute_dtype(torch.float32)
    norm, act, sn, adv = D_CONFIGS[name]
    _check_network(models.Discriminator(8, norm, act, sn, adv), golden("variants_%s.npz" % name), dev, 5)


def _gan_cases(z):
    for ci in range(int(z["ncases"])):
        tag = "c%02d" % ci
        mode, target, for_real, for_fake, for_d = [str(v) for v in z[tag + ".meta"]]
        conv = {"True": True, "False": False, "None": None}
        yield tag, mode, conv[target], conv[for_real], conv[for_fake], conv[for_d]


def test_oracle_gan_loss_matches_reference_fixture():
    z = golden("variants_ganloss.npz")
    real = [tens(z, "real%d" % i) for i in range(5)]
    fake = [tens(z, "fake%d" % i) for i in range(5)]
    for tag, mode, target, for_real, for_fake, for_d in _gan_cases(z):
        rr = [t.clone().requires_grad_(True) for t in real]
        ff = [t.clone().requires_grad_(True) for t in fake]
        loss = O.gan_loss(mode, rr, ff, target, for_real, for_fake, for_d)
        assert loss.shape == (1,) and abs(float(loss) - float(z[tag + ".loss"][0])) < 1e-5 * max(1.0, abs(float(loss))), (tag, mode)
        loss.sum().backward()
        for i in range(5):
            for g, key in ((rr[i].grad, "greal"), (ff[i].grad, "gfake")):
                ref = tens(z, "%s.%s%d" % (tag, key, i))
                g = g if g is not None else torch.zeros_like(ref)
                assert float((g - ref).abs().max()) < 1e-6, (tag, mode, key, i)


@pytest.mark.parametrize("backend", BACKENDS)
def test_gan_loss_modes_match_reference(backend):
    dev = use_backend(backend)
    z = golden("variants_ganloss.npz")
    real = [tens(z, "real%d" % i, dev) for i in range(5)]
    fake = [tens(z, "fake%d" % i, dev) for i in range(5)]
    n = 0
    for tag, mode, target, for_real, for_fake, for_d in _gan_cases(z):
        crit = losses.GANLoss(mode)
        rr = [t.clone().requires_grad_(True) for t in real]
        ff = [t.clone().requires_grad_(True) for t in fake]
        loss = crit(rr, ff, target, for_real, for_fake, for_discriminator=for_d)
        ref = float(z[tag + ".loss"][0])
        assert loss.shape == (1,) and abs(float(loss) - ref) < 1e-5 * max(1.0, abs(ref)), (tag, mode, float(loss), ref)
        (loss.sum() * 1.5).backward()                                    # (a non-unit upstream gradient)
        for i in range(5):
            for t, key in ((rr[i], "greal"), (ff[i], "gfake")):
                want = 1.5 * tens(z, "%s.%s%d" % (tag, key, i))
                g = t.grad.cpu() if t.grad is not None else torch.zeros_like(want)
                assert float((g - want).abs().max()) < 2e-6, (tag, mode, key, i)
        n += 1
    assert n == int(z["ncases"]) and n >= 20
    # the one-list modes need for_real or for_fake: the reference raises, and its trainer (which passes neither) cannot use them
    for mode in ("original", "ls", "hinge", "w"):
        with pytest.raises(NotImplementedError):
            losses.GANLoss(mode)(real, fake, None, None, None, for_discriminator=True)
    with pytest.raises(ValueError):
        losses.GANLoss("lsgan")


def _msrec_cases(z):
    for ci in range(int(z["ncases"])):
        sc, kind, ms = [str(v) for v in z["c%02d.meta" % ci]]
        yield "c%02d" % ci, int(sc), kind, ms == "True"


def test_oracle_multiscale_rec_loss_matches_reference_fixture():
    z = golden("variants_msrec.npz")
    for tag, sc, kind, ms in _msrec_cases(z):
        a = tens(z, "a").requires_grad_(True)
        loss = O.multiscale_rec(a, tens(z, "b"), sc, kind, ms)
        loss.backward()
        assert abs(float(loss) - float(z[tag + ".loss"][0])) < 1e-6, (tag, sc, kind, ms)
        assert float((a.grad - tens(z, tag + ".ga")).abs().max()) < 1e-7, (tag, sc, kind, ms)
    for ri in range(int(z["nragged"])):           # sizes AvgPool2d floors
        sc, kind, size = [str(v) for v in z["r%02d.meta" % ri]]
        a = tens(z, "r%02d.a" % ri).requires_grad_(True)
        loss = O.multiscale_rec(a, tens(z, "r%02d.b" % ri), int(sc), kind, True)
        loss.backward()
        assert abs(float(loss) - float(z["r%02d.loss" % ri][0])) < 1e-6 and float((a.grad - tens(z, "r%02d.ga" % ri)).abs().max()) < 1e-7, (sc, kind, size)


@pytest.mark.parametrize("backend", BACKENDS)
def test_multiscale_rec_loss_variants_match_reference(backend):
    """MultiscaleRecLoss(scale, rec_loss_type, multiscale) (losses.py:202-231): l1 / smoothl1 / l2, weight lists of 1, 2, 3 (and 5 -> 3)
    entries, the plain single-scale form -- value and gradient against the reference's own module (tools/make_golden_variants.py)."""
    dev = use_backend(backend)
    z = golden("variants_msrec.npz")
    for tag, sc, kind, ms in _msrec_cases(z):
        crit = losses.MultiscaleRecLoss(scale=sc, rec_loss_type=kind, multiscale=ms)
        a = tens(z, "a", dev).requires_grad_(True)
        loss = crit(a, tens(z, "b", dev))
        (loss * 1.5).backward()
        assert abs(float(loss) - float(z[tag + ".loss"][0])) < 2e-6 * max(1.0, abs(float(loss))), (tag, sc, kind, ms)
        assert float((a.grad.cpu() / 1.5 - tens(z, tag + ".ga")).abs().max()) < 1e-7, (tag, sc, kind, ms)
    a = tens(z, "odd_a", dev).requires_grad_(True)       # odd sizes: legal without pooling
    loss = losses.MultiscaleRecLoss(rec_loss_type="smoothl1", multiscale=False)(a, tens(z, "odd_b", dev))
    loss.backward()
    assert abs(float(loss) - float(z["odd_loss"][0])) < 2e-6 and float((a.grad.cpu() - tens(z, "odd_ga")).abs().max()) < 1e-7
    with pytest.raises(NotImplementedError):
        losses.MultiscaleRecLoss(rec_loss_type="huber")
    # sizes that are not multiples of 4: AvgPool2d(2, 2) floors (losses.py:225-227) -- value and gradient against the reference's own module
    for ri in range(int(z["nragged"])):
        sc, kind, size = [str(v) for v in z["r%02d.meta" % ri]]
        a = tens(z, "r%02d.a" % ri, dev).requires_grad_(True)
        loss = losses.MultiscaleRecLoss(scale=int(sc), rec_loss_type=kind, multiscale=True)(a, tens(z, "r%02d.b" % ri, dev))
        (loss * 0.5).backward()
        assert abs(float(loss) - float(z["r%02d.loss" % ri][0])) < 2e-6 * max(1.0, abs(float(loss))), (sc, kind, size)
        assert float((a.grad.cpu() * 2 - tens(z, "r%02d.ga" % ri)).abs().max()) < 1e-7, (sc, kind, size)
    with pytest.raises(RuntimeError):                     # a map that would pool to nothing (torch: "Output size is too small")
        losses.MultiscaleRecLoss()(tens(z, "odd_a", dev)[:, :, :3], tens(z, "odd_b", dev)[:, :, :3])


@pytest.mark.parametrize("backend", BACKENDS)
def test_rmsprop_matches_torch(backend):
    dev = use_backend(backend)
    torch.manual_seed(3)
    shapes = [(5, 3, 3, 3), (7,), (4, 5, 1, 1)]
    ps = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    ref_ps = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    ora_ps = [p.detach().clone() for p in ps]
    ora_v = [torch.zeros_like(p) for p in ps]
    ref = torch.optim.RMSprop(ref_ps, lr=1e-2, alpha=0.9)              # trainer.py:341
    ps = [torch.nn.Parameter(p.detach().to(dev)) for p in ps]
    opt = variants.FusedRMSprop(ps, 1e-2, 0.9)
    for step in range(4):
        gs = [torch.randn(s) * (1 + step) for s in shapes]
        opt.zero_grad()
        for p, rp, g in zip(ps, ref_ps, gs):
            p.grad.copy_(g.to(dev))
            rp.grad = g.clone()
        opt.step()
        ref.step()
        O.rmsprop_step(ora_ps, gs, ora_v, 1e-2, 0.9)
        for p, rp, op in zip(ps, ref_ps, ora_ps):
            assert rel(p, rp) < 1e-6 and rel(op, rp) < 1e-6
    # torch.optim.RMSprop's checkpoint format, both ways
    sd, rsd = opt.state_dict(), ref.state_dict()
    assert set(sd["state"][0].keys()) <= set(rsd["state"][0].keys()) and "square_avg" in sd["state"][0]
    for k in ("lr", "alpha", "eps", "momentum", "centered", "weight_decay"):
        assert sd["param_groups"][0][k] == rsd["param_groups"][0][k], k
    for i in range(3):
        assert rel(sd["state"][i]["square_avg"], rsd["state"][i]["square_avg"]) < 1e-6
    opt2 = variants.FusedRMSprop([torch.nn.Parameter(p.detach().clone()) for p in ps], 1.0, 0.5)
    opt2.load_state_dict({"state": {i: {"step": st["step"], "square_avg": st["square_avg"]} for i, st in rsd["state"].items()},
                          "param_groups": rsd["param_groups"]})
    assert opt2.lr == 1e-2 and opt2.alpha == 0.9 and opt2.step_count == 4


@pytest.mark.parametrize("backend", BACKENDS)
def test_trainer_with_non_default_flags(backend):
    """One optimiser step of trainer.py:77-119 with BatchNorm + Swish generator, InstanceNorm + SELU discriminator without spectral
    norm, 'rals' loss and RMSprop, against the same step assembled from torch autograd over this repo's modules evaluated per line:
    the fused path must switch itself off, the losses must be finite and every parameter must move."""
    dev = use_backend(backend)
    ops.set_compute_dtype(torch.float32)
    torch.manual_seed(5)
    G = models.Generator(8, "BatchNorm", "Swish", True).to(dev)
    D = models.Discriminator(8, "InstanceNorm", "SELU", False, "rals").to(dev)
    percep = losses.PerceptualLoss(vgg_weights="seeded", width_div=8).to(dev)
    tr = Trainer(G, D, percep=percep, pool_size=4, adv_loss_type="rals", optimizer_type="rmsprop", alpha=0.9)
    assert tr.fused_passes is False and isinstance(tr.g_optimizer, variants.FusedRMSprop) and tr.criterionGAN.gan_mode == "rals"
    before = {k: v.detach().clone() for k, v in list(G.named_parameters()) + list(D.named_parameters())}
    g = torch.Generator().manual_seed(1)
    raw = (torch.rand(1, 3, 96, 96, generator=g) * 2 - 1).to(dev)       # (96 x 96: the smallest map D's fifth scale accepts)
    exp = (torch.rand(1, 3, 96, 96, generator=g) * 2 - 1).to(dev)
    out = tr.train_step(raw, exp)
    vals = tr.loss_items()
    assert all(np.isfinite(v) for v in vals.values()), vals
    # RMSprop has no weight decay: exactly the forward-dead attention parameters (zero gradient) stay put
    still = [k for k, p in list(G.named_parameters()) + list(D.named_parameters()) if float((p.detach() - before[k]).abs().max()) == 0]
    assert len(still) == 15 and all(k.endswith(DEAD) for k in still), still
    assert int(G.enc1.main[2].num_batches_tracked) == 2                 # G ran twice in training mode (trainer.py:85, :112)
    with pytest.raises(NotImplementedError):
        Trainer(G, D, percep=percep, optimizer_type="sgd")
