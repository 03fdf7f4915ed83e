"""Oracle contact ABOVE the 96^2 / 64^2 fixtures for the discriminator, the full-width VGG fidelity loss and the whole training step
(VERDICT r3 "missing" #1): fp32 mode on the MI355X against the CPU oracle on identical weights and inputs, at sizes where D's parity-class
data gradients span several tiles, its 7x7 layers take the interior / frame split, `uegan_sn_act_bwd` sees real map sizes and
`conv_wide_kernel` runs inside the VGG chain.  The oracle itself is pinned to the reference by tests/test_oracle_golden.py.

The oracle is evaluated in FLOAT64 here: at these sizes the step contains discontinuous pieces (the hinge of the relativistic loss on an
8 x 8 map, where one pixel is 1/64 of a scale's gradient; ReLU / InstanceNorm on nearly dead VGG channels) on which TWO fp32 evaluations of
the same formula differ by far more than 1e-3 -- the oracle's own fp32 run deviates from its fp64 run by 1.4e-2 on d4's weight gradient
and by 1.1e-3 on the VGG image gradient (tools/d_cond.py), so an fp32-vs-fp32 comparison would measure which side of a hinge a
rounding error fell on.  Against the exact value of the restated formula the bound is north_star's 1e-3, ELEMENT-WISE relative for every
element whose reference magnitude is above 1e-3 of its tensor's largest (smaller elements: absolute, against that floor).  Where a quantity
is ill-conditioned in fp32 itself, the oracle's fp32 run is the yardstick: see `check`.  Costs ~30 s of CPU in total."""
import random

import pytest
import torch
import torch.nn.functional as F

from helpers import use_backend
from oracle import uegan_oracle as O
from uegan_amd import fused, losses, models, ops, trainer

pytestmark = pytest.mark.gpu
DEAD = ("conv.0.weight", "conv.2.weight", "fuse.0.bias")
TOL = 1e-3
FP32_SLACK = 20.0


F16_PIX_ABS = 5e-3      # fp16 storage, enhanced pixels, max abs against the oracle at 16 x 512^2 (observed 2.2e-3 in round 4)


def _record_vs_oracle(mode, rec):
    """observed deviations of one full-size step against the ORACLE, per storage mode -> gpurun_out/bf16_deviation.json (bench.py embeds the
    committed copy under profiles/)"""
    import json
    import os
    from helpers import ROOT
    path = os.path.join(ROOT, "gpurun_out", "bf16_deviation.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    try:
        d = json.load(open(path))
    except (OSError, ValueError):
        d = {}
    d.setdefault("full_step_16x512_vs_oracle", {})[mode] = {k: float("%.4g" % v) for k, v in rec.items()}
    json.dump(d, open(path, "w"), indent=1, sort_keys=True)


def elem_rel(got, ref, floor=1e-3):
    """max over elements of |got - ref| / max(|ref|, floor * max|ref|)"""
    got, ref = got.detach().double().cpu(), ref.detach().double()
    fl = floor * float(ref.abs().max()) + 1e-300
    return float(((got - ref).abs() / ref.abs().clamp_min(fl)).max())


def rel_max(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double()
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-300))


def check(worst, key, got, ref64, ref32=None):
    """(A) element-wise 1e-3 against the fp64 oracle.  Sums of 10^4..10^7 fp32 terms do not always get there -- the REFERENCE's arithmetic
    does not either: the oracle's own fp32 run misses (A) on the same tensors (e.g. D.d2's bias gradient: 4.4e-5 of the maximum, which is
    2e-3 of an element at the 1e-3 floor) -- so a tensor also passes when (B) it is no further from the exact value than FP32_SLACK x the
    fp32 oracle is, in both the element-wise and the max-norm measure, and within 1e-3 max-norm unless the fp32 oracle itself is beyond
    1e-4 (ill-conditioned quantities: the hinge on an 8 x 8 map, InstanceNorm of nearly dead random-weight VGG channels, and every
    generator gradient downstream of them).  FP32_SLACK = 20: the exact-fp32 MFMA accumulates a K = 4608 reduction as ONE chain of fused
    multiply-adds (error ~ sqrt(K) eps), oneDNN's AVX-512 kernels as 16 interleaved chains -- about 4x less rounding on the deep layers,
    which the ill-conditioned stages amplify; measured 7x (VGG image gradient) to 14x (G.enc4's bias gradient).  The direction of every
    such tensor is checked separately (cos > 0.9999)."""
    e = elem_rel(got, ref64)
    if e < TOL:
        return
    if ref32 is None:
        worst[key] = e
        return
    own_e, own_m, mine_m = elem_rel(ref32, ref64), rel_max(ref32, ref64), rel_max(got, ref64)
    g, r = got.detach().double().cpu().flatten(), ref64.detach().double().flatten()
    cos = float((g * r).sum() / (g.norm() * r.norm() + 1e-300))
    if e <= FP32_SLACK * own_e and mine_m <= FP32_SLACK * own_m and (mine_m < TOL or own_m >= TOL / 10) and cos > 0.9999:
        return
    worst[key] = dict(elem=e, maxnorm=mine_m, fp32_oracle_elem=own_e, fp32_oracle_maxnorm=own_m, cos=cos)


@pytest.fixture(autouse=True)
def _fp32_mode():
    ops.set_compute_dtype(torch.float32)
    yield
    ops.set_compute_dtype(torch.float32)
    ops.set_precise(False)


def _images(B, S, seed):
    g = torch.Generator().manual_seed(seed)
    lo = torch.rand(B, 3, S // 32, S // 32, generator=g)
    x = torch.nn.functional.interpolate(lo, size=(S, S), mode="bicubic", align_corners=False) + 0.05 * torch.randn(B, 3, S, S, generator=g)
    return (x.clamp(0, 1) * 2 - 1).contiguous()


def _leaves(P, dt):
    return {k: (v.clone().to(dt) if k.endswith(O.D_BUFFER_SUFFIXES) else v.clone().to(dt).requires_grad_(True)) for k, v in P.items()}


def test_discriminator_forward_backward_256_against_oracle():
    """models.py:139-182 at 1 x 3 x 256^2, conv_dim 32: five prediction maps, dx, every parameter gradient (incl. torch's spectral-norm
    correction through sigma = u^T W v), u / v after the power iteration -- module path (one autograd node per layer)."""
    dev = use_backend("gpu")
    PD = O.init_params(O.discriminator_param_shapes(32), 42, "default")
    x = _images(1, 256, 3)
    g = torch.Generator().manual_seed(11)
    cots = None
    refs = {}
    for dt in (torch.float64, torch.float32):
        Pr = _leaves(PD, dt)
        xr = x.detach().clone().to(dt).requires_grad_(True)
        preds_r = O.discriminator_forward(Pr, xr, True)
        if cots is None:
            cots = [torch.randn(p.shape, generator=g) for p in preds_r]
        sum((p * c.to(dt)).sum() for p, c in zip(preds_r, cots)).backward()
        refs[dt] = ([p.detach() for p in preds_r], xr.grad, Pr)
    (preds_r, dx_r, Pr), (preds_32, dx_32, P32) = refs[torch.float64], refs[torch.float32]

    D = models.Discriminator(32, "none", "LeakyReLU", True, "rahinge")
    D.load_state_dict(PD)
    D = D.to(dev).train()
    xd = x.to(dev).requires_grad_(True)
    preds = D(xd)
    sum((p * c.to(dev)).sum() for p, c in zip(preds, cots)).backward()
    worst = {}
    for i, (p, pr) in enumerate(zip(preds, preds_r)):
        assert tuple(p.shape) == tuple(pr.shape)
        check(worst, "pred%d" % i, p, pr)
    check(worst, "dx", xd.grad, dx_r, dx_32)
    for k, p in D.named_parameters():
        assert p.grad is not None and Pr[k].grad is not None, k
        check(worst, "grad/" + k, p.grad, Pr[k].grad, P32[k].grad)
    sd = D.state_dict()
    for k in PD:
        if k.endswith(O.D_BUFFER_SUFFIXES):
            check(worst, "uv/" + k, sd[k], Pr[k])
    assert not worst, worst


def _d_update_oracle(PD, exp, fake, raw, dt):
    Pr = _leaves(PD, dt)
    rp = O.discriminator_forward(Pr, exp.to(dt), True)
    fp = O.discriminator_forward(Pr, fake.to(dt), True)
    loss = O.rahinge_loss(rp, fp, True)
    ip = O.discriminator_forward(Pr, raw.to(dt), True)
    loss = loss + O.rahinge_loss(rp, ip, True)
    loss.backward()
    return float(loss.detach()), Pr


def test_fused_discriminator_loss_256_against_oracle():
    """the batched D update the trainer runs (fused.discriminator_loss: three image groups, rahinge behind the heads, sn_act_bwd) at
    3 x (1 x 3 x 256^2): d_loss, every parameter gradient and u / v against trainer.py:90-96 restated by the oracle"""
    dev = use_backend("gpu")
    PD = O.init_params(O.discriminator_param_shapes(32), 42, "default")
    exp, fake, raw = _images(1, 256, 5), _images(1, 256, 6), _images(1, 256, 7)
    loss_r, Pr = _d_update_oracle(PD, exp, fake, raw, torch.float64)
    _, P32 = _d_update_oracle(PD, exp, fake, raw, torch.float32)

    D = models.Discriminator(32, "none", "LeakyReLU", True, "rahinge")
    D.load_state_dict(PD)
    D = D.to(dev).train()
    loss = fused.discriminator_loss(D, [exp.to(dev), fake.to(dev), raw.to(dev)], [(0, 1), (0, 2)], True)
    loss.backward()
    assert abs(float(loss.detach()) - loss_r) < TOL * abs(loss_r)
    worst = {}
    for k, p in D.named_parameters():
        assert p.grad is not None, k
        check(worst, k, p.grad, Pr[k].grad, P32[k].grad)
    sd = D.state_dict()
    for k in PD:
        if k.endswith(O.D_BUFFER_SUFFIXES):
            check(worst, "uv/" + k, sd[k], Pr[k])
    assert not worst, worst


def test_full_width_vgg_fidelity_loss_256_against_oracle():
    """losses.py:22-36 + 120-164 with the full-width (64..512 channel) seeded VGG19 at 1 x 3 x 256^2: the loss and its image gradient.
    The image gradient of this loss on a random-weight VGG is ill-conditioned in fp32 (InstanceNorm divides by the sigma of nearly dead
    channels: the oracle's own fp32 run is 1.1e-3 from its fp64 run), hence the yardstick form of `check` and a direction check."""
    dev = use_backend("gpu")
    V = O.make_vgg_weights(seed=1234, width_div=1)
    x, y = _images(1, 256, 21), _images(1, 256, 22)
    ref = {}
    for dt in (torch.float64, torch.float32):
        xr = x.detach().clone().to(dt).requires_grad_(True)
        lr = O.perceptual_loss({k: v.to(dt) for k, v in V.items()}, (xr + 1.) / 2., (y.to(dt) + 1.) / 2.)
        lr.backward()
        ref[dt] = (float(lr.detach()), xr.grad)
    (l64, g64), (l32, g32) = ref[torch.float64], ref[torch.float32]
    P = losses.PerceptualLoss(vgg_weights=V, width_div=1).to(dev)
    for fz in (True, False):
        P.fused = fz
        xd = x.to(dev).requires_grad_(True)
        l = P(xd, y.to(dev), input_range01=False)
        l.backward()
        assert abs(float(l.detach()) - l64) < TOL * abs(l64), (fz, float(l.detach()), l64)
        worst = {}
        check(worst, "dx", xd.grad, g64, g32)
        assert not worst, (fz, worst)
        gd = xd.grad.double().cpu()
        assert float((gd * g64).sum() / gd.norm() / g64.norm()) > 0.99999, fz


def test_train_step_256_against_oracle():
    """ONE Trainer.train_step (trainer.py:85-119) at 2 x 3 x 256^2, conv_dim 32, full-width VGG, pool 50, against oracle.train_step:
    the five logged losses, the generated images, every parameter's gradient (element-wise), the weights after both Adam updates."""
    dev = use_backend("gpu")
    PG = O.init_params(O.generator_param_shapes(32), 41, "default")
    PD = O.init_params(O.discriminator_param_shapes(32), 42, "default")
    V = O.make_vgg_weights(seed=1234, width_div=1)
    raw, exp = _images(2, 256, 31), _images(2, 256, 32)
    refs = {}
    for dt in (torch.float64, torch.float32):
        S = O.TrainState({k: v.clone().to(dt) for k, v in PG.items()}, {k: v.clone().to(dt) for k, v in PD.items()},
                         {k: v.to(dt) for k, v in V.items()}, pool_size=50, rng=random.Random(1990))
        refs[dt] = (O.train_step(S, raw.to(dt), exp.to(dt), return_grads=True), S)
    (ref, S), (ref32, _) = refs[torch.float64], refs[torch.float32]

    G = models.Generator(32, "none", "LeakyReLU", False)
    D = models.Discriminator(32, "none", "LeakyReLU", True, "rahinge")
    G.load_state_dict(PG)
    D.load_state_dict(PD)
    T = trainer.Trainer(G.to(dev), D.to(dev), losses.PerceptualLoss(vgg_weights=V, width_div=1).to(dev), pool_size=50, rng=random.Random(1990))
    T.train_step(raw.to(dev), exp.to(dev))
    got = T.loss_items()
    for k in ("d_loss", "g_adv", "g_percep", "g_idt", "g_loss"):
        assert abs(got[k] - ref[k]) <= TOL * abs(ref[k]) + 1e-7, (k, got[k], ref[k])
    worst = {}
    check(worst, "fake_exp", T.fake_exp, ref["fake_exp"], ref32["fake_exp"])
    for name, net, key in (("G", G, "g_grads"), ("D", D, "d_grads")):
        for k, p in net.named_parameters():
            if k.endswith(DEAD):
                continue
            check(worst, name + "." + k, p.grad, ref[key][k], ref32[key][k])      # (.grad aliases the optimizer's flat bucket: still this step's gradient)
    assert not worst, worst
    # after Adam: step 1 moves every element by ~lr * sign(g) -- elements whose gradient is rounding noise may move the other way
    for name, net, want, lr in (("G", G, S.G, 1e-4), ("D", D, S.D, 4e-4)):
        sd = {k: v.detach().double().cpu() for k, v in net.state_dict().items()}
        for k, w in want.items():
            if k.endswith(DEAD):
                continue
            diff = (sd[k] - w).abs()
            assert float(diff.max()) <= 2.2 * lr + 1e-3 * float(w.abs().max()), (name, k, float(diff.max()))
            assert float((diff > 1e-3 * (w.abs().max() + lr)).float().mean()) < 0.02, (name, k)


# per-parameter bounds of the fp16 storage mode at 2 x 256^2.  A stored activation carries half an ulp = 2^-12 relative
# error and a parameter gradient sits downstream of 10 .. 40 stored tensors.  Observed (MI355X, round 5, gpurun_out/f16_param_grads.json ->
# profiles/r05_f16_param_grads.json): D's parameters within 2e-2 of their largest element, cos > 0.9999; the generator's within 4.6e-2 except the
# deep encoder (enc4 / enc5 biases and enc5's weight: 7.2e-2, cos 0.9976) -- the tensors that are ill-conditioned in fp32 already (they sit behind
# the hinge on small maps and the InstanceNorm of nearly dead VGG channels: `check` above; the fp32 ORACLE is 1.4e-2 off its own fp64 run on
# D.d4's weight gradient).  Bounds = observed worst x 1.5.
F16_PARAM_MAXNORM = 0.11
F16_PARAM_COS = 0.996


def test_train_step_256_fp16_per_parameter_against_oracle():
    """fp16 storage (libuegan_hip_f16.so) against the fp64 ORACLE at 2 x 3 x 256^2, conv_dim 32, full-width VGG: EVERY parameter gradient of G
    and D on its own (max-norm relative error and cosine), not only the flat bucket's norm and direction."""
    import json
    import os
    from helpers import ROOT
    dev = use_backend("gpu")
    PG = O.init_params(O.generator_param_shapes(32), 41, "default")
    PD = O.init_params(O.discriminator_param_shapes(32), 42, "default")
    V = O.make_vgg_weights(seed=1234, width_div=1)
    raw, exp = _images(2, 256, 31), _images(2, 256, 32)
    dt = torch.float64
    S = O.TrainState({k: v.clone().to(dt) for k, v in PG.items()}, {k: v.clone().to(dt) for k, v in PD.items()},
                     {k: v.to(dt) for k, v in V.items()}, pool_size=50, rng=random.Random(1990))
    ref = O.train_step(S, raw.to(dt), exp.to(dt), return_grads=True)
    ops.set_compute_dtype(torch.float16)
    G = models.Generator(32, "none", "LeakyReLU", False)
    D = models.Discriminator(32, "none", "LeakyReLU", True, "rahinge")
    G.load_state_dict(PG)
    D.load_state_dict(PD)
    T = trainer.Trainer(G.to(dev), D.to(dev), losses.PerceptualLoss(vgg_weights=V, width_div=1).to(dev), pool_size=50, rng=random.Random(1990))
    T.train_step(raw.to(dev), exp.to(dev))
    got = T.loss_items()
    for k in ("d_loss", "g_adv", "g_percep", "g_idt", "g_loss"):
        assert abs(got[k] - ref[k]) <= TOL * abs(ref[k]) + 1e-7, (k, got[k], ref[k])
    rec, bad = {}, {}
    for name, net, key in (("G", G, "g_grads"), ("D", D, "d_grads")):
        for k, p in net.named_parameters():
            if k.endswith(DEAD):
                continue
            g = (p.grad.double().cpu() / T.loss_scale).flatten()
            r = ref[key][k].double().flatten()
            mx = float((g - r).abs().max() / (r.abs().max() + 1e-300))
            cos = float((g * r).sum() / (g.norm() * r.norm() + 1e-300))
            rec[name + "." + k] = (float("%.3g" % mx), float("%.6g" % cos))
            # (a bias gradient of one or three numbers has no direction to speak of)
            if mx > F16_PARAM_MAXNORM or (r.numel() > 8 and cos < F16_PARAM_COS):
                bad[name + "." + k] = rec[name + "." + k]
    out = os.path.join(ROOT, "gpurun_out", "f16_param_grads.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    worst_mx = max(v[0] for v in rec.values())
    worst_cos = min(v[1] for v in rec.values())
    json.dump({"worst_maxnorm": worst_mx, "worst_cos": worst_cos, "per_parameter": rec}, open(out, "w"), indent=1, sort_keys=True)
    assert not bad, bad


# Trained-VGG19 activation statistics.  The seeded stand-in has He-scaled weights: O(1) activations at every depth.  The real
# vgg19-dcbb9e9d.pth (losses.py:43-44; unavailable offline) is not normalised: on ImageNet-normalised photographs its ReLU outputs grow with depth --
# per-tap maxima of the order of 1e1 (relu1_1), 1e2 (relu2_1), several 1e2 (relu3_1), 1e3 (relu4_1) and several 1e2 again at relu5_1; this growth is why
# Gatys et al. (2016, "Image Style Transfer Using CNNs", section 2) rescale the network before using it.  The targets below are those orders of magnitude
# (from the literature and experience with torchvision's weights; NOT measured here -- there is no network), with 3 x headroom at the deepest taps.
TRAINED_VGG_TAP_MAX = {"relu1_1": 15.0, "relu2_1": 150.0, "relu3_1": 800.0, "relu4_1": 3000.0, "relu5_1": 600.0}


def _vgg_with_trained_statistics(V, x01):
    """rescale the seeded VGG19 layer by layer (weights and bias of a conv by one positive factor: ReLU and max-pool commute with it) so that on the
    image batch x01 (in [0, 1]) every conv's output maximum follows a geometric path through TRAINED_VGG_TAP_MAX"""
    mean = torch.tensor(O.IMAGENET_MEAN).view(1, -1, 1, 1)
    std = torch.tensor(O.IMAGENET_STD).view(1, -1, 1, 1)
    h = (x01 - mean) / std
    taps = list(O.VGG_TAPS.items())                    # [(conv idx, name)] in depth order
    tap_idx = [i for i, _ in taps]
    out = {}
    prev_target, prev_pos = float(h.abs().max()), -1
    convs = [i for i in O.VGG_CONV_IDX if i <= tap_idx[-1]]
    layer = 0
    ci = 0
    with torch.no_grad():
        for v in O.VGG_CFG:
            if v == "M":
                h = F.max_pool2d(h, 2, 2)
                layer += 1
                continue
            idx = O.VGG_CONV_IDX[ci]
            if idx > tap_idx[-1]:
                break
            nxt = min(t for t in tap_idx if t >= idx)
            pos, npos = convs.index(idx), convs.index(nxt)
            tgt_next = TRAINED_VGG_TAP_MAX[O.VGG_TAPS[nxt]]
            # geometric interpolation between the previous tap's maximum and the next one's
            target = prev_target * (tgt_next / prev_target) ** ((pos - prev_pos) / float(npos - prev_pos))
            w, b = V["features.%d.weight" % idx], V["features.%d.bias" % idx]
            y = F.relu(F.conv2d(h, w, b, padding=1))
            sc = target / float(y.max())
            out["features.%d.weight" % idx], out["features.%d.bias" % idx] = w * sc, b * sc
            h = y * sc
            if idx == nxt:
                prev_target, prev_pos = tgt_next, pos
            layer += 2
            ci += 1
    for k, v in V.items():
        out.setdefault(k, v)
    return out


def test_fp16_step_with_trained_vgg_statistics_against_oracle():
    """fp16 storage (the mode recommended for accuracy) with VGG19 activations of a TRAINED network's magnitude instead of the He-scaled stand-in's O(1):
    one train step at 2 x 3 x 256^2 with the default loss scale (2^14) -- every stored activation and gradient stays inside fp16's range (finite gradient
    buckets, no inf / nan loss), the five losses within 1e-3 of the fp64 oracle; plain and precise mode."""
    dev = use_backend("gpu")
    PG = O.init_params(O.generator_param_shapes(32), 41, "default")
    PD = O.init_params(O.discriminator_param_shapes(32), 42, "default")
    raw, exp = _images(2, 256, 51), _images(2, 256, 52)
    V = _vgg_with_trained_statistics(O.make_vgg_weights(seed=1234, width_div=1), (raw + 1) / 2)
    # the rescaled network really has those statistics on this batch
    with torch.no_grad():
        mean = torch.tensor(O.IMAGENET_MEAN).view(1, -1, 1, 1)
        std = torch.tensor(O.IMAGENET_STD).view(1, -1, 1, 1)
        tp = O.vgg_taps(V, ((raw + 1) / 2 - mean) / std)
    for t, (name, want) in zip(tp, TRAINED_VGG_TAP_MAX.items()):
        assert abs(float(t.max()) - want) < 1e-3 * want, (name, float(t.max()), want)
    dt = torch.float64
    S = O.TrainState({k: v.clone().to(dt) for k, v in PG.items()}, {k: v.clone().to(dt) for k, v in PD.items()},
                     {k: v.to(dt) for k, v in V.items()}, pool_size=50, rng=random.Random(1990))
    ref = O.train_step(S, raw.to(dt), exp.to(dt), return_grads=True)
    for precise in (False, True):
        ops.set_compute_dtype(torch.float16)
        ops.set_precise(precise)
        G = models.Generator(32, "none", "LeakyReLU", False)
        D = models.Discriminator(32, "none", "LeakyReLU", True, "rahinge")
        G.load_state_dict(PG)
        D.load_state_dict(PD)
        T = trainer.Trainer(G.to(dev), D.to(dev), losses.PerceptualLoss(vgg_weights=V, width_div=1).to(dev), pool_size=50, rng=random.Random(1990))
        assert T.loss_scale == 16384.0
        T.train_step(raw.to(dev), exp.to(dev))
        got = T.loss_items()
        for k in ("d_loss", "g_adv", "g_percep", "g_idt", "g_loss"):
            assert got[k] == got[k] and abs(got[k]) != float("inf"), (precise, k, got[k])
            assert abs(got[k] - ref[k]) <= TOL * abs(ref[k]) + 1e-7, (precise, k, got[k], ref[k])
        for name, net, key in (("G", G, "g_grads"), ("D", D, "d_grads")):
            gg = torch.cat([p.grad.flatten() for k, p in net.named_parameters() if not k.endswith(DEAD)]).double().cpu() / T.loss_scale
            rr = torch.cat([ref[key][k].flatten() for k, p in net.named_parameters() if not k.endswith(DEAD)]).double()
            assert bool(torch.isfinite(gg).all()), (precise, name)
            cos, ratio = float((gg * rr).sum() / gg.norm() / rr.norm()), float(gg.norm() / rr.norm())
            # (regression guards, not accuracy claims: with activations this large the eps of the fidelity loss's InstanceNorm no longer damps the nearly
            # dead channels of a random-weight VGG, and the generator's gradient reacts to 1e-4 changes of the generated pixels with percents -- every
            # variation of the 16-bit forward lands between cos 0.9979 and 0.9999, norm ratio 0.979 and 0.996: tools/diag_trained_vgg.py)
            assert cos > 0.995 and abs(ratio - 1) < 4e-2, (precise, name, cos, ratio)
        del T, G, D
    ops.set_precise(False)


def test_train_step_full_size_16x512_against_oracle():
    """The benchmark's own configuration (config 2 of BASELINE.json: 16 x 3 x 512^2, conv_dim 32, full-width VGG, pool 50) against the
    oracle: ONE step in fp32 mode, the five losses within 1e-3 and the generated batch element-wise.  Batch 16 (32 through the batched
    generator / VGG passes, 48 through the discriminator) is what selects the large-grid launch variants the timed region runs -- the
    2 x 256^2 test above does not reach them.  The oracle runs in fp32 here (~1 min on 16 host threads; fp64 would take ~10): the
    losses are means over >= 10^5 elements and continuous at that resolution."""
    dev = use_backend("gpu")
    PG = O.init_params(O.generator_param_shapes(32), 41, "default")
    PD = O.init_params(O.discriminator_param_shapes(32), 42, "default")
    V = O.make_vgg_weights(seed=1234, width_div=1)
    raw, exp = _images(16, 512, 41), _images(16, 512, 42)
    nt = torch.get_num_threads()
    torch.set_num_threads(min(16, nt))           # (oneDNN oversubscribes badly on the 256-thread host)
    try:
        S = O.TrainState({k: v.clone() for k, v in PG.items()}, {k: v.clone() for k, v in PD.items()}, V, pool_size=50, rng=random.Random(1990))
        ref = O.train_step(S, raw, exp, return_grads=True)
    finally:
        torch.set_num_threads(nt)
    # fp32 mode: north_star's bound.  fp16 storage mode (the same bytes and speed as the benchmarked bf16 mode): the FAST arithmetic against the
    # oracle directly, at the benchmark's size -- five losses within 1e-3, pixels within 5e-3 absolute, gradient buckets within 1 % in norm.
    # bf16 storage -- the dtype BASELINE.json's config names and the headline number runs in -- meets the ORACLE here too, not only its fp32
    # HIP sibling (tests/test_parity_full.py): 8 significant bits cannot reach 1e-3 on pixels, so its bounds are the
    # observed deviations x 1.5, recorded into gpurun_out/bf16_deviation.json (`full_step_16x512_vs_oracle`) and quoted in the bench line.
    # f16p (round 6): fp16 storage with uegan_amd.set_precise -- the generator's full-resolution chain on hi + lo pairs (uegan_conv2d_fwd_ex).  The 16-bit-rate
    # mode that is INSIDE north_star's tolerance: the five losses within TOL and the enhanced pixels within TOL in max-norm (pixels in [-1, 1]; CPU
    # emulation of the same arithmetic, tools/diag_g_hilo.py -- what is left comes from the MFMA-bound deep layers' fp16 tensors; measured 7.8e-4 here.  The
    # maximum is a tail statistic that depends on the WEIGHTS: tests/test_precise.py::test_precise_pixels_across_weight_seeds_recorded has a second seed at 1.02e-3).
    BOUNDS = {"f32": dict(loss=TOL, cos=0.9999, norm=1e-3), "f16": dict(loss=TOL, pix_abs=F16_PIX_ABS, cos=0.9995, norm=1e-2),
              "f16p": dict(loss=TOL, pix_abs=TOL, cos=0.9995, norm=1e-2), "bf16": dict(loss=5e-3, pix_abs=2.5e-2, cos=0.999, norm=4e-2)}
    for mode, dt in (("f32", torch.float32), ("f16", torch.float16), ("f16p", torch.float16), ("bf16", torch.bfloat16)):
        ops.set_compute_dtype(dt)
        ops.set_precise(mode == "f16p")
        bd = BOUNDS[mode]
        G = models.Generator(32, "none", "LeakyReLU", False)
        D = models.Discriminator(32, "none", "LeakyReLU", True, "rahinge")
        G.load_state_dict(PG)
        D.load_state_dict(PD)
        T = trainer.Trainer(G.to(dev), D.to(dev), losses.PerceptualLoss(vgg_weights=V, width_div=1).to(dev), pool_size=50, rng=random.Random(1990))
        T.train_step(raw.to(dev), exp.to(dev))
        got = T.loss_items()
        rec = {k + "_rel": abs(got[k] - ref[k]) / (abs(ref[k]) + 1e-30) for k in ("d_loss", "g_adv", "g_percep", "g_idt", "g_loss")}
        diff = (T.fake_exp.float().cpu() - ref["fake_exp"])
        rec["fake_abs"], rec["fake_rms"] = float(diff.abs().max()), float(diff.pow(2).mean().sqrt())
        rec["fake_elem_rel_floor1e-2"] = elem_rel(T.fake_exp, ref["fake_exp"], floor=1e-2)
        grads = {}
        for name, net, key in (("G", G, "g_grads"), ("D", D, "d_grads")):
            gg = torch.cat([p.grad.flatten() for k, p in net.named_parameters() if not k.endswith(DEAD)]).double().cpu() / T.loss_scale
            rr = torch.cat([ref[key][k].flatten() for k, p in net.named_parameters() if not k.endswith(DEAD)]).double()
            grads[name] = (float((gg * rr).sum() / gg.norm() / rr.norm()), float(gg.norm() / rr.norm()))
            rec[name + "_grad_cos"], rec[name + "_grad_norm_ratio"] = grads[name]
        _record_vs_oracle(mode, rec)
        for k in ("d_loss", "g_adv", "g_percep", "g_idt", "g_loss"):
            assert abs(got[k] - ref[k]) <= bd["loss"] * abs(ref[k]) + 1e-7, (mode, k, got[k], ref[k])
        if mode == "f32":
            # enhanced pixels in [-1, 1]: 1e-3 relative, pixels below 1 % of the range judged against that floor (|error| <= 1e-5)
            assert rec["fake_elem_rel_floor1e-2"] < TOL
        else:
            assert rec["fake_abs"] < bd["pix_abs"], (mode, rec["fake_abs"])
        # gradient buckets: direction and size (element-wise checks against the fp64 oracle: the 2 x 256^2 tests)
        for name, (cos, ratio) in grads.items():
            assert cos > bd["cos"] and abs(ratio - 1) < bd["norm"], (mode, name, cos, ratio)
        del T, G, D
    ops.set_precise(False)
