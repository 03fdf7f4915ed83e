"""The full training iteration (trainer.py:77-119) against fixtures produced by a transcription of those lines over the
REFERENCE's own modules + torch.optim.Adam (tools/make_golden.py): five logged scalars, generated images, weights and
spectral-norm vectors after each of 3 steps (pool_size 3 so the ImagePool swap branch runs by step 2).

The emulator variants are slow (minutes per step) and only run with UEGAN_SLOW=1; the GPU variants always run."""
import os
import random

import numpy as np
import pytest
import torch

from helpers import BACKENDS, golden, tens, use_backend
from oracle import uegan_oracle as O
from uegan_amd import losses, models, ops, trainer

NAMES = ("d_loss", "g_adv", "g_percep", "g_idt", "g_loss")
DEAD = ("conv.0.weight", "conv.2.weight", "fuse.0.bias")


def _params(z, prefix):
    return {k[len(prefix):]: tens(z, k) for k in z.files if k.startswith(prefix)}


def _build(cd, PG, PD, dev, pool=3):
    zl = golden("losses.npz")
    V = _params(zl, "vgg8/")
    G = models.Generator(cd, "none", "LeakyReLU", False)
    D = models.Discriminator(cd, "none", "LeakyReLU", True, "rahinge")
    G.load_state_dict(PG)
    D.load_state_dict(PD)
    P = losses.PerceptualLoss(vgg_weights=V, width_div=8)
    return trainer.Trainer(G.to(dev), D.to(dev), P.to(dev), pool_size=pool, rng=random.Random(1990)), G, D


def _slow_ok(backend):
    if backend == "emu" and not os.environ.get("UEGAN_SLOW"):
        pytest.skip("emulated full train steps take minutes; set UEGAN_SLOW=1")


def _check_losses(got, ref, step, rtol=1e-3):
    for k, r in zip(NAMES, ref):
        assert abs(got[k] - r) <= rtol * abs(r) + 1e-6, (step, k, got[k], float(r))


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("mode", ["default", "orthogonal"])
def test_train_steps_cd8(backend, mode):
    _slow_ok(backend)
    dev = use_backend(backend)
    ops.set_compute_dtype(torch.float32)
    z = golden("train_cd8_%s.npz" % mode)
    if mode == "default":
        PG, PD = _params(z, "G_init/"), _params(z, "D_init/")
    else:
        PG = O.init_params(O.generator_param_shapes(8), 41, mode)
        PD = O.init_params(O.discriminator_param_shapes(8), 42, mode)
    T, G, D = _build(8, PG, PD, dev)
    nsteps = 3 if backend == "gpu" else 2
    for step in range(nsteps):
        T.train_step(tens(z, "raw%d" % step, dev), tens(z, "exp%d" % step, dev))
        _check_losses(T.loss_items(), z["losses%d" % step], step)
        fake = tens(z, "fake%d" % step)
        assert float((T.fake_exp.cpu() - fake).abs().max()) < 1e-3 * float(fake.abs().max())
        for net, tag, lr in ((G, "G", 1e-4), (D, "D", 4e-4)):
            sd = net.state_dict()
            for k in z.files:
                if k.startswith("%s%d/" % (tag, step)):
                    name = k.split("/", 1)[1]
                    # orthogonal init makes every singular value of the D weights equal, so the power-iteration vectors
                    # u, v are not determined by W: skip them there (pinned by the default set and by test_models)
                    if mode == "orthogonal" and name.endswith(("weight_u", "weight_v")):
                        continue
                    ref = tens(z, k)
                    diff = (sd[name].cpu() - ref).abs()
                    # Adam normalises the gradient: where |g + wd*w| is at fp-noise level (forward-dead GAM parameters,
                    # whose reference gradients ARE fp noise; isolated elements at orthogonal-0.02 init) the reference's own
                    # update is +-lr by the sign of noise.  So: every element within (steps+1)*2.2*lr, and all but a small
                    # fraction within 1e-3 relative.
                    assert float(diff.max()) <= 2.2 * lr * (step + 1) + 1e-3 * float(ref.abs().max()), (step, k, float(diff.max()))
                    # orthogonal-0.02 init: G == identity, so the fidelity-loss gradient into G is pure rounding noise of
                    # IN(VGG(fake)) - IN(VGG(raw)) with fake == raw (SURVEY.md 7 "ill-conditioned"); only the lr bound is
                    # meaningful for G there.  D's gradients are well conditioned in both sets.
                    # The same holds for D at that init: all singular values of every D weight are equal (u, v undetermined) and
                    # the 8-element biases see near-cancelling sums.  So the element-wise agreement is asserted on the
                    # well-conditioned default set only; the orthogonal set pins losses, images and the lr bound.
                    if not name.endswith(DEAD) and mode == "default":
                        frac = float((diff > 1e-3 * (ref.abs().max() + lr)).float().mean())
                        assert frac < 0.02, (step, k, frac)

@pytest.mark.gpu
def test_train_steps_cd32_checksums():
    """reference width (conv_dim 32): losses and per-tensor weight checksums after each of 3 steps"""
    dev = use_backend("gpu")
    ops.set_compute_dtype(torch.float32)
    z = golden("train_cd32_default.npz")
    PG = O.init_params(O.generator_param_shapes(32), 41, "default")
    PD = O.init_params(O.discriminator_param_shapes(32), 42, "default")
    cs = np.array([[float(PG[k].double().sum()), float(PG[k].double().abs().sum()), float((PG[k].double() ** 2).sum())] for k in sorted(PG)])
    assert np.allclose(cs, z["G_init_checksums"], rtol=1e-6, atol=1e-9)
    T, G, D = _build(32, PG, PD, dev)
    for step in range(3):
        T.train_step(tens(z, "raw%d" % step, dev), tens(z, "exp%d" % step, dev))
        _check_losses(T.loss_items(), z["losses%d" % step], step, rtol=2e-4)       # (observed <= 3.4e-5)
        # generated image: signed sum, |.| sum, square sum and the first 13 pixel values
        f = T.fake_exp.double().cpu().reshape(-1)
        fr = z["fake%d" % step]
        got = np.array([float(f.sum()), float(f.abs().sum()), float((f * f).sum())] + f[:13].tolist())
        # (the signed sum cancels from 28 000 to -800: its error scales with the |.| sum)
        assert abs(got[0] - fr[0]) <= 2e-5 * fr[1] and np.allclose(got[1:3], fr[1:3], rtol=2e-4) and np.allclose(got[3:], fr[3:], atol=2e-4), (step, got, fr)
        # every parameter's gradient norm of this step (still in the optimizers' flat buckets) against the reference's autograd
        for net, opt, key in ((G, T.g_optimizer, "ggradnorm%d"), (D, T.d_optimizer, "dgradnorm%d")):
            named = dict(net.named_parameters())
            ref = z[key % step]
            assert len(ref) == len(named)
            for i, k in enumerate(sorted(named)):
                if k.endswith(DEAD):
                    continue          # forward-dead GAM parameters: the reference's gradients are fp noise (exact zeros here)
                n = float(named[k].grad.norm())
                # step 0 starts from identical weights (observed <= 2.2e-4); later steps inherit the +-lr steps Adam makes of rounding-level
                # gradient differences between the two implementations, and bias gradients -- signed sums over every pixel -- are the most
                # cancellation-prone.  The step is bit-reproducible (test_train_steps_are_bit_reproducible), so these are fixed numbers:
                # observed 0.94 % (dec5.0 bias) at step 1, 2.04 % (dec5.1 bias) at step 2
                assert abs(n - ref[i]) <= (5e-4, 1.5e-2, 3e-2)[step] * ref[i] + 1e-7, (step, k, n, ref[i])
        for net, tag in ((G, "G"), (D, "D")):
            sd = net.state_dict()
            ref = z["%ssum%d" % (tag, step)]
            for i, k in enumerate(sorted(sd.keys())):
                t = sd[k].double().cpu()
                if k.endswith(DEAD):
                    continue
                # signed sum (sees sign errors), |.| sum and square sum of every tensor after the Adam update
                scale = ref[i][1] + 1e-6
                # (observed <= 1.1e-4 of the |.| sum after three steps)
                assert abs(float(t.sum()) - ref[i][0]) <= 5e-4 * scale, (step, k, "sum")
                assert abs(float(t.abs().sum()) - ref[i][1]) <= 5e-4 * scale, (step, k, "abs")
                assert abs(float((t * t).sum()) - ref[i][2]) <= 1e-3 * ref[i][2] + 1e-9, (step, k, "sq")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, "f16p"], ids=["f32", "bf16", "f16p"])
def test_train_steps_are_bit_reproducible(dtype):
    """Two runs of three conv_dim-32 steps from the same weights, inputs and ImagePool seed give bit-identical losses, images, weights,
    spectral-norm vectors and Adam moments: every reduction of the step has a fixed summation order (no float atomics)."""
    dev = use_backend("gpu")
    ops.set_compute_dtype(torch.float16 if dtype == "f16p" else dtype)
    ops.set_precise(dtype == "f16p")          # (round 6: fp16 storage with the generator's full-resolution chain on hi + lo pairs)
    try:
        z = golden("train_cd32_default.npz")
        PG = O.init_params(O.generator_param_shapes(32), 41, "default")
        PD = O.init_params(O.discriminator_param_shapes(32), 42, "default")
        runs = []
        for _ in range(2):
            T, G, D = _build(32, PG, PD, dev)
            rec = []
            for step in range(3):
                T.train_step(tens(z, "raw%d" % step, dev), tens(z, "exp%d" % step, dev))
                rec.append(([T.loss_items()[k] for k in NAMES], T.fake_exp.clone()))
            state = {("G", k): v.clone() for k, v in G.state_dict().items()}
            state.update({("D", k): v.clone() for k, v in D.state_dict().items()})
            for tag, opt in (("g", T.g_optimizer), ("d", T.d_optimizer)):
                for i, st in opt.state_dict()["state"].items():
                    state[(tag, i, "m")], state[(tag, i, "v")] = st["exp_avg"].clone(), st["exp_avg_sq"].clone()
            runs.append((rec, state))
        (ra, sa), (rb, sb) = runs
        for step in range(3):
            assert ra[step][0] == rb[step][0], (step, ra[step][0], rb[step][0])
            assert torch.equal(ra[step][1], rb[step][1]), step
        assert sa.keys() == sb.keys()
        for k in sa:
            assert torch.equal(sa[k], sb[k]), k
    finally:
        ops.set_precise(False)
        ops.set_compute_dtype(torch.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_stream_schedules_are_bit_identical(dtype):
    """The step's three schedules -- one stream, VGG passes + identity loss on a second stream, and real_raw's VGG taps computed at the start of
    the step beside the generator's forward (one pass of B instead of half a pass of 2B) -- only reorder independent work: two conv_dim-32 steps
    give bit-identical losses, images and weights under all three."""
    dev = use_backend("gpu")
    ops.set_compute_dtype(dtype)
    try:
        z = golden("train_cd32_default.npz")
        PG = O.init_params(O.generator_param_shapes(32), 41, "default")
        PD = O.init_params(O.discriminator_param_shapes(32), 42, "default")
        runs = []
        for overlap, early in ((False, False), (True, False), (True, True)):
            T, G, D = _build(32, PG, PD, dev)
            T.overlap, T.early_taps = overlap, early
            rec = []
            for step in range(2):
                T.train_step(tens(z, "raw%d" % step, dev), tens(z, "exp%d" % step, dev))
                rec.append(([T.loss_items()[k] for k in NAMES], T.fake_exp.clone()))
            state = {("G", k): v.clone() for k, v in G.state_dict().items()}
            state.update({("D", k): v.clone() for k, v in D.state_dict().items()})
            runs.append((rec, state))
        for (rb, sb) in runs[1:]:
            (ra, sa) = runs[0]
            for step in range(2):
                assert ra[step][0] == rb[step][0], (step, ra[step][0], rb[step][0])
                assert torch.equal(ra[step][1], rb[step][1]), step
            for k in sa:
                assert torch.equal(sa[k], sb[k]), k
    finally:
        ops.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("backend", BACKENDS)
def test_image_pool_call_order(backend):
    """ImagePool.query (utils.py:30-50): pass-through while filling, then uniform()>0.5 -> randint swap; the Python
    `random` call order is part of the behaviour."""
    dev = use_backend(backend)
    pool, ref = trainer.ImagePool(3, random.Random(7)), O.ImagePool(3, random.Random(7))
    g = torch.Generator().manual_seed(0)
    for _ in range(6):
        imgs = torch.rand(2, 3, 4, 4, generator=g)
        a = pool.query(imgs.to(dev))
        b = ref.query(imgs)
        assert torch.equal(a.cpu(), b)
    assert trainer.ImagePool(0).query(imgs) is imgs


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("pool_size,batch", [(3, 4), (5, 2), (50, 16), (2, 7)])
def test_image_pool_ring_matches_reference_semantics(backend, pool_size, batch):
    """the device ring resolves the reference's sequential loop to index tables: batches larger than the pool, and a later
    image of a batch drawing the slot an earlier image of the SAME batch was just stored into, must give the same images"""
    dev = use_backend(backend)
    pool, ref = trainer.ImagePool(pool_size, random.Random(11)), O.ImagePool(pool_size, random.Random(11))
    g = torch.Generator().manual_seed(1)
    for it in range(3 + 60 // batch):
        imgs = torch.rand(batch, 3, 6, 5, generator=g)
        a = pool.query(imgs.to(dev))
        assert torch.equal(a.cpu(), ref.query(imgs)), it
    assert pool.rng.random() == ref.rng.random()                 # same number of draws from the host RNG
    with pytest.raises(RuntimeError):
        pool.query(torch.rand(batch, 3, 4, 4).to(dev))           # image shape changed


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_loss_scale_is_divided_out(mode):
    """Trainer(loss_scale=S): both backward sweeps start from loss * S, the optimizer kernels divide by S (exact for a power of two in fp32,
    within rounding in bf16: every backward op is linear in the incoming gradient) -- what the fp16 storage mode relies on"""
    import random
    from uegan_amd import losses, models, ops, trainer
    dev = use_backend("gpu")
    ops.set_compute_dtype(torch.float32 if mode == "f32" else torch.bfloat16)
    try:
        z = golden("train_cd8_default.npz")
        PG = {k[len("G_init/"):]: tens(z, k) for k in z.files if k.startswith("G_init/")}
        PD = {k[len("D_init/"):]: tens(z, k) for k in z.files if k.startswith("D_init/")}
        zl = golden("losses.npz")
        V = {k[len("vgg8/"):]: tens(zl, k) for k in zl.files if k.startswith("vgg8/")}
        res = []
        for scale in (1.0, 4096.0):
            G = models.Generator(8, "none", "LeakyReLU", False)
            D = models.Discriminator(8, "none", "LeakyReLU", True, "rahinge")
            G.load_state_dict(PG)
            D.load_state_dict(PD)
            T = trainer.Trainer(G.to(dev), D.to(dev), losses.PerceptualLoss(vgg_weights=V, width_div=8).to(dev), pool_size=3, rng=random.Random(1990),
                                loss_scale=scale)
            T.train_step(tens(z, "raw0", dev), tens(z, "exp0", dev))
            res.append((T.loss_items(), T.g_optimizer.flat_grad.clone() / scale, T.d_optimizer.flat_grad.clone() / scale,
                        torch.cat([p.detach().flatten() for p in list(G.parameters()) + list(D.parameters())])))
        (la, gga, gda, wa), (lb, ggb, gdb, wb) = res
        assert la == lb                                                  # the reported losses are unscaled
        tol = 1e-6 if mode == "f32" else 2e-2
        for a, b in ((gga, ggb), (gda, gdb)):
            assert float((a - b).abs().max()) <= tol * float(a.abs().max()) + 1e-12
        assert float((wa - wb).abs().max()) <= (1e-7 if mode == "f32" else 2.2e-4 * 4)      # (bf16: rounding-level gradient differences -> +-lr Adam steps)
    finally:
        ops.set_compute_dtype(torch.float32)


@pytest.mark.gpu
def test_dynamic_loss_scale_skips_overflowing_steps_fp16():
    """Trainer(loss_scale="dynamic") in the fp16 storage mode: a scale far too large overflows the fp16 gradients -- those optimizer steps are
    skipped (weights untouched) and the scale is halved until a sweep is finite; from then on training proceeds with finite weights"""
    import random
    from uegan_amd import losses, models, ops, trainer
    dev = use_backend("gpu")
    ops.set_compute_dtype(torch.float16)
    try:
        z = golden("train_cd8_default.npz")
        PG = {k[len("G_init/"):]: tens(z, k) for k in z.files if k.startswith("G_init/")}
        PD = {k[len("D_init/"):]: tens(z, k) for k in z.files if k.startswith("D_init/")}
        zl = golden("losses.npz")
        V = {k[len("vgg8/"):]: tens(zl, k) for k in zl.files if k.startswith("vgg8/")}
        G = models.Generator(8, "none", "LeakyReLU", False)
        D = models.Discriminator(8, "none", "LeakyReLU", True, "rahinge")
        G.load_state_dict(PG)
        D.load_state_dict(PD)
        T = trainer.Trainer(G.to(dev), D.to(dev), losses.PerceptualLoss(vgg_weights=V, width_div=8).to(dev), pool_size=3, rng=random.Random(1990),
                            loss_scale="dynamic")
        assert T.dynamic_scale and T.loss_scale == 65536.0
        T.loss_scale = 2.0 ** 36                                       # far beyond fp16's range for these gradients
        w0 = torch.cat([p.detach().flatten().clone() for p in list(G.parameters()) + list(D.parameters())])
        T.train_step(tens(z, "raw0", dev), tens(z, "exp0", dev))
        T.sync()
        w1 = torch.cat([p.detach().flatten() for p in list(G.parameters()) + list(D.parameters())])
        assert T.skipped_steps >= 1 and T.loss_scale < 2.0 ** 36 and torch.equal(w0, w1)      # overflow: nothing was applied
        for i in range(60):
            T.train_step(tens(z, "raw%d" % (i % 3), dev), tens(z, "exp%d" % (i % 3), dev))
            T.sync()
            if T._clean_steps >= 2:
                break
        assert T._clean_steps >= 2, (T.loss_scale, T.skipped_steps)
        w2 = torch.cat([p.detach().flatten() for p in list(G.parameters()) + list(D.parameters())])
        assert torch.isfinite(w2).all() and not torch.equal(w0, w2)
        assert all(v == v for v in T.loss_items().values())
    finally:
        ops.set_compute_dtype(torch.float32)
