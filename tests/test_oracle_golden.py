"""The CPU oracle (oracle/uegan_oracle.py) against the fixtures generated from the REFERENCE's own code
(tools/make_golden.py).  Runs without a GPU."""
import random

import numpy as np
import torch

from helpers import golden, tens, rel
from oracle import uegan_oracle as O


def _params(z, prefix):
    return {k[len(prefix):]: tens(z, k) for k in z.files if k.startswith(prefix)}


def test_generator_forward_backward_default():
    z = golden("g_cd8_default.npz")
    P = {k: v.requires_grad_(True) for k, v in _params(z, "param/").items()}
    x = tens(z, "x").requires_grad_(True)
    out, acts = O.generator_forward(P, x, return_acts=True)
    assert rel(out, tens(z, "out")) < 1e-5
    for k in ("x1", "x5", "y1", "y4", "res"):
        assert rel(acts[k], tens(z, "act/" + k)) < 1e-5, k
    (out * tens(z, "r")).sum().backward()
    assert rel(x.grad, tens(z, "gx")) < 1e-4
    for k in z.files:
        if k.startswith("grad/"):
            assert rel(P[k[5:]].grad, tens(z, k)) < 1e-4, k
    assert rel(O.generator_forward(P, tens(z, "xs")), tens(z, "out_s")) < 1e-5


def test_generator_orthogonal_seeded_params():
    z = golden("g_cd8_orthogonal.npz")
    P = O.init_params(O.generator_param_shapes(8), int(z["param_seed"]), "orthogonal")
    cs = np.array([[float(P[k].double().sum()), float(P[k].double().abs().sum()), float((P[k].double() ** 2).sum())] for k in sorted(P)])
    assert np.allclose(cs, z["param_checksums"], rtol=1e-6, atol=1e-9)       # the seeded recipe reproduces
    out = O.generator_forward(P, tens(z, "x"))
    assert rel(out, tens(z, "out")) < 1e-5


def test_discriminator_spectral_norm_state():
    z = golden("d_cd8.npz")
    P = _params(z, "param/")
    P = {k: (v.requires_grad_(True) if not k.endswith(("_u", "_v")) else v) for k, v in P.items()}
    x = tens(z, "x").requires_grad_(True)
    preds = O.discriminator_forward(P, x, True)
    for i, p in enumerate(preds):
        assert rel(p, tens(z, "pred%d" % i)) < 1e-5
    sum((p * tens(z, "r%d" % i)).sum() for i, p in enumerate(preds)).backward()
    assert rel(x.grad, tens(z, "gx")) < 1e-4
    for k in z.files:
        if k.startswith("grad/"):
            assert rel(P[k[5:]].grad, tens(z, k)) < 1e-4, k
        if k.startswith("uv1/"):
            assert rel(P[k[4:]], tens(z, k)) < 1e-5, k
    with torch.no_grad():
        p2 = O.discriminator_forward(P, x, True)          # u, v advance once more
        pe = O.discriminator_forward(P, x, False)         # eval: no iteration
    for i in range(5):
        assert rel(p2[i], tens(z, "pred2_%d" % i)) < 1e-5
        assert rel(pe[i], tens(z, "pred_eval%d" % i)) < 1e-5


def test_losses():
    z = golden("losses.npz")
    reals = [tens(z, "real%d" % i).requires_grad_(True) for i in range(5)]
    fakes = [tens(z, "fake%d" % i).requires_grad_(True) for i in range(5)]
    for name, ford in (("d", True), ("g", False)):
        l = O.rahinge_loss(reals, fakes, ford)
        assert tuple(l.shape) == (1,)
        assert abs(float(l) - float(z["rahinge_%s" % name])) < 1e-6
        gs = torch.autograd.grad(l.sum(), reals + fakes)
        for i in range(5):
            assert rel(gs[i], tens(z, "rahinge_%s_greal%d" % (name, i))) < 1e-5
            assert rel(gs[5 + i], tens(z, "rahinge_%s_gfake%d" % (name, i))) < 1e-5
    a = tens(z, "msl1_a").requires_grad_(True)
    l = O.multiscale_l1(a, tens(z, "msl1_b"))
    assert abs(float(l) - float(z["msl1"])) < 1e-6
    l.backward()
    assert rel(a.grad, tens(z, "msl1_ga")) < 1e-5
    V = _params(z, "vgg8/")
    V2 = O.make_vgg_weights(seed=1234, width_div=8)
    assert all(torch.equal(V[k], V2[k]) for k in V)             # committed weights == seeded recipe
    px = tens(z, "percep_x").requires_grad_(True)
    l = O.perceptual_loss(V, px, tens(z, "percep_y"))
    assert abs(float(l) - float(z["percep"])) / float(z["percep"]) < 1e-5
    l.backward()
    assert rel(px.grad, tens(z, "percep_gx")) < 1e-4


def test_perceptual_full_width_seeded_vgg():
    z = golden("percep_full.npz")
    V = O.make_vgg_weights(seed=1234, width_div=1)
    cs = np.array([[float(V[k].double().sum()), float(V[k].double().abs().sum()), float((V[k].double() ** 2).sum())] for k in sorted(V)])
    assert np.allclose(cs, z["wsum"], rtol=1e-6)
    l = O.perceptual_loss(V, tens(z, "x"), tens(z, "y"))
    assert abs(float(l) - float(z["percep"])) / float(z["percep"]) < 1e-5


def test_train_steps_cd8_default():
    z = golden("train_cd8_default.npz")
    zl = golden("losses.npz")
    V = _params(zl, "vgg8/")
    S = O.TrainState(_params(z, "G_init/"), _params(z, "D_init/"), V, pool_size=3, rng=random.Random(1990))
    for step in range(3):
        o = O.train_step(S, tens(z, "raw%d" % step), tens(z, "exp%d" % step))
        ref = z["losses%d" % step]
        for k, r in zip(("d_loss", "g_adv", "g_percep", "g_idt", "g_loss"), ref):
            assert abs(o[k] - r) <= 2e-4 * abs(r) + 2e-7, (step, k, o[k], r)
    for k in z.files:
        if k.startswith("D2/"):
            ref = tens(z, k)
            assert float((S.D[k[3:]] - ref).abs().max() / (ref.abs().max() + 4e-4)) < 1e-3, k
