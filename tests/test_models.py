"""Generator / Discriminator modules against the fixtures generated from the REFERENCE's models.py (tests/golden):
same state-dict keys, outputs, input/parameter gradients, spectral-norm u/v evolution, train vs eval behaviour."""
import os

import pytest
import torch

from helpers import BACKENDS, golden, rel, tens, use_backend
from oracle import uegan_oracle as O
from uegan_amd import models, ops

TOL = 1e-4          # north_star gate is 1e-3 relative; fp32 path sits ~1e-6


def _params(z, prefix):
    return {k[len(prefix):]: tens(z, k) for k in z.files if k.startswith(prefix)}


DEAD = ("conv.0.weight", "conv.2.weight", "fuse.0.bias")      # GAM parameters that are forward-dead under InstanceNorm


@pytest.mark.parametrize("backend", BACKENDS)
def test_generator_matches_reference(backend):
    dev = use_backend(backend)
    ops.set_compute_dtype(torch.float32)
    z = golden("g_cd8_default.npz")
    P = _params(z, "param/")
    G = models.Generator(8, "none", "LeakyReLU", False)
    assert set(G.state_dict().keys()) == set(P.keys())                       # drop-in state-dict
    assert list(G.state_dict().keys())[:2] == ["enc1.main.1.weight", "enc1.main.1.bias"]
    G.load_state_dict(P)
    G = G.to(dev)
    x = tens(z, "x", dev).requires_grad_(True)
    out = G(x)
    assert out.dtype == torch.float32 and out.shape == x.shape
    assert rel(out, tens(z, "out")) < TOL
    (out * tens(z, "r", dev)).sum().backward()
    assert rel(x.grad, tens(z, "gx")) < TOL
    named = dict(G.named_parameters())
    for k in z.files:
        if k.startswith("grad/"):
            assert rel(named[k[5:]].grad, tens(z, k)) < TOL, k
        if k.startswith("gradnorm/") and not k.endswith(DEAD):
            gn = float(z[k])
            assert abs(float(named[k[9:]].grad.norm()) - gn) < 1e-3 * gn + 1e-7, k
    for n, p in named.items():                                               # dead GAM params: exactly zero, never None
        if n.endswith(DEAD):
            assert p.grad is not None and float(p.grad.abs().max()) == 0.0, n
    with torch.no_grad():
        G.eval()
        assert rel(G(tens(z, "xs", dev)), tens(z, "out_s")) < TOL           # 32x32: smallest legal input, eval == train


@pytest.mark.parametrize("backend", BACKENDS)
def test_generator_orthogonal_init_set(backend):
    """the reference's default init (orthogonal, gain 0.02): G is ~identity, outputs must still agree"""
    dev = use_backend(backend)
    ops.set_compute_dtype(torch.float32)
    z = golden("g_cd8_orthogonal.npz")
    P = O.init_params(O.generator_param_shapes(8), int(z["param_seed"]), "orthogonal")
    G = models.Generator(8, "none", "LeakyReLU", False)
    G.load_state_dict(P)
    G = G.to(dev)
    with torch.no_grad():
        out = G(tens(z, "x", dev))
    assert rel(out, tens(z, "out")) < TOL


@pytest.mark.parametrize("backend", BACKENDS)
def test_generator_rejects_bad_shapes(backend):
    dev = use_backend(backend)
    G = models.Generator(8, "none", "LeakyReLU", False).to(dev)
    for shp in ((1, 3, 56, 56), (1, 3, 16, 16), (1, 1, 32, 32)):             # reference fails on these too
        with pytest.raises(RuntimeError):
            G(torch.zeros(*shp, device=dev))
    with pytest.raises(NotImplementedError):                                 # unknown flag values raise like the reference's get_*_fun
        models.Generator(8, "LayerNorm", "LeakyReLU", False)
    with pytest.raises(NotImplementedError):
        models.Generator(8, "none", "GELU", False)
    with pytest.raises(NotImplementedError):
        models.Discriminator(8, "none", "LeakyReLU", True, "wgan-gp")


@pytest.mark.parametrize("backend", BACKENDS)
def test_discriminator_matches_reference(backend):
    dev = use_backend(backend)
    ops.set_compute_dtype(torch.float32)
    z = golden("d_cd8.npz")
    P = _params(z, "param/")
    D = models.Discriminator(8, "none", "LeakyReLU", True, "rahinge")
    assert set(D.state_dict().keys()) == set(P.keys())
    D.load_state_dict(P)
    D = D.to(dev)
    D.train()
    x = tens(z, "x", dev).requires_grad_(True)
    preds = D(x)
    assert [tuple(p.shape) for p in preds] == [(2, 1, 48, 48), (2, 1, 24, 24), (2, 1, 12, 12), (2, 1, 6, 6), (2, 1, 3, 3)]
    for i, p in enumerate(preds):
        assert rel(p, tens(z, "pred%d" % i)) < TOL
    sum((p * tens(z, "r%d" % i, dev)).sum() for i, p in enumerate(preds)).backward()
    assert rel(x.grad, tens(z, "gx")) < TOL
    named = dict(D.named_parameters())
    for k in z.files:
        if k.startswith("grad/"):
            assert rel(named[k[5:]].grad, tens(z, k)) < TOL, k
        if k.startswith("gradnorm/"):
            gn = float(z[k])
            assert abs(float(named[k[9:]].grad.norm()) - gn) < 1e-3 * gn + 1e-7, k
    sd = D.state_dict()
    for k in z.files:
        if k.startswith("uv1/"):
            assert rel(sd[k[4:]], tens(z, k)) < TOL, k                       # one power iteration happened, in place
    with torch.no_grad():
        p2 = D(x)                                                            # second TRAINING forward: u, v advance again
    sd = D.state_dict()
    for i in range(5):
        assert rel(p2[i], tens(z, "pred2_%d" % i)) < TOL
    for k in z.files:
        if k.startswith("uv2/"):
            assert rel(sd[k[4:]], tens(z, k)) < TOL, k
    D.eval()
    with torch.no_grad():
        pe = D(x)                                                            # eval: no iteration
        pe2 = D(x)
    for i in range(5):
        assert rel(pe[i], tens(z, "pred_eval%d" % i)) < TOL
        assert torch.equal(pe[i], pe2[i])
    with pytest.raises(RuntimeError, match="Padding size"):                 # reference D cannot run at 64x64 either
        D(torch.zeros(1, 3, 64, 64, device=dev))


@pytest.mark.parametrize("backend", BACKENDS)
def test_init_weights_hook_reaches_our_modules(backend):
    """trainer.py:357-390 `init_weights` keys on class names containing 'Conv' with a `.weight`: it must land in our
    parameters (incl. weight_orig of the spectral-normed convs)."""
    use_backend(backend)
    D = models.Discriminator(8, "none", "LeakyReLU", True, "rahinge")
    G = models.Generator(8, "none", "LeakyReLU", False)
    hit = []

    def init_func(m):                                                        # restated from the reference's behaviour
        classname = m.__class__.__name__
        if hasattr(m, "weight") and classname.find("Conv") != -1:
            torch.nn.init.orthogonal_(m.weight.data, gain=0.02)
            if hasattr(m, "bias") and m.bias is not None:
                torch.nn.init.constant_(m.bias.data, 0.0)
            hit.append(classname)

    D.apply(init_func)
    G.apply(init_func)
    assert len(hit) == 10 + 30
    w = D.d3[0][1].weight_orig
    assert abs(float(w.std()) - 0.02 / (w[0].numel() ** 0.5)) < 2e-5
    assert float(D.d1[0][1].bias.abs().max()) == 0.0


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("fmt", ["bf16", "f16"])
def test_bf16_mode_close_to_fp32(backend, fmt):
    """16-bit storage (bf16: the throughput mode; f16: the fp16-format build of the library) stays within that format's rounding of the fp32
    parity path -- the generator forward against the reference fixture"""
    dev = use_backend(backend)
    z = golden("g_cd8_default.npz")
    G = models.Generator(8, "none", "LeakyReLU", False)
    G.load_state_dict(_params(z, "param/"))
    G = G.to(dev).eval()
    try:
        ops.set_compute_dtype(torch.bfloat16 if fmt == "bf16" else torch.float16)
        with torch.no_grad():
            out = G(tens(z, "xs", dev))
    finally:
        ops.set_compute_dtype(torch.float32)
    assert float((out.cpu() - tens(z, "out_s")).abs().max()) < (0.05 if fmt == "bf16" else 0.008)


def test_data_parallel_replicas_are_refused():
    """nn.DataParallel with several device ids replicates the module per forward (trainer.py:317-321): the replicas raise"""
    G = models.Generator(8, "none", "LeakyReLU", False)
    G._is_replica = True            # what torch.nn.parallel.replicate sets on every replica
    with pytest.raises(RuntimeError, match="one process per GPU"):
        G(torch.zeros(1, 3, 32, 32))
