"""Round 6: uegan_conv2d_fwd_ex -- hi + lo PAIRS of 16-bit planes through the generator's full-resolution layers (`uegan_amd.set_precise`), the product
epilogue (models.py:69 formed in dec4's epilogue) and the residual + clamp epilogue (models.py:70-72 in dec5.1's).

Kernel level: each of the five layer shapes against torch's fp64 convolution of the SAME operands (hi + lo sums), element-wise -- the pair arithmetic
drops only the lo x lo products (2^-22 relative) and accumulates in fp32, so the result pair must sit within a few 1e-6 of the tensor's scale.
Network level: the eval-mode generator against the reference-pinned oracle; the backward of the precise mode (it reads the hi planes: the graph is
the plain mode's) against the plain 16-bit mode.  Every test runs on the CPU emulator (same kernel sources) and on the MI355X."""
import pytest
import torch
import torch.nn.functional as F

from helpers import BACKENDS, use_backend
from oracle import uegan_oracle as O
from uegan_amd import models, ops

DTYPES = [pytest.param(torch.float16, id="f16"), pytest.param(torch.bfloat16, id="bf16")]


def _pair(t, dt):
    hi = t.to(dt)
    lo = (t - hi.float()).to(dt)
    return hi, lo


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def _nchw(t):
    return t.permute(0, 3, 1, 2)


def _ref_conv(x, w, b, pad):
    xp = F.pad(x.double(), (pad,) * 4, mode="reflect") if pad else x.double()
    return F.conv2d(xp, w.double(), None if b is None else b.double())


@pytest.fixture(autouse=True)
def _modes():
    yield
    ops.set_precise(False)
    ops.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("backend", BACKENDS)
def test_conv_pairs_single_source(backend, dt):
    """dec5.0's and ga1's shapes (3x3 / 1x1, 32 -> 32): source pair, weight pair, result pair (+ the InstanceNorm moments on the 1x1)"""
    dev = use_backend(backend)
    ops.set_compute_dtype(dt)
    g = torch.Generator().manual_seed(5)
    B, H, W, C = 2, 32, 48, 32
    for k, act in ((3, ops.ACT_NONE), (1, ops.ACT_NONE), (3, ops.ACT_LRELU)):
        x = torch.randn(B, C, H, W, generator=g)
        w = torch.randn(32, C, k, k, generator=g) / (C * k * k) ** 0.5
        b = torch.randn(32, generator=g) * 0.1 if k == 3 else None
        xh, xl = _pair(_nhwc(x), dt)
        cfg = ops.ConvCfg(1, ops.PAD_REFLECT, act)
        ex = ops.ConvExtras(x1_lo=xl.to(dev), pair_w=True, want_lo=True)
        holder = ops.StatsHolder() if k == 1 else None
        with torch.no_grad():
            y = ops.conv2d(xh.to(dev), None, w.to(dev), None if b is None else b.to(dev), cfg, stats=holder, ex=ex)
        assert ex.taken and ex.y_lo is not None
        wh, wl = _pair(w, dt)
        xs, ws = _nchw(xh.float() + xl.float()), wh.float() + wl.float()
        ref = _ref_conv(xs, ws, b, (k - 1) // 2)
        if act == ops.ACT_LRELU:
            ref = F.leaky_relu(ref, 0.2)
        got = _nchw(y.float().cpu() + ex.y_lo.float().cpu()).double()
        scale = float(ref.abs().max())
        # fp32 accumulation + the dropped lo x lo terms + the result pair's own 2^-2p rounding (p = 11 / 8 bits)
        tol = (2e-6 if dt == torch.float16 else 4e-5) * scale
        assert float((got - ref).abs().max()) < tol, (k, act, float((got - ref).abs().max()), scale)
        # the hi plane alone is the correctly rounded result
        assert float((_nchw(y.float().cpu()).double() - ref).abs().max()) < (2.0 ** (-11 if dt == torch.float16 else -8)) * scale
        if k == 1:
            assert holder.value is not None
            mean, rstd = holder.value[0].cpu().double(), holder.value[1].cpu().double()
            m_ref, v_ref = ref.mean(dim=(2, 3)), ref.var(dim=(2, 3), unbiased=False)
            assert float((mean - m_ref).abs().max()) < 1e-5 * scale
            assert float((rstd - 1.0 / torch.sqrt(v_ref + ops.IN_EPS)).abs().max() / rstd.abs().max()) < 1e-4
            # the InstanceNorm of the pair
            with torch.no_grad():
                n, n_lo = ops.instnorm_pair(y, ex.y_lo, holder.value)
            nref = (ref - m_ref[:, :, None, None]) / torch.sqrt(v_ref + ops.IN_EPS)[:, :, None, None]
            gotn = _nchw(n.float().cpu() + n_lo.float().cpu()).double()
            assert float((gotn - nref).abs().max()) < (2e-5 if dt == torch.float16 else 3e-4) * float(nref.abs().max())


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("backend", BACKENDS)
def test_conv_pairs_first_layer_and_two_sources(backend, dt):
    """enc1's shape (7x7, 3 -> 32: the image's own pair in channels 3..5 of its 8-channel pixels, weights repeated there) and dec4's (3x3, 32 + 32 -> 32:
    second source a pair, LeakyReLU, product with a pair formed in the epilogue)"""
    dev = use_backend(backend)
    ops.set_compute_dtype(dt)
    g = torch.Generator().manual_seed(6)
    B, H, W = 2, 32, 64
    # ---- enc1
    img = torch.rand(B, 3, H, W, generator=g) * 2 - 1
    w = torch.randn(32, 3, 7, 7, generator=g) / (3 * 49) ** 0.5
    b = torch.randn(32, generator=g) * 0.1
    cfg = ops.ConvCfg(1, ops.PAD_REFLECT, ops.ACT_LRELU)
    ex = ops.ConvExtras(pair_w=True, dup_cin=True, want_lo=True)
    with torch.no_grad():
        xin = ops.to_nhwc(img.to(dev), pair=True)
        y = ops.conv2d(xin, None, w.to(dev), b.to(dev), cfg, ex=ex)
    assert ex.taken
    xi = xin.float().cpu()
    assert torch.equal(xi[..., 6:], torch.zeros_like(xi[..., 6:]))
    img_pair = _nchw(xi[..., :3] + xi[..., 3:6])
    assert float((img_pair - img).abs().max()) < (1e-6 if dt == torch.float16 else 2e-5)
    wh, wl = _pair(w, dt)
    ref = F.leaky_relu(_ref_conv(img_pair, wh.float() + wl.float(), b, 3), 0.2)
    got = _nchw(y.float().cpu() + ex.y_lo.float().cpu()).double()
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) < (2e-6 if dt == torch.float16 else 4e-5) * scale
    # ---- dec4 + product
    C = 32
    u = torch.randn(B, C, H, W, generator=g)
    a = torch.randn(B, C, H, W, generator=g)
    m = torch.randn(B, C, H, W, generator=g)
    w4 = torch.randn(32, 2 * C, 3, 3, generator=g) / (2 * C * 9) ** 0.5
    b4 = torch.randn(32, generator=g) * 0.1
    uh = _nhwc(u).to(dt)
    ah, al = _pair(_nhwc(a), dt)
    mh, ml = _pair(_nhwc(m), dt)
    ex4 = ops.ConvExtras(x2_lo=al.to(dev), pair_w=True, mul=mh.to(dev), mul_lo=ml.to(dev), want_mul_lo=True)
    with torch.no_grad():
        y4 = ops.conv2d(uh.to(dev), ah.to(dev), w4.to(dev), b4.to(dev), cfg, ex=ex4)
    assert ex4.taken and ex4.prod is not None and ex4.prod_lo is not None
    w4h, w4l = _pair(w4, dt)
    src = torch.cat([_nchw(uh.float()), _nchw(ah.float() + al.float())], 1)
    ref4 = F.leaky_relu(_ref_conv(src, w4h.float() + w4l.float(), b4, 1), 0.2)
    s4 = float(ref4.abs().max())
    assert float((_nchw(y4.float().cpu()).double() - ref4).abs().max()) < (2.0 ** (-11 if dt == torch.float16 else -8)) * s4
    refp = ref4 * _nchw(mh.float() + ml.float()).double()
    gotp = _nchw(ex4.prod.float().cpu() + ex4.prod_lo.float().cpu()).double()
    assert float((gotp - refp).abs().max()) < (3e-6 if dt == torch.float16 else 6e-5) * float(refp.abs().max())
    # ---- the product epilogue on plain operands (what every 16-bit mode runs): same product, rounded once
    exq = ops.ConvExtras(mul=mh.to(dev))
    with torch.no_grad():
        yq = ops.conv2d(uh.to(dev), ah.to(dev), w4.to(dev), b4.to(dev), cfg, ex=exq)
    assert exq.taken
    refq = F.leaky_relu(_ref_conv(torch.cat([_nchw(uh.float()), _nchw(ah.float())], 1), w4h.float(), b4, 1), 0.2)
    sq = float(refq.abs().max())
    half = 2.0 ** (-11 if dt == torch.float16 else -8)
    assert float((_nchw(yq.float().cpu()).double() - refq).abs().max()) < half * sq + 2e-5 * sq
    refqp = refq * _nchw(mh.float()).double()
    assert float((_nchw(exq.prod.float().cpu()).double() - refqp).abs().max()) < half * float(refqp.abs().max()) + 2e-5 * float(refqp.abs().max())


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("backend", BACKENDS)
def test_weight_pairs_on_plain_sources(backend, dt):
    """enc2's shape (3x3 stride 2, 32 -> 64: the pair as ONE [hi | lo] matrix, the source's channels read twice by the 64-channel stride-2 kernel) and
    upsample4's (1x1, 64 -> 32): plain source and result, weights as a pair -- the result is the correctly rounded conv with W = Whi + Wlo"""
    dev = use_backend(backend)
    ops.set_compute_dtype(dt)
    g = torch.Generator().manual_seed(8)
    half = 2.0 ** (-11 if dt == torch.float16 else -8)
    for (cin, cout, k, stride, act) in ((32, 64, 3, 2, ops.ACT_LRELU), (64, 32, 1, 1, ops.ACT_NONE)):
        B, H, W = 2, 32, 64
        x = torch.randn(B, cin, H, W, generator=g)
        w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
        b = torch.randn(cout, generator=g) * 0.1
        xh = _nhwc(x).to(dt)
        cfg = ops.ConvCfg(stride, ops.PAD_REFLECT, act)
        ex = ops.ConvExtras(pair_w=True)
        with torch.no_grad():
            y = ops.conv2d(xh.to(dev), None, w.to(dev), b.to(dev), cfg, ex=ex)
            y0 = ops.conv2d(xh.to(dev), None, w.to(dev), b.to(dev), cfg)
        assert ex.taken
        wh, wl = _pair(w, dt)
        pad = (k - 1) // 2
        xp = F.pad(_nchw(xh.float()).double(), (pad,) * 4, mode="reflect") if pad else _nchw(xh.float()).double()
        ref = F.conv2d(xp, (wh.float() + wl.float()).double(), b.double(), stride=stride)
        ref0 = F.conv2d(xp, wh.float().double(), b.double(), stride=stride)
        if act == ops.ACT_LRELU:
            ref, ref0 = F.leaky_relu(ref, 0.2), F.leaky_relu(ref0, 0.2)
        sc = float(ref.abs().max())
        e_pair = float((_nchw(y.float().cpu()).double() - ref).abs().max())
        e_plain_vs_pair_ref = float((_nchw(y0.float().cpu()).double() - ref).abs().max())
        assert e_pair < half * sc + 2e-5 * sc, (cin, cout, e_pair, sc)
        assert float((_nchw(y0.float().cpu()).double() - ref0).abs().max()) < half * sc + 2e-5 * sc
        # the pair result is measurably closer to the exact-weights convolution than the plain one whenever the weight rounding is visible at all
        assert e_pair <= e_plain_vs_pair_ref + 1e-7 * sc
        # backward through the same node: unchanged (plain packs)
        xg = xh.to(dev).clone().requires_grad_(True)
        wg = w.to(dev).clone().requires_grad_(True)
        out = ops.conv2d(xg, None, wg, None, cfg, ex=ops.ConvExtras(pair_w=True))
        out.float().sum().backward()
        xg0 = xh.to(dev).clone().requires_grad_(True)
        wg0 = w.to(dev).clone().requires_grad_(True)
        out0 = ops.conv2d(xg0, None, wg0, None, cfg)
        out0.float().sum().backward()
        if act == ops.ACT_NONE:      # (with an activation the mask follows the slightly different forward values)
            assert torch.equal(xg.grad, xg0.grad) and torch.equal(wg.grad, wg0.grad)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("backend", BACKENDS)
def test_toeplitz_pairs_and_residual_epilogue(backend, dt):
    """dec5.1 (7x7, 32 -> 3, tanh): source + weight pairs, clamp(tanh(conv) + x, -1, 1) written as NCHW fp32 for one and for two image sets; and the
    residual epilogue alone on plain operands"""
    dev = use_backend(backend)
    ops.set_compute_dtype(dt)
    g = torch.Generator().manual_seed(7)
    B, H, W, C = 3, 32, 64, 32
    x = torch.randn(B, C, H, W, generator=g) * 0.5
    w = torch.randn(3, C, 7, 7, generator=g) / (C * 49) ** 0.5
    b = torch.randn(3, generator=g) * 0.1
    img = torch.rand(B, 3, H, W, generator=g) * 2 - 1
    xh, xl = _pair(_nhwc(x), dt)
    wh, wl = _pair(w, dt)
    cfg = ops.ConvCfg(1, ops.PAD_REFLECT, ops.ACT_TANH)
    for sets in (1, 2):
        xs = (img.to(dev),) if sets == 1 else (img[:1].contiguous().to(dev), img[1:].contiguous().to(dev))
        for pairs in (True, False):
            ex = ops.ConvExtras(x1_lo=xl.to(dev), pair_w=True, res=xs) if pairs else ops.ConvExtras(res=xs)
            with torch.no_grad():
                res = ops.conv2d(xh.to(dev), None, w.to(dev), b.to(dev), cfg, ex=ex)
            assert ex.taken and len(ex.res_out) == sets
            src = _nchw(xh.float() + xl.float()) if pairs else _nchw(xh.float())
            t = torch.tanh(_ref_conv(src, (wh.float() + wl.float()) if pairs else wh.float(), b, 3))
            want = torch.clamp(t + img.double(), -1, 1)
            got = torch.cat([o.cpu() for o in ex.res_out]).double()
            assert float((got - want).abs().max()) < (3e-6 if dt == torch.float16 else 5e-5), (sets, pairs, float((got - want).abs().max()))
            # `res` itself (the backward's tanh'): the 16-bit rounding of tanh, channels 3.. zero
            r = res.float().cpu()
            assert torch.equal(r[..., 3:], torch.zeros_like(r[..., 3:]))
            assert float((_nchw(r[..., :3]).double() - t).abs().max()) < 2.0 ** (-11 if dt == torch.float16 else -8)


@pytest.mark.parametrize("backend", BACKENDS)
def test_generator_precise_against_oracle(backend):
    """the eval-mode generator (conv_dim 32) in fp16 storage: the precise mode is >= 2.5 x closer to the fp32 oracle than the plain mode in max-norm and rms
    (4 x at 16 x 512^2: tests/test_oracle_at_size.py), one image set and two (forward_pair); in fp32 mode set_precise changes nothing"""
    dev = use_backend(backend)
    P = O.init_params(O.generator_param_shapes(32), 41, "default")
    g = torch.Generator().manual_seed(3)
    S = (32, 64) if backend == "emu" else (128, 160)
    x = torch.rand(2, 3, *S, generator=g) * 2 - 1
    with torch.no_grad():
        ref = O.generator_forward(P, x)
    G = models.Generator(32, "none", "LeakyReLU", False)
    G.load_state_dict(P)
    G = G.to(dev).eval()
    err = {}
    for prec in (False, True):
        ops.set_compute_dtype(torch.float16)
        ops.set_precise(prec)
        with torch.no_grad():
            out = G(x.to(dev)).cpu()
            oa, ob = G.forward_pair(x[:1].contiguous().to(dev), x[1:].contiguous().to(dev))
        assert torch.equal(torch.cat([oa.cpu(), ob.cpu()]), out)
        d = out - ref
        err[prec] = (float(d.abs().max()), float(d.pow(2).mean().sqrt()))
    assert err[True][0] < 0.4 * err[False][0] and err[True][1] < 0.4 * err[False][1], err
    assert err[True][0] < 6e-4, err
    ops.set_compute_dtype(torch.float32)
    ops.set_precise(True)
    with torch.no_grad():
        out32 = G(x.to(dev)).cpu()
    assert float((out32 - ref).abs().max()) < 2e-5


@pytest.mark.parametrize("backend", BACKENDS)
def test_precise_backward_is_the_plain_backward(backend):
    """the backward sweep of the precise mode reads the hi planes and the plain packed weights: same graph, same kernels as the plain fp16 mode.  Against
    the fp32 mode's gradients the precise mode is therefore no further off than the plain fp16 mode (its forward values are closer; the deep encoder's
    gradients on a map this small are ill-conditioned in any 16-bit arithmetic, hence the comparison instead of an absolute bound)"""
    dev = use_backend(backend)
    P = O.init_params(O.generator_param_shapes(32), 41, "default")
    g = torch.Generator().manual_seed(4)
    x = (torch.rand(1, 3, 32, 64, generator=g) * 2 - 1).to(dev)
    t = (torch.rand(1, 3, 32, 64, generator=g) * 2 - 1).to(dev)
    grads = {}
    for mode, dt, prec in (("f32", torch.float32, False), ("plain", torch.float16, False), ("precise", torch.float16, True)):
        ops.set_compute_dtype(dt)
        ops.set_precise(prec)
        G = models.Generator(32, "none", "LeakyReLU", False)
        G.load_state_dict(P)
        G = G.to(dev).train()
        out = G(x)
        (((out - t) ** 2).mean() * 1024.0).backward()
        grads[mode] = {k: p.grad.detach().float().cpu().clone() for k, p in G.named_parameters() if p.grad is not None}
    assert grads["precise"].keys() == grads["plain"].keys() == grads["f32"].keys()
    flat = {m: torch.cat([v.double().flatten() for v in grads[m].values()]) for m in grads}
    e_plain = float((flat["plain"] - flat["f32"]).norm() / flat["f32"].norm())
    e_prec = float((flat["precise"] - flat["f32"]).norm() / flat["f32"].norm())
    assert e_prec < 1.2 * e_plain + 1e-3, (e_prec, e_plain)
    for k in grads["f32"]:
        r = grads["f32"][k].double().flatten()
        if float(r.norm()) == 0.0:
            assert float(grads["precise"][k].norm()) == 0.0, k
            continue
        ep = float((grads["plain"][k].double().flatten() - r).norm() / r.norm())
        eq = float((grads["precise"][k].double().flatten() - r).norm() / r.norm())
        assert eq < 2.0 * ep + 1e-2, (k, eq, ep)


@pytest.mark.parametrize("backend", BACKENDS)
def test_precise_refuses_other_generators(backend):
    """the pair kernels exist for the reference's default generator at conv_dim 32: another width fails loudly instead of running the plain mode"""
    dev = use_backend(backend)
    ops.set_compute_dtype(torch.float16)
    ops.set_precise(True)
    G = models.Generator(8, "none", "LeakyReLU", False).to(dev).eval()
    with pytest.raises(RuntimeError, match="conv_dim 32"):
        with torch.no_grad():
            G(torch.zeros(1, 3, 32, 32, device=dev))
    ops.set_precise(False)
    with torch.no_grad():
        G(torch.zeros(1, 3, 32, 32, device=dev))


@pytest.mark.gpu
def test_precise_pixels_across_weight_seeds_recorded():
    """How far the claim "enhanced pixels within 1e-3 of the fp32 reference" reaches.  The RMS error of the precise fp16 mode is 5 - 6e-5 for every weight
    initialisation and image set tried; its MAXIMUM over ~10^7 pixels is a 15 - 20 sigma tail statistic that depends on the weights (the tail is the
    systematic rounding of the deep layers' fp16 weights and activations, tools/diag_g_hilo.py): 6.4 - 7.8e-4 with the oracle tests' seed-41 weights on every
    image set, 0.9 - 1.2e-3 (CPU emulation) with seed 5.  Asserted here for both seeds at 4 x 512^2: rms < 1e-4 (north_star's bound / 10), maximum < 1.5e-3 and
    >= 2 x smaller than the plain fp16 mode's; the observed values go to gpurun_out/precise_weight_seeds.json (committed copy under profiles/)."""
    import json
    import os
    from helpers import ROOT
    dev = use_backend("gpu")
    g = torch.Generator().manual_seed(123)
    lo = torch.rand(4, 3, 16, 16, generator=g)
    x = F.interpolate(lo, size=(512, 512), mode="bicubic", align_corners=False) + 0.05 * torch.randn(4, 3, 512, 512, generator=g)
    x = (x.clamp(0, 1) * 2 - 1).contiguous()
    rec = {}
    for seed in (41, 5):
        P = O.init_params(O.generator_param_shapes(32), seed, "default")
        with torch.no_grad():
            ref = O.generator_forward(P, x)
        G = models.Generator(32, "none", "LeakyReLU", False)
        G.load_state_dict(P)
        G = G.to(dev).eval()
        for prec in (False, True):
            ops.set_compute_dtype(torch.float16)
            ops.set_precise(prec)
            with torch.no_grad():
                d = G(x.to(dev)).cpu() - ref
            rec["seed%d_%s" % (seed, "precise" if prec else "plain")] = {"max": float("%.4g" % float(d.abs().max())), "rms": float("%.4g" % float(d.pow(2).mean().sqrt())),
                                                                         "pixels_above_1e-3": int((d.abs() > 1e-3).sum()), "pixels": d.numel()}
        a, b = rec["seed%d_precise" % seed], rec["seed%d_plain" % seed]
        assert a["rms"] < 1e-4 and a["max"] < 1.5e-3 and a["max"] < 0.5 * b["max"], (seed, a, b)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rec, open(os.path.join(ROOT, "gpurun_out", "precise_weight_seeds.json"), "w"), indent=1, sort_keys=True)
