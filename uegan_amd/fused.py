"""Whole-network functions: the kernels of a FIXED network sequenced explicitly, forward and backward, inside one
torch.autograd.Function each -- instead of one autograd node per layer (uegan_amd/ops.py), which costs an elementwise `add`
kernel wherever an activation has two consumers, forces every repeated application of a network to be a separate pass, and
cannot run a backward over part of a batch.

  vgg_fidelity_loss    PerceptualLoss.__call__ (losses.py:22-36) + VGG19_relu.forward (losses.py:120-164): BOTH images go
                       through the frozen VGG19 as one batch of 2B (the weights are the same), the backward runs over the
                       first B images only (losses.py:117-118: no gradient into VGG; trainer.py:108: none into real_raw)
  generator_pair       G(real_raw) and G(real_exp) (trainer.py:85,112: same weights, G is updated at :118 only) as one batch of 2B
  discriminator_loss   the D passes of one optimizer step (trainer.py:90-95 three, :102-104 two) as one batch with a
                       per-image-group spectral-norm scale, the prediction heads and the relativistic-average hinge loss fused behind it

Per-sample independence makes the batching exact: InstanceNorm is per (sample, channel) and the default configuration has no
BatchNorm (config.py:27-28).  Every function here has a parity test against the per-layer autograd path and against the
reference-generated fixtures (tests/test_fused.py, tests/test_train_step.py).
"""
import ctypes as C

import torch

from . import _lib as L
from . import ops
from .ops import ACT_NONE, ACT_RELU, _dt, _p, _stream, lib


# --------------------------------------------------------------------------------------------------------------------
# VGG19 fidelity loss
# --------------------------------------------------------------------------------------------------------------------
class _VGGFidelityFn(torch.autograd.Function):
    """loss = sum_t w_t * MSE(IN(tap_t(x)), IN(tap_t(y)));  d loss / d x.  `vgg` is a losses.VGG19_relu (frozen)."""

    @staticmethod
    def forward(ctx, x, y, vgg, weights, a, b, y_taps=None):
        from .losses import VGG_TAP_IDX
        B = x.shape[0]
        need = ctx.needs_input_grad[0]
        dt = ops.get_compute_dtype()
        # [2B, H, W, Cp], the x images first -- or [B, ...] when the taps of y were computed ahead (vgg_reference_taps)
        h = ops.raw_to_nhwc([x, y] if y_taps is None else [x], dt, a, b)
        st = _stream()
        recs, taps, tap_stats = [], [], []
        pooled = None            # the 2x2 max-pool of the current activation, when the producing conv's epilogue already wrote it
        pool_idx = None          # ... and the window positions of its maxima (first B images)
        for pi, (kind, idx) in enumerate(vgg.plan):
            if kind == "pool":
                Bt, H, W, Cc = h.shape
                if pooled is not None:
                    o = pooled
                else:
                    o = torch.empty((Bt, H // 2, W // 2, Cc), dtype=h.dtype, device=h.device)
                    L.check(lib().uegan_maxpool2x2_fwd(_dt(h), _p(h), _p(o), Bt, H, W, Cc, st))
                # backward: through the window positions of the maxima + the pooled tensor when the conv's epilogue stored them, else through h
                recs.append(("pool", h if pool_idx is None else None, (o, pool_idx, (Bt, H, W, Cc)), None, False))
                h, pooled, pool_idx = o, None, None
            else:
                conv = vgg.features[str(idx)]
                next_is_pool = pi + 1 < len(vgg.plan) and vgg.plan[pi + 1][0] == "pool" and h.shape[1] % 2 == 0 and h.shape[2] % 2 == 0
                if next_is_pool:
                    # a conv in front of a pool is never a tap (losses.py:74-104): nothing but the pool reads its output in the forward, and the
                    # backward sweep routes the pool's gradient through one byte per pooled element (the position of the maximum, first B
                    # images) + the pooled tensor -- so its full-resolution output is not stored for ANY image (uegan_conv2d_fwd_pool_idx)
                    if idx in VGG_TAP_IDX:
                        o, d, ihwo, pooled = ops.raw_conv_fwd(h, None, conv.weight, conv.bias, conv.cfg, pool=True)
                    elif need:
                        o, d, ihwo, pooled, pool_idx = ops.raw_conv_fwd(h, None, conv.weight, conv.bias, conv.cfg, pool=True, n_full=0, n_idx=B)
                    else:
                        o, d, ihwo, pooled = ops.raw_conv_fwd(h, None, conv.weight, conv.bias, conv.cfg, pool=True, n_full=0)
                    holder = None
                else:
                    # (a tap's InstanceNorm moments ride along in the conv's epilogue where the streaming kernel takes it: conv1_1, the 1-GB tap)
                    holder = ops.StatsHolder() if (idx in VGG_TAP_IDX and y_taps is None) else None
                    o, d, ihwo = ops.raw_conv_fwd(h, None, conv.weight, conv.bias, conv.cfg, stats=holder)
                is_tap = idx in VGG_TAP_IDX
                recs.append(("conv", h, d, ihwo, is_tap))
                h = o
                if is_tap:
                    taps.append(o)
                    tap_stats.append(None if holder is None else holder.value)
        loss = ops.zero_(torch.empty((1,), dtype=torch.float32, device=x.device))
        tmps = []
        for i, (w, t) in enumerate(zip(weights, taps)):
            Bt, H, W, Cc = t.shape
            ty = t[B:] if y_taps is None else y_taps[i]
            if y_taps is not None and (ty.shape != t.shape or ty.dtype != t.dtype):
                raise RuntimeError("vgg_fidelity_loss: the precomputed taps do not match this batch")
            tmp = torch.empty((3 * lib().uegan_reduce_workspace_floats(B, H * W, Cc),), dtype=torch.float32, device=x.device)
            ts = tap_stats[i]
            if ts is not None and y_taps is None:
                # [2, 2B, C] (mean, rstd) of the batch-concatenated tap: the x images first
                L.check(lib().uegan_percep_tap_fwd_given(_dt(t), _p(t), _p(ty), float(w), _p(loss), _p(tmp), B, H * W, Cc, _p(ts[0, :B]), _p(ts[1, :B]),
                                                         _p(ts[0, B:]), _p(ts[1, B:]), st))
            else:
                L.check(lib().uegan_percep_tap_fwd(_dt(t), _p(t), _p(ty), float(w), _p(loss), _p(tmp), B, H * W, Cc, ops.IN_EPS, st))
            tmps.append(tmp)
        if need:
            # per layer: its input activation, and for a tap layer its output; plain references (nothing here is an autograd input)
            ctx.recs, ctx.taps, ctx.tmps, ctx.weights, ctx.a, ctx.B, ctx.C = recs, taps, tmps, weights, a, B, x.shape[1]
            ctx.ytaps = y_taps
            ctx.last = h
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        B = ctx.B
        g = g.contiguous().float().reshape(1)
        st = _stream()
        cur = None               # d loss / d (current activation), first B images, ALREADY multiplied by relu'(activation)
        ti = len(ctx.taps) - 1
        for li in range(len(ctx.recs) - 1, -1, -1):
            kind, xin, d, ihwo, is_tap = ctx.recs[li]
            if kind == "pool":
                yp, pidx, (Bt, H, W, Cc) = d
                gx = torch.empty((B, H, W, Cc), dtype=yp.dtype, device=yp.device)
                # the pool's input is a ReLU output whose act' was deferred to its consumers: applied here
                if pidx is not None:
                    L.check(lib().uegan_maxpool2x2_bwd_idx(_dt(yp), ACT_RELU, _p(yp), _p(pidx), _p(cur), _p(gx), B, H, W, Cc, st))
                else:
                    L.check(lib().uegan_maxpool2x2_bwd_act(_dt(xin), ACT_RELU, _p(xin), _p(cur), _p(gx), B, H, W, Cc, st))
                cur = gx
                continue
            if is_tap:
                t, tmp, w = ctx.taps[ti], ctx.tmps[ti], ctx.weights[ti]
                ti -= 1
                _, H, W, Cc = t.shape
                acc = 1
                if cur is None:
                    cur = torch.empty((B, H, W, Cc), dtype=t.dtype, device=t.device)
                    acc = 0
                ty = t[B:] if ctx.ytaps is None else ctx.ytaps[ti + 1]
                L.check(lib().uegan_percep_tap_bwd_acc(_dt(t), ACT_RELU, _p(t), _p(ty), float(w), _p(g), _p(cur), _p(tmp), B, H * W, Cc,
                                                       ops.IN_EPS, acc, st))
            if cur is None:
                continue         # layers behind the last tap contribute nothing (none exist: the plan ends at relu5_1)
            if li == 0:
                dx, _ = ops.raw_conv_dgrad(d, cur, ihwo, nb=B)        # the image itself: no activation in front
                return ops.raw_to_nchw_grad(dx, ctx.C, ctx.a), None, None, None, None, None, None
            prev_is_conv = ctx.recs[li - 1][0] == "conv"
            cur, _ = ops.raw_conv_dgrad(d, cur, ihwo, in_act=ACT_RELU if prev_is_conv else ACT_NONE, x_act=xin if prev_is_conv else None, nb=B)
        raise RuntimeError("VGG plan does not start with a convolution")


def vgg_fidelity_loss(vgg, weights, x, y, a, b, y_taps=None):
    """x, y: [B,3,H,W] fp32 NCHW; (x*a + b) per channel is the ImageNet normalisation (and the (img+1)/2 rescale) folded into
    the layout conversion.  Gradient flows to x only.  y_taps: vgg_reference_taps(vgg, y, a, b) computed ahead (y is then not read)."""
    if not vgg.deferred_act_grad:
        raise RuntimeError("the fused fidelity loss applies every ReLU gradient at the consumers (VGG19_relu(deferred_act_grad=True))")
    return _VGGFidelityFn.apply(x, y, vgg, tuple(weights), a, b, None if y_taps is None else tuple(y_taps))


@torch.no_grad()
def vgg_reference_taps(vgg, y, a, b):
    """The five taps of the image the fidelity loss compares AGAINST (losses.py:29-30: no gradient reaches it): real_raw does not depend on the
    generator, so the trainer runs this pass at the very start of the step, on its second stream, beside the generator's forward -- an
    MFMA-bound pass beside an HBM-bound one -- and hands the taps to vgg_fidelity_loss(..., y_taps=...)."""
    from .losses import VGG_TAP_IDX
    h = ops.raw_to_nhwc([y], ops.get_compute_dtype(), a, b)
    st = _stream()
    taps, pooled = [], None
    for pi, (kind, idx) in enumerate(vgg.plan):
        if kind == "pool":
            Bt, H, W, Cc = h.shape
            if pooled is None:
                pooled = torch.empty((Bt, H // 2, W // 2, Cc), dtype=h.dtype, device=h.device)
                L.check(lib().uegan_maxpool2x2_fwd(_dt(h), _p(h), _p(pooled), Bt, H, W, Cc, st))
            h, pooled = pooled, None
        else:
            conv = vgg.features[str(idx)]
            next_is_pool = pi + 1 < len(vgg.plan) and vgg.plan[pi + 1][0] == "pool" and h.shape[1] % 2 == 0 and h.shape[2] % 2 == 0
            if next_is_pool:
                h, _, _, pooled = ops.raw_conv_fwd(h, None, conv.weight, conv.bias, conv.cfg, pool=True, n_full=0 if idx not in VGG_TAP_IDX else None)
            else:
                h, _, _ = ops.raw_conv_fwd(h, None, conv.weight, conv.bias, conv.cfg)
            if idx in VGG_TAP_IDX:
                taps.append(h)
    return taps


# --------------------------------------------------------------------------------------------------------------------
# Discriminator: all passes of one optimizer step as one batch, heads and relativistic-average hinge loss fused behind it
# --------------------------------------------------------------------------------------------------------------------
def _d_layers(D):
    """[(trunk SpectralNormConv2d, head Conv2d)] x 5 (models.py:139-155)"""
    return [(getattr(D, "d%d" % i)[0][1], getattr(D, "d%d_pred" % i)[0][1]) for i in range(1, 6)]


def _sn_rounds(trunks, n_rounds, do_iter, keep_uv):
    """Power-iteration rounds for the five spectral-normalised layers in one call (uegan_specnorm_multi): round r is the r-th
    application of D within this pass, in the order the reference calls it (u / v advance once per training-mode forward,
    models.py:185-188).  Returns per layer (sigma [R], inv_sigma [R], u_hist [R, rows] | None, v_hist [R, cols] | None)."""
    arr = (L.SnLayer * len(trunks))()
    out, keep = [], []
    for i, m in enumerate(trunks):
        w = m.weight_orig.detach()
        rows, cols = w.shape[0], w[0].numel()
        dev = w.device
        sig = torch.empty((2, n_rounds), dtype=torch.float32, device=dev)
        uh = torch.empty((n_rounds, rows), dtype=torch.float32, device=dev) if keep_uv else None
        vh = torch.empty((n_rounds, cols), dtype=torch.float32, device=dev) if keep_uv else None
        tmp = torch.empty((lib().uegan_specnorm_multi_workspace_floats(rows, cols),), dtype=torch.float32, device=dev)
        ops._chk(w, m.weight_u, m.weight_v)
        arr[i].w, arr[i].u, arr[i].v = _p(w), _p(m.weight_u), _p(m.weight_v)
        arr[i].sigma, arr[i].inv_sigma = _p(sig[0]), _p(sig[1])
        arr[i].u_hist, arr[i].v_hist, arr[i].tmp = _p(uh), _p(vh), _p(tmp)
        arr[i].rows, arr[i].cols = rows, cols
        out.append((sig[0], sig[1], uh, vh))
        keep.append(tmp)
    L.check(lib().uegan_specnorm_multi(arr, len(trunks), n_rounds, 1 if do_iter else 0, ops.SN_EPS, _stream()))
    return out


def _dgrad_padded(d, dz, ihwo, ohwi, scale_ptr):
    """uegan_conv2d_dgrad_padded: (gradient on the padded grid [B, H + 2p, W + 2p, C1], p) where one launch computes it, else (None, 0)"""
    nbytes = lib().uegan_conv2d_dgrad_padded_bytes(C.byref(d))
    if not nbytes or d.pad == 0:
        return None, 0
    ws = torch.empty((d.B, d.H + 2 * d.pad, d.W + 2 * d.pad, d.C1), dtype=dz.dtype, device=dz.device)
    pad = C.c_int(-1)
    L.check(lib().uegan_conv2d_dgrad_padded(C.byref(d), _p(dz), _p(ihwo), _p(ohwi), scale_ptr, _p(ws), nbytes, C.byref(pad), _stream()))
    if pad.value <= 0:
        return None, 0
    return ws, pad.value


class _DiscriminatorLossFn(torch.autograd.Function):
    """sum over (real group, fake group) pairs of GANLoss('rahinge') over the five prediction scales (losses.py:348-362, 393-409),
    D applied to every image group in ONE batched pass.

    inputs: (D, pairs, for_discriminator, n_groups, *images, *parameters) -- images: n_groups NCHW fp32 batches of one shape;
    parameters: D's parameters (they are inputs so that autograd knows whether D is being trained)."""

    @staticmethod
    def forward(ctx, D, pairs, for_d, ng, xpre, sn_pre, *rest):
        imgs, params = rest[:ng], rest[ng:]
        nb = imgs[0].shape[0]
        dt = ops.get_compute_dtype()
        layers = _d_layers(D)
        trunks = [t for t, _ in layers]
        train_d = any(ctx.needs_input_grad[6 + ng + i] for i in range(len(params)))
        img_grad = [bool(ctx.needs_input_grad[6 + g]) for g in range(ng)]
        # (sn_pre: the power-iteration rounds of this pass already done by the caller -- discriminator_sn -- e.g. on another stream, ahead of time)
        sn = sn_pre if sn_pre is not None else _sn_rounds(trunks, ng, D.training, keep_uv=train_d)
        x = xpre if xpre is not None else ops.raw_to_nhwc(list(imgs), dt)      # [ng*nb, H, W, Cp]
        st = _stream()
        recs, heads = [], []
        h = x
        for (trunk, head), (sig, inv, uh, vh) in zip(layers, sn):
            desc = ops._desc(h, None, trunk.weight_orig, trunk.cfg)
            desc.scale_group = nb                                     # image b is scaled by 1/sigma of round b // nb
            ohwi, ihwo = trunk.cfg.packed.get(trunk.weight_orig, h.dtype, desc.C1, desc.Cout)
            y = torch.empty((desc.B, desc.Ho, desc.Wo, desc.Cout), dtype=h.dtype, device=h.device)
            L.check(lib().uegan_conv2d_fwd(C.byref(desc), _p(h), None, _p(ohwi), _p(trunk.bias.detach()), _p(inv), _p(y), st))
            p, hdesc, hihwo = ops.raw_conv_fwd(y, None, head.weight, None, head.cfg)       # tanh fused
            recs.append((h, desc, ihwo, y, hdesc, hihwo, trunk.cfg.packed.version, head.cfg.packed.version))
            heads.append(p)
            h = y
        ns = len(heads)
        loss = torch.empty((1,), dtype=torch.float32, device=x.device)
        tmp = torch.empty((lib().uegan_rahinge_heads_workspace_floats(ns),), dtype=torch.float32, device=x.device)
        pix = (C.c_int64 * ns)(*[p.shape[1] * p.shape[2] for p in heads])
        cp = heads[0].shape[3]
        pr = (C.c_int32 * (2 * len(pairs)))(*[v for pq in pairs for v in pq])
        tab = (C.c_void_p * ns)(*[_p(p) for p in heads])
        L.check(lib().uegan_rahinge_heads_fwd(_dt(heads[0]), ns, tab, pix, nb, cp, ng, len(pairs), pr, 1 if for_d else 0, _p(loss), _p(tmp), st))
        if train_d or any(img_grad):
            ctx.state = (layers, sn, recs, heads, tmp, pix, pr, nb, ng, cp, bool(for_d), len(pairs), train_d, img_grad, imgs[0].shape[1], params)
        return loss

    @staticmethod
    def backward(ctx, g):
        layers, sn, recs, heads, tmp, pix, pr, nb, ng, cp, for_d, npairs, train_d, img_grad, Cimg, params = ctx.state
        g = g.contiguous().float().reshape(1)
        st = _stream()
        ns = len(heads)
        # Which image groups take part in the backward: all of them when D is being trained; otherwise only the groups whose
        # images need a gradient (the generator update, trainer.py:102-104, where D's own gradients are dead work: the real_exp
        # half of the pass has no backward at all).  Groups are contiguous batch ranges, so "some groups" = a sub-batch.
        active = list(range(ng)) if train_d else [i for i in range(ng) if img_grad[i]]
        g0, g1 = min(active), max(active) + 1
        nact, b0 = (g1 - g0) * nb, g0 * nb
        mask = 0
        for i in range(g0, g1):
            mask |= 1 << i
        gmaps = [torch.empty_like(p) for p in heads]
        tab = (C.c_void_p * ns)(*[_p(p) for p in heads])
        gtab = (C.c_void_p * ns)(*[_p(p) for p in gmaps])
        L.check(lib().uegan_rahinge_heads_bwd(_dt(heads[0]), ns, tab, pix, nb, cp, ng, npairs, pr, 1 if for_d else 0, _p(tmp), _p(g), gtab, mask, st))

        def sub(t):               # the active image range of a batched tensor
            return t[b0:b0 + nact]

        pgrads = {}
        cur, cur_pad = None, 0    # gradient w.r.t. the trunk activation of the current scale coming from the NEXT scale's trunk conv (cur_pad > 0: on that conv's padded grid)
        for li in range(len(layers) - 1, -1, -1):
            trunk, head = layers[li]
            d_in, desc, ihwo, y, hdesc, hihwo, tver, hver = recs[li]
            for cfgp, ver, saved in ((trunk.cfg.packed, tver, ihwo), (head.cfg.packed, hver, hihwo)):
                if cfgp.ihwo is saved and cfgp.version != ver:
                    # (the same refusal as ops._ConvFn.backward: `loss = discriminator_loss(...); optimizer.step(); loss.backward()`)
                    raise RuntimeError("discriminator_loss backward: D's weights were updated by an optimizer step after this forward (the packed "
                                       "copies saved for the data gradients have been rewritten in place); run backward() before step()")
            sig, inv, uh, vh = sn[li]
            dzp = sub(gmaps[li])                                      # pre-tanh gradient of this scale's prediction head
            hd = ops._sub_desc(hdesc, nact)
            ya = sub(y)
            # head -> trunk activation.  Where a kernel computes the gradient on the PADDED grid in one launch (head_dgrad_mfma_kernel) it is left
            # there: the activation backward below adds the mirror images while it reads it (`gh_pad` > 0) -- no fold pass, no folded copy
            gh, gh_pad = _dgrad_padded(hd, dzp, hihwo, head.cfg.packed.ohwi_for(hihwo), None)
            if gh is None:
                gh, _ = ops.raw_conv_dgrad(hd, dzp, hihwo)
                gh_pad = 0
            if train_d:
                dw, _ = ops.raw_conv_wgrad(hd, ya, None, dzp, head.weight, None)
                pgrads[id(head.weight)] = dw
            td = ops._sub_desc(desc, nact)
            dz = torch.empty_like(ya)
            if train_d:
                # dz = (gh + cur) * LeakyReLU'(y) / sigma_r per image group r, with the per-group projection coefficients of the spectral-norm
                # gradient and the bias gradient reduced in the same pass (uegan_sn_act_bwd): the weight gradient below is then ONE launch
                # over all groups and neither it nor the data gradient needs a per-group scale
                w, bias = trunk.weight_orig, trunk.bias
                ngr = g1 - g0
                snws = torch.empty((lib().uegan_sn_act_bwd_workspace_floats(ngr, td.Cout),), dtype=torch.float32, device=gh.device)
                nbx = lib().uegan_sn_act_bwd_p(_dt(gh), trunk.cfg.act, _p(gh), gh_pad, _p(cur), cur_pad, _p(ya), _p(bias.detach()), bias.numel(),
                                               _p(inv[g0:]), _p(dz), _p(snws), nb * td.Ho * td.Wo, td.Ho, td.Wo, td.Cout, ngr, st)
                if nbx <= 0:
                    L.check(nbx if nbx < 0 else -1)
                scale_ptr, sg = None, 0
            else:
                # dz = (gh + cur) * LeakyReLU'(y): the two consumers of the trunk activation summed inside the activation backward
                L.check(lib().uegan_act_bwd_p(_dt(gh), trunk.cfg.act, _p(gh), gh_pad, _p(cur), cur_pad, _p(ya), _p(dz), nact, td.Ho, td.Wo, td.Cout, st))
                scale_ptr, sg = _p(inv[g0:]), nb                          # sub-batch image b' belongs to round g0 + b' // nb
            if li > 0 or any(img_grad):
                tds = L.ConvDesc.from_buffer_copy(td)
                tds.scale_group = sg
                # ... and the trunk conv's own data gradient: on the padded grid too where conv_flat_kernel takes the layer (d3 - d5), for the
                # activation backward of the scale below; the gradient w.r.t. the images (li == 0) is wanted folded
                cur, cur_pad = _dgrad_padded(tds, dz, ihwo, None, scale_ptr) if li > 0 else (None, 0)
                if cur is None:
                    cur_pad = 0
                    cur = torch.empty((nact, td.H, td.W, td.C1), dtype=dz.dtype, device=dz.device)
                    dwsb = lib().uegan_conv2d_dgrad_workspace_bytes(C.byref(tds))
                    dws = torch.empty((dwsb + 3) // 4, dtype=torch.float32, device=dz.device) if dwsb else None
                    L.check(lib().uegan_conv2d_dgrad_ws(C.byref(tds), _p(dz), _p(ihwo), scale_ptr, _p(cur), None, _p(dws), dwsb, st))
            else:
                cur, cur_pad = None, 0
            if train_d:
                # dW (+)= wgrad(x, dz) - sum_r c_r u_r v_r^T with the u, v of round r (torch spectral_norm: constants of the call);
                # db (+)= sum dz_raw.  Straight into the optimizer bucket when there is one.
                wd = w.detach()
                rows, cols = wd.shape[0], wd[0].numel()
                wsink, bsink = ops._sink_of(w), ops._sink_of(bias)
                dw_acc = wsink.view if wsink is not None else torch.empty_like(wd)
                db_acc = bsink.view if bsink is not None else torch.empty_like(bias.detach())
                w_live = wsink is not None and wsink.dirty         # the bucket already holds a gradient of this step
                b_live = bsink is not None and bsink.dirty
                gd = L.ConvDesc.from_buffer_copy(td)
                gd.scale_group = 0
                wsb = lib().uegan_conv2d_wgrad_workspace_bytes(C.byref(gd))
                ws = torch.empty((max(wsb, 4) + 3) // 4, dtype=torch.float32, device=wd.device)
                L.check(lib().uegan_conv2d_wgrad_acc(C.byref(gd), _p(sub(d_in)), None, _p(dz), None, _p(dw_acc), None, _p(ws), wsb, 1 if w_live else 0, st))
                L.check(lib().uegan_sn_grad_finish(_p(dw_acc), _p(db_acc), _p(snws), nbx, ngr, _p(uh[g0:]), _p(vh[g0:]), rows, cols, td.Cout,
                                                   1 if b_live else 0, st))
                if wsink is not None:
                    wsink.mark()
                if bsink is not None:
                    bsink.mark()
                pgrads[id(w)] = None if wsink is not None else dw_acc
                pgrads[id(bias)] = None if bsink is not None else db_acc
        igrads = []
        for gi in range(ng):
            if img_grad[gi] and cur is not None and g0 <= gi < g1:
                igrads.append(ops.raw_to_nchw_grad(cur[(gi - g0) * nb:(gi - g0 + 1) * nb], Cimg))
            else:
                igrads.append(None)
        return (None, None, None, None, None, None) + tuple(igrads) + tuple(pgrads.get(id(p)) for p in params)


def discriminator_input(images):
    """the batch-concatenated NHWC copy of `images` that discriminator_loss works on, for callers that want to queue the conversion early
    (Trainer: before the wait for D's all-reduce) and hand it back as `x_nhwc`"""
    return ops.raw_to_nhwc(list(images), ops.get_compute_dtype())


def discriminator_sn(D, n_groups, keep_uv=True):
    """The spectral-norm power-iteration rounds of ONE batched pass over n_groups image groups (u / v advance n_groups times in training mode, in the
    order the reference applies D, models.py:185-188), for a caller that wants them queued early: they depend on D's weights only, so the Trainer runs
    the D update's rounds at the very start of the step on its second stream, beside the generator's forward, and hands them to
    discriminator_loss(..., sn=...)."""
    return _sn_rounds([t for t, _ in _d_layers(D)], n_groups, D.training, keep_uv)


def discriminator_loss(D, images, pairs, for_discriminator, x_nhwc=None, sn=None):
    """GANLoss('rahinge') summed over `pairs` = [(real group, fake group), ...] of indices into `images` (NCHW fp32 batches of one
    shape), with D applied to all of them in one batched pass; the image groups are applied in list order as far as the
    spectral-norm state is concerned (group g uses the u, v, sigma of the g-th power iteration of this call).  Returns shape [1].

        D update  (trainer.py:90-95):   discriminator_loss(D, [real_exp, fake_store, real_raw], [(0, 1), (0, 2)], True)
        G update  (trainer.py:102-104): discriminator_loss(D, [real_exp, fake_exp], [(0, 1)], False)
    """
    images = list(images)
    if not 2 <= len(images) <= 4:
        raise ValueError("discriminator_loss takes 2..4 image groups")
    for x in images:
        if x.dim() != 4 or x.shape[1] != 3 or x.shape != images[0].shape:
            raise RuntimeError("Discriminator expects [B,3,H,W] batches of one shape (got %s)" % (tuple(x.shape),))
    params = [p for p in D.parameters()]
    return _DiscriminatorLossFn.apply(D, tuple(tuple(p) for p in pairs), bool(for_discriminator), len(images), x_nhwc, sn, *images, *params)
