"""MI355X-native mirror of the reference's models.py: same class names, constructor / call signatures and
state-dict keys, every device op a hand-written gfx950 kernel from libuegan_hip.so.

    Generator(conv_dim, norm_fun, act_fun, use_sn)(x)                  -> Tensor       (models.py:10-74)
    Discriminator(conv_dim, norm_fun, act_fun, use_sn, adv_loss_type)(x) -> [5 Tensors] (models.py:104-155)

Inputs/outputs are the reference's NCHW float32 tensors; inside, activations are NHWC in the compute dtype
(ops.set_compute_dtype).  The reference's DEFAULT hyper-parameters (config.py:11-81: norm 'none', LeakyReLU(0.2), g_use_sn False,
d_use_sn True, rahinge/hinge heads) take the restructured path described below; every other value of those flags runs layer by layer
(uegan_amd/variants.py: BatchNorm / InstanceNorm, ReLU / Swish / SELU / none, spectral norm on or off in either network, sigmoid heads
for 'ls' / 'rals'); values the reference does not know raise its NotImplementedError.

nn.DataParallel (trainer.py:317-321): with one device id the wrapper is a pass-through and works; with several it would replicate the
module per forward (new parameter tensors every call: the packed-weight caches and the optimizer's gradient bucket would be bypassed) --
the replicas raise instead.  Scale with one process per GPU and uegan_amd.trainer.Trainer (RCCL all-reduce of the two gradient buckets).

Exact algebraic restructurings (each covered by a parity test against the reference-pinned oracle):
  * upsample path: conv1x1(bilinear_up(x)) is computed as bilinear_up(conv1x1(x)) -- both are linear and the bilinear
    weights sum to 1, so the bias commutes too; 4x fewer MACs and bytes (models.py:23-26).
  * GAM (models.py:230-237) with norm=True: the gate branch and the fuse bias are constant over H x W and are removed
    exactly by the following non-affine InstanceNorm, so ga(x) = IN(W_fuse[:, :C] * x).  The gate parameters
    (conv.0, conv.2, fuse bias, fuse.weight[:, C:]) still exist, keep their state-dict keys and receive exactly-zero
    gradients, so Adam's weight decay moves them as in the reference.
  * torch.cat (models.py:55,59,63,67) is virtual: the decoder convs read their two sources directly.
"""
import math


def _refuse_replica(m):
    if getattr(m, "_is_replica", False):
        raise RuntimeError("uegan_amd modules do not run as nn.DataParallel replicas (several device ids): use one process per GPU and "
                           "uegan_amd.trainer.Trainer -- see INTEGRATION.md")


import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from . import variants


def _require(cond, what):
    if not cond:
        raise NotImplementedError("uegan_amd implements the reference's default configuration only: " + what)


class ReflectionPadTag(nn.Module):
    """Index-0 placeholder of the reference's nn.Sequential(ReflectionPad2d, Conv2d, ...): the padding is fused
    into the conv kernel's tile loads (index reflection), so this module is never called."""

    def __init__(self, padding):
        super().__init__()
        self.padding = padding

    def forward(self, x):  # pragma: no cover
        raise RuntimeError("padding is fused into the convolution kernel")


class _InvalidatingModule(nn.Module):
    """nn.Module whose weights have packed device copies (ops.PackedWeight): `load_state_dict` and `apply` (the reference's
    `net.apply(init_func)`, trainer.py:390, edits `m.weight.data` in place, which no version counter sees) drop them."""

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        ops.invalidate_weight_caches()

    def apply(self, fn):
        r = super().apply(fn)
        ops.invalidate_weight_caches()
        return r


class Conv2d(_InvalidatingModule):
    """Parameter holder + launcher for one fused [reflect/zero pad -> conv -> bias -> activation] kernel.
    Class name contains 'Conv' and exposes `.weight` so trainer.py:357-390 `init_weights` reaches it."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, bias=True, act=ops.ACT_NONE, pad_mode=ops.PAD_REFLECT):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size, self.stride = in_channels, out_channels, kernel_size, stride
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.cfg = ops.ConvCfg(stride, pad_mode, act)
        self.reset_parameters()

    def reset_parameters(self):  # nn.Conv2d default init
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.in_channels * self.kernel_size * self.kernel_size
            bound = 1 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x, x2=None, n_out=1, ex=None):
        return ops.conv2d(x, x2, self.weight, self.bias, self.cfg, n_out=n_out, ex=ex)


class SpectralNormConv2d(_InvalidatingModule):
    """Conv2d under torch.nn.utils.spectral_norm semantics (models.py:185-188): parameters weight_orig / bias and
    buffers weight_u / weight_v with the reference's state-dict names; one power iteration per TRAINING forward."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, bias=True, act=ops.ACT_NONE):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size, self.stride = in_channels, out_channels, kernel_size, stride
        w = torch.empty(out_channels, in_channels, kernel_size, kernel_size)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        if bias:
            bound = 1 / math.sqrt(in_channels * kernel_size * kernel_size)
            nn.init.uniform_(self.bias, -bound, bound)
        self.weight_orig = nn.Parameter(w)
        self.register_buffer("weight_u", F.normalize(torch.randn(out_channels), dim=0, eps=ops.SN_EPS))
        self.register_buffer("weight_v", F.normalize(torch.randn(in_channels * kernel_size * kernel_size), dim=0, eps=ops.SN_EPS))
        self.cfg = ops.ConvCfg(stride, ops.PAD_REFLECT, act)

    @property
    def weight(self):          # init_weights writes m.weight.data -> lands in weight_orig
        return self.weight_orig

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight_orig, a=math.sqrt(5))

    def forward(self, x, x2=None, n_out=1, ex=None):
        sn = ops.specnorm_sigma(self.weight_orig, self.weight_u, self.weight_v, do_iter=self.training)
        return ops.conv2d(x, x2, self.weight_orig, self.bias, self.cfg, sn=sn, n_out=n_out, ex=ex)


class Identity(nn.Module):
    def forward(self, x):
        return x


def get_act_fun(act_fun_type="LeakyReLU"):
    """models.py:249-263.  A position holder: LeakyReLU / ReLU live in the conv epilogue when no norm sits between, every other
    combination is evaluated by the block's normalise+activate kernel (variants.NormAct)."""
    return variants.ActTag(act_fun_type)


def get_norm_fun(norm_fun_type="none"):
    """models.py:271-281"""
    if norm_fun_type == "BatchNorm":
        return variants.BatchNorm2d
    if norm_fun_type == "InstanceNorm":
        return variants.InstanceNorm2d
    if norm_fun_type == "none":
        return lambda c: Identity()
    raise NotImplementedError("normalization function [%s] is not found" % norm_fun_type)


def _conv_norm_act(in_channels, out_channels, kernel_size, stride, use_bias, norm_fun, act_fun, use_sn):
    """[pad tag, conv, norm, act] of models.py:88-99 / 158-167 and the normalise+activate stage behind the conv (None when the
    conv's epilogue does everything: no norm and an epilogue activation)."""
    act = get_act_fun(act_fun)
    norm = get_norm_fun(norm_fun)(out_channels)
    fused = norm_fun == "none" and act_fun in variants.EPILOGUE_ACTS
    conv_cls = SpectralNormConv2d if use_sn else Conv2d
    conv = conv_cls(in_channels, out_channels, kernel_size, stride, use_bias, act=act.code if fused else ops.ACT_NONE)
    post = None if fused else variants.NormAct(norm if isinstance(norm, variants._Norm2d) else None, act)
    return [ReflectionPadTag((kernel_size - 1) // 2), conv, norm, act], post


class ConvBlock(nn.Module):
    """models.py:88-101: ReflectionPad2d -> (SpectralNorm) Conv2d(bias) -> norm -> activation.  Default flags (norm 'none',
    LeakyReLU): one kernel; otherwise the conv kernel followed by one normalise+activate kernel."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation, use_bias, norm_fun, act_fun, use_sn):
        super().__init__()
        _require(dilation == 1, "dilation 1")
        self.padding = (kernel_size - 1) // 2
        mods, post = _conv_norm_act(in_channels, out_channels, kernel_size, stride, use_bias, norm_fun, act_fun, use_sn)
        self.main = nn.Sequential(*mods)
        self.post = post

    def forward(self, x, x2=None, n_out=1, ex=None):
        y = self.main[1](x, x2, n_out=n_out, ex=ex)
        if self.post is None:
            return y
        if n_out != 1 or ex is not None:
            raise RuntimeError("output aliases / convolution extras are features of the single-kernel block")
        return self.post(y)


class SNConv(nn.Module):
    """models.py:77-86: ReflectionPad2d -> (SpectralNorm) Conv2d, no activation (act set by the caller: dec5.1's tanh)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation, use_bias, use_sn, act=ops.ACT_NONE):
        super().__init__()
        _require(dilation == 1, "dilation 1")
        self.padding = (kernel_size - 1) // 2
        conv_cls = SpectralNormConv2d if use_sn else Conv2d
        self.main = nn.Sequential(ReflectionPadTag(self.padding), conv_cls(in_channels, out_channels, kernel_size, stride, use_bias, act=act))

    def forward(self, x, ex=None):
        return self.main[1](x, ex=ex)


class Interpolate(nn.Module):
    """models.py:191-201 (scale 2, bilinear, align_corners=True)."""

    def __init__(self, scale_factor, mode, align_corners):
        super().__init__()
        _require(scale_factor == 2 and mode == "bilinear" and align_corners, "Interpolate(2,'bilinear',True)")

    def forward(self, x):
        return ops.upsample2x(x)


class _TouchParams(torch.autograd.Function):
    """Identity on y that makes forward-dead parameters part of the graph with exactly-zero gradients."""

    @staticmethod
    def forward(ctx, y, *params):
        ctx.meta = [(p.shape, p.dtype, p.device) for p in params]
        ctx.sinks = [ops._sink_of(p) for p in params]
        ctx.sunk = [sk is not None for sk in ctx.sinks]
        return y.view_as(y)

    @staticmethod
    def backward(ctx, g):
        # (a parameter whose gradient lives in an optimizer bucket needs nothing: the bucket is zeroed by zero_grad and "+= 0" is a
        # no-op; its sink is marked so that the bucket knows the parameter is done)
        for sk in ctx.sinks:
            if sk is not None:
                sk.mark()
        return (g,) + tuple(None if sunk else torch.zeros(s, dtype=d, device=dev) for (s, d, dev), sunk in zip(ctx.meta, ctx.sunk))


class GAM(nn.Module):
    """Global attention module (models.py:215-237) with norm=True; see the module docstring for the exact shortcut."""

    def __init__(self, in_nc, out_nc, reduction=8, bias=False, use_sn=False, norm=False):
        super().__init__()
        _require(norm and not bias and in_nc == out_nc, "GAM(norm=True, bias=False)")
        self.conv = nn.Sequential(Conv2d(in_nc * 2, in_nc // reduction, 1, 1, bias=False), Identity(),
                                  Conv2d(in_nc // reduction, out_nc, 1, 1, bias=False))
        # (g_use_sn: sigma is the spectral norm of the FULL [out_nc, 2*in_nc] matrix, models.py:223 -- the forward-dead columns still
        # enter sigma, and the spectral-norm correction of the gradient reaches them)
        self.fuse = nn.Sequential((SpectralNormConv2d if use_sn else Conv2d)(in_nc * 2, out_nc, 1, 1, bias=True))
        self.use_sn = use_sn
        self.in_nc = in_nc
        self.norm = norm
        # the fuse conv restricted to its first in_nc input channels: packed from, and its weight gradient written into, that
        # column slice of the full [out_nc, 2*in_nc, 1, 1] parameter (no sliced copy, no scatter-add of the gradient)
        self._cfg = ops.ConvCfg(1, ops.PAD_REFLECT, ops.ACT_NONE, cin_used=in_nc)

    def forward(self, x, x_lo=None):
        """x_lo: x is a hi + lo pair (ops.set_precise; the full-resolution module ga1): weights as a pair, the conv's result and the normalised
        output as pairs -> returns (y, y_lo)"""
        fuse = self.fuse[0]
        sn = ops.specnorm_sigma(fuse.weight_orig, fuse.weight_u, fuse.weight_v, do_iter=self.training) if self.use_sn else None
        # (the moments of the InstanceNorm ride along in the conv's epilogue where the streaming kernel takes the layer -- ga1, ga2 -- and the norm is
        # then ONE pass over y instead of two)
        holder = ops.StatsHolder()
        if x_lo is not None:
            ex = ops.ConvExtras(x1_lo=x_lo, pair_w=True, want_lo=True)
            y = ops.conv2d(x, None, fuse.weight, None, self._cfg, sn=sn, stats=holder, ex=ex)
            if holder.value is None:
                raise RuntimeError("GAM on a hi + lo pair: the convolution did not deliver the InstanceNorm moments")
            y, y_lo = ops.instnorm_pair(y, ex.y_lo, holder.value)
        else:
            y = ops.conv2d(x, None, fuse.weight, None, self._cfg, sn=sn, stats=holder)
            y = ops.instnorm(y, holder.value)
        if torch.is_grad_enabled():
            dead = [p for p in (self.conv[0].weight, self.conv[2].weight, fuse.bias) if p.requires_grad]
            if dead:
                y = _TouchParams.apply(y, *dead)
        return y if x_lo is None else (y, y_lo)


class Generator(_InvalidatingModule):
    """Generator network (models.py:10-74)."""

    def __init__(self, conv_dim, norm_fun, act_fun, use_sn):
        super().__init__()
        # the default flags (config.py:23-27) take the restructured single-kernel blocks, aliases and deferred activation gradients
        # below; any other combination runs the same dataflow layer by layer (_body_plain)
        self.default_flags = norm_fun == "none" and act_fun == "LeakyReLU" and not use_sn
        cd = conv_dim
        kw = dict(padding=0, dilation=1, use_bias=True, norm_fun=norm_fun, act_fun=act_fun, use_sn=use_sn)
        self.enc1 = ConvBlock(3, cd, 7, 1, **kw)
        self.enc2 = ConvBlock(cd, cd * 2, 3, 2, **kw)
        self.enc3 = ConvBlock(cd * 2, cd * 4, 3, 2, **kw)
        self.enc4 = ConvBlock(cd * 4, cd * 8, 3, 2, **kw)
        self.enc5 = ConvBlock(cd * 8, cd * 16, 3, 2, **kw)

        self.upsample1 = nn.Sequential(Interpolate(2, "bilinear", True), SNConv(cd * 16, cd * 8, 1, 1, 0, 1, True, use_sn))
        self.upsample2 = nn.Sequential(Interpolate(2, "bilinear", True), SNConv(cd * 8, cd * 4, 1, 1, 0, 1, True, use_sn))
        self.upsample3 = nn.Sequential(Interpolate(2, "bilinear", True), SNConv(cd * 4, cd * 2, 1, 1, 0, 1, True, use_sn))
        self.upsample4 = nn.Sequential(Interpolate(2, "bilinear", True), SNConv(cd * 2, cd * 1, 1, 1, 0, 1, True, use_sn))

        self.dec1 = ConvBlock(cd * 16, cd * 8, 3, 1, **kw)
        self.dec2 = ConvBlock(cd * 8, cd * 4, 3, 1, **kw)
        self.dec3 = ConvBlock(cd * 4, cd * 2, 3, 1, **kw)
        self.dec4 = ConvBlock(cd * 2, cd * 1, 3, 1, **kw)
        self.dec5 = nn.Sequential(SNConv(cd, cd, 3, 1, 0, 1, True, False), SNConv(cd, 3, 7, 1, 0, 1, True, False, act=ops.ACT_TANH),
                                  Identity())   # index 2 was nn.Tanh(): fused into dec5.1's epilogue

        self.ga5 = GAM(cd * 16, cd * 16, reduction=8, bias=False, use_sn=use_sn, norm=True)
        self.ga4 = GAM(cd * 8, cd * 8, reduction=8, bias=False, use_sn=use_sn, norm=True)
        self.ga3 = GAM(cd * 4, cd * 4, reduction=8, bias=False, use_sn=use_sn, norm=True)
        self.ga2 = GAM(cd * 2, cd * 2, reduction=8, bias=False, use_sn=use_sn, norm=True)
        self.ga1 = GAM(cd * 1, cd * 1, reduction=8, bias=False, use_sn=use_sn, norm=True)
        # Deferred activation gradients (ops.ConvCfg; an exact restructuring, the same the VGG chain uses): a layer whose output has
        # ONE consumer skips its own activation-backward pass, the consumer's backward multiplies by act'(that output) where it
        # reads it anyway -- the 1x1 upsample / attention convs' data-gradient epilogues, y4.mul(x1)'s backward, the final clamp's.
        self.dec5[1].main[1].cfg.premasked = True       # consumer: residual_clamp (tanh'); dec5 is the same under every flag
        if not self.default_flags:
            return
        L = ops.ACT_LRELU
        for prod, cons in ((self.enc5, self.ga5), (self.dec1, self.upsample2[1]), (self.dec2, self.upsample3[1]), (self.dec3, self.upsample4[1])):
            prod.main[1].cfg.premasked = True
            (cons._cfg if isinstance(cons, GAM) else cons.main[1].cfg).in_act = L
        self.dec4.main[1].cfg.premasked = True          # consumer: mul (below)

    @staticmethod
    def _up(block, x, ex=None):
        # reference: conv1x1(bilinear_up(x)); here bilinear_up(conv1x1(x)) (exact, see module docstring)
        return block[0](block[1](x, ex=ex))

    @staticmethod
    def _check_input(x):
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] % 16 or x.shape[3] % 16 or min(x.shape[2:]) < 32:
            raise RuntimeError("Generator expects [B,3,H,W] with H,W multiples of 16 and >= 32 (got %s)" % (tuple(x.shape),))

    def _precise(self):
        """ops.set_precise in force for this network?  The pair kernels of uegan_conv2d_fwd_ex exist for the default flags at conv_dim 32 (the reference's
        configuration, config.py:23-27): any other generator refuses the mode instead of silently running the plain one."""
        if not ops.precise():
            return False
        if not (self.default_flags and self.enc1.main[1].out_channels == 32):
            raise RuntimeError("uegan_amd.set_precise(True) covers the default generator (norm 'none', LeakyReLU, no spectral norm) at conv_dim 32; "
                               "call set_precise(False) for this network")
        return True

    def forward(self, x):
        _refuse_replica(self)
        self._check_input(x)
        res, outs = self._body(ops.to_nhwc(x, pair=self._precise()), (x,))
        return ops.residual_clamp(res, x, ops.ACT_TANH, given=outs)     # clamp(res + x, -1, 1), NCHW fp32 (outs: dec5.1's epilogue already wrote it)

    def forward_pair(self, xa, xb, xin=None):
        """(G(xa), G(xb)) as ONE pass over the batch-concatenated images: the two generator calls of a training step
        (trainer.py:85 and :112) see the same weights, and every op is per-sample (InstanceNorm included), so this is exact;
        it halves the launches and gives the small-map layers grids that fill the chip."""
        self._check_input(xa)
        self._check_input(xb)
        if not self.default_flags:
            raise RuntimeError("forward_pair batches two generator calls: exact only without batch statistics (default flags)")
        if xa.shape[1:] != xb.shape[1:]:
            raise RuntimeError("forward_pair: both image sets must have one image shape")
        # (xin: self.input_pair(xa, xb) when the caller has already queued the conversion -- Trainer.train_step does, ahead of the
        # previous step's pending optimizer update)
        res, outs = self._body(xin if xin is not None else self.input_pair(xa, xb), (xa, xb))
        return ops.residual_clamp_pair(res, xa, xb, ops.ACT_TANH, given=outs)

    def input_pair(self, xa, xb):
        """the batch-concatenated NHWC copy of two image sets that forward_pair works on (for callers that queue the conversion early)"""
        return ops.to_nhwc_pair(xa, xb, pair=self._precise())

    def _body_plain(self, xin):
        """models.py:46-71 layer by layer (non-default norm / activation / spectral-norm flags): no output aliases, no deferred
        activation gradients -- autograd sums the gradients of the multi-consumer encoder activations"""
        x1 = self.enc1(xin)
        x2 = self.enc2(x1)
        x3 = self.enc3(x2)
        x4 = self.enc4(x3)
        x5 = self.ga5(self.enc5(x4))
        y1 = self.dec1(self._up(self.upsample1, x5), self.ga4(x4))
        y2 = self.dec2(self._up(self.upsample2, y1), self.ga3(x3))
        y3 = self.dec3(self._up(self.upsample3, y2), self.ga2(x2))
        y4 = self.dec4(self._up(self.upsample4, y3), self.ga1(x1))
        return self.dec5[1](self.dec5[0](ops.mul(y4, x1)))

    def _body(self, xin, xs):
        """models.py:46-71 on an NHWC (channel-padded) image batch -> (the tanh residual `res` (NHWC, channel-padded), outs).  xs: the NCHW fp32 image
        set(s) xin was made from; outs: clamp(res + x, -1, 1) per set when dec5.1's epilogue wrote it (16-bit storage), else None.
        ops.set_precise: the full-resolution chain image -> x1 -> ga1 -> y4 * x1 -> dec5.0 -> dec5.1 runs on hi + lo pairs (uegan_conv2d_fwd_ex)."""
        if not self.default_flags:
            return self._body_plain(xin), None
        P = self._precise()
        X = ops.ConvExtras
        # encoder activations with several consumers (next encoder stage, attention module, final modulation) come back as one
        # alias per consumer: their gradients meet inside the producing conv's activation-backward kernel (ops._ConvFn)
        ex1 = X(pair_w=True, dup_cin=True, want_lo=True) if P else None
        x1a, x1b, x1c = self.enc1(xin, n_out=3, ex=ex1)
        x1_lo = ex1.y_lo if P else None
        # (precise: enc2's and upsample4's WEIGHTS as pairs too -- the two deep layers whose weight rounding, a systematic perturbation, carried the tail of
        # the pixel error in tools/diag_g_hilo.py; their sources and results stay plain)
        x2a, x2b = self.enc2(x1a, n_out=2, ex=X(pair_w=True) if P else None)
        x3a, x3b = self.enc3(x2a, n_out=2)
        x4a, x4b = self.enc4(x3a, n_out=2)
        x5 = self.enc5(x4a)
        x5 = self.ga5(x5)

        y1 = self.dec1(self._up(self.upsample1, x5), self.ga4(x4b))
        y2 = self.dec2(self._up(self.upsample2, y1), self.ga3(x3b))
        y3 = self.dec3(self._up(self.upsample3, y2), self.ga2(x2b))
        # y4.mul(x1) (models.py:69) is formed by dec4's epilogue from its fp32 result where a kernel does that (16-bit storage); `mul` then only records
        # the backward.  clamp(tanh(dec5.1) + x) likewise by dec5.1's epilogue (outs).
        if P:
            g1, g1_lo = self.ga1(x1b, x_lo=x1_lo)
            ex4 = X(x2_lo=g1_lo, pair_w=True, mul=x1c, mul_lo=x1_lo, want_mul_lo=True)
        else:
            g1 = self.ga1(x1b)
            ex4 = X(mul=x1c) if (xin.dtype != torch.float32 and ops.fuse_epilogues[0]) else None
        y4 = self.dec4(self._up(self.upsample4, y3, ex=X(pair_w=True) if P else None), g1, ex=ex4)
        prod = ops.mul(y4, x1c, act_a=ops.ACT_LRELU, given=ex4.prod if ex4 is not None else None)      # y4's LeakyReLU' applied in mul's backward
        ex5 = X(x1_lo=ex4.prod_lo, pair_w=True, want_lo=True) if P else None
        d50 = self.dec5[0](prod, ex=ex5)
        ex6 = X(x1_lo=ex5.y_lo, pair_w=True, res=xs) if P else (X(res=xs) if (xin.dtype != torch.float32 and ops.fuse_epilogues[0]) else None)
        res = self.dec5[1](d50, ex=ex6)                                                                # tanh fused in dec5.1
        return res, (ex6.res_out if ex6 is not None else None)


class _DisBlock(nn.Sequential):
    """Sequential [pad tag, conv, norm, act] (the reference's module indices) that runs conv -> normalise+activate"""

    def __init__(self, mods, post):
        super().__init__(*mods)
        object.__setattr__(self, "_post", post)

    def forward(self, x):
        y = self[1](x)
        return y if self._post is None else self._post(y)


def dis_conv_block(in_channels, out_channels, kernel_size, stride, padding, dilation, use_bias, norm_fun, act_fun, use_sn):
    """models.py:158-167."""
    _require(dilation == 1, "dilation 1")
    mods, post = _conv_norm_act(in_channels, out_channels, kernel_size, stride, use_bias, norm_fun, act_fun, use_sn)
    return _DisBlock(mods, post)


def dis_pred_conv_block(in_channels, out_channels, kernel_size, stride, padding, dilation, use_bias, type):
    """models.py:170-182: prediction head, no spectral norm; sigmoid for ls / rals, tanh for (ra)hinge -- in the conv's epilogue."""
    if type in ("ls", "rals"):
        act = variants.ACT_SIGMOID
    elif type in ("hinge", "rahinge"):
        act = ops.ACT_TANH
    else:
        raise NotImplementedError("Adversarial loss [{}] is not found".format(type))
    _require(dilation == 1, "dilation 1")
    pad = (kernel_size - 1) // 2
    return nn.Sequential(ReflectionPadTag(pad), Conv2d(in_channels, out_channels, kernel_size, stride, use_bias, act=act), Identity())


class Discriminator(_InvalidatingModule):
    """Multi-scale discriminator (models.py:104-155): returns the 5 prediction maps [B,1,H/2^k,W/2^k] (NCHW fp32)."""

    def __init__(self, conv_dim, norm_fun, act_fun, use_sn, adv_loss_type):
        super().__init__()
        cd = conv_dim
        cin = 3
        # (uegan_amd/fused.py batches the passes of a step and fuses the 'rahinge' loss behind them: default flags only)
        self.default_flags = norm_fun == "none" and act_fun == "LeakyReLU" and use_sn and adv_loss_type in ("rahinge", "hinge")
        for i, (k, m) in enumerate(zip((7, 7, 7, 5, 5), (1, 2, 4, 8, 16))):
            cout = cd * m
            setattr(self, "d%d" % (i + 1), nn.Sequential(dis_conv_block(cin, cout, k, 2, 0, 1, True, norm_fun, act_fun, use_sn)))
            setattr(self, "d%d_pred" % (i + 1), nn.Sequential(dis_pred_conv_block(cout, 1, k, 1, 0, 1, False, adv_loss_type)))
            cin = cout

    def forward(self, x):
        _refuse_replica(self)
        if x.dim() != 4 or x.shape[1] != 3:
            raise RuntimeError("Discriminator expects [B,3,H,W] (got %s)" % (tuple(x.shape),))
        h = ops.to_nhwc(x)
        preds = []
        for i in range(1, 6):
            h = getattr(self, "d%d" % i)[0](h)
            preds.append(ops.to_nchw(getattr(self, "d%d_pred" % i)[0][1](h), 1))      # head tensor is channel-padded; keep channel 0
        return preds
