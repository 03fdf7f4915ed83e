"""MI355X-native mirror of the reference's losses.py (the parts on the hot path).

    PerceptualLoss()(x, y)                                   -> 0-dim Tensor   (losses.py:12-36; VGG19_relu 39-164)
    GANLoss('rahinge', tensor=...)(real_preds, fake_preds, None, None, for_discriminator=...) -> Tensor [1]  (255-411)
    MultiscaleRecLoss(3, 'l1', True)(input, target)          -> 0-dim Tensor   (losses.py:202-231)
    TVLoss                                                   name only (tester.py:9 imports it, never calls it)

VGG19 weights: the reference downloads torchvision's pretrained `vgg19-dcbb9e9d.pth` (losses.py:43-44) and fails if it
cannot.  `PerceptualLoss` likewise REQUIRES that file (argument, $UEGAN_VGG19_WEIGHTS or ./models/vgg19-dcbb9e9d.pth,
torchvision keys `features.N.weight|bias`) and raises when it is missing.  Benchmarks and tests, which have no network, opt in
explicitly to a documented seeded stand-in of the same architecture with `PerceptualLoss(vgg_weights="seeded")`
("parity unpinned" w.r.t. the pretrained network).
"""
import math
import os

import torch
import torch.nn as nn

from . import fused, ops, variants
from .models import Conv2d

VGG_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
VGG_CONV_IDX = [0, 2, 5, 7, 10, 12, 14, 16, 19, 21, 23, 25, 28, 30, 32, 34]   # torchvision `features` indices (cfg "E")
VGG_TAP_IDX = (0, 5, 10, 19, 28)                                             # relu1_1, 2_1, 3_1, 4_1, 5_1 (losses.py:30-34)
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def seeded_vgg19_weights(seed=1234, width_div=1):
    """Stand-in for the pretrained weights: per conv `torch.Generator().manual_seed(seed + features_idx)`,
    weight = randn * sqrt(2 / fan_in), bias = randn * 0.05 (drawn from the same generator, after the weight)."""
    W = {}
    c = 3
    ci = 0
    for v in VGG_CFG:
        if v == "M":
            continue
        v = max(v // width_div, 1)
        idx = VGG_CONV_IDX[ci]
        g = torch.Generator().manual_seed(seed + idx)
        W["features.%d.weight" % idx] = torch.randn(v, c, 3, 3, generator=g) * math.sqrt(2.0 / (c * 9))
        W["features.%d.bias" % idx] = torch.randn(v, generator=g) * 0.05
        c = v
        ci += 1
    return W


class VGG19_relu(nn.Module):
    """torchvision VGG19 `features` through relu5_1 (the last tap PerceptualLoss consumes): 13 x [conv3x3 zero-pad 1 +
    bias + ReLU] with MaxPool2d(2,2) at features idx 4, 9, 18, 27.  The reference also evaluates conv5_2..5_4 and
    discards them (losses.py:137-140); they cannot influence any output and are not computed.  Parameters are
    frozen (losses.py:117-118) and keep torchvision's names `features.N.weight|bias`."""

    def __init__(self, state_dict=None, width_div=1, deferred_act_grad=False):
        """deferred_act_grad: no ReLU-backward passes -- each conv's relu'(y) is applied by the consumers of y instead (the
        next conv's dgrad epilogue, the max-pool backward, the fidelity-loss gradient; ops.ConvCfg).  The caller must then
        consume the taps ONLY through ops.perceptual_taps_loss(..., in_act=ACT_RELU) (PerceptualLoss does)."""
        super().__init__()
        self.deferred_act_grad = deferred_act_grad
        self.features = nn.ModuleDict()
        self.plan = []          # ("conv", idx) / ("pool",)
        c = 3
        ci = 0
        layer = 0
        for v in VGG_CFG:
            if v == "M":
                self.plan.append(("pool", None))
                layer += 1
                continue
            v = max(v // width_div, 1)
            idx = VGG_CONV_IDX[ci]
            assert idx == layer
            self.features[str(idx)] = Conv2d(c, v, 3, 1, bias=True, act=ops.ACT_RELU, pad_mode=ops.PAD_ZERO)
            if deferred_act_grad:
                cfg = self.features[str(idx)].cfg
                cfg.premasked = True
                cfg.in_act = ops.ACT_RELU if (self.plan and self.plan[-1][0] == "conv") else ops.ACT_NONE
            self.plan.append(("conv", idx))
            c = v
            ci += 1
            layer += 2
            if idx == VGG_TAP_IDX[-1]:
                break
        if state_dict is not None:
            own = self.state_dict()
            self.load_state_dict({k: state_dict[k] for k in own})
        for p in self.parameters():
            p.requires_grad = False

    def forward(self, x_nhwc):
        taps = []
        h = x_nhwc
        for kind, idx in self.plan:
            if kind == "pool":
                h = ops.maxpool2x2(h, ops.ACT_RELU if self.deferred_act_grad else ops.ACT_NONE)
            else:
                h = self.features[str(idx)](h)
                if idx in VGG_TAP_IDX:
                    taps.append(h)
        return taps


def _find_vgg_weights(path):
    cands = [path, os.environ.get("UEGAN_VGG19_WEIGHTS"), os.path.join(".", "models", "vgg19-dcbb9e9d.pth")]
    for c in cands:
        if c and os.path.exists(c):
            return torch.load(c, map_location="cpu", weights_only=True)
    raise FileNotFoundError(
        "PerceptualLoss needs the pretrained torchvision VGG19 weights (vgg19-dcbb9e9d.pth, losses.py:43-44): pass the path / a "
        "state dict as `vgg_weights`, set $UEGAN_VGG19_WEIGHTS, or put the file at ./models/vgg19-dcbb9e9d.pth (looked at: %s).  "
        "For benchmarks and tests without the file, opt in to the seeded stand-in explicitly: PerceptualLoss(vgg_weights='seeded')."
        % ", ".join(repr(c) for c in cands if c))


class PerceptualLoss(nn.Module):
    """Fidelity loss (losses.py:12-36): ImageNet-normalise, VGG19 taps relu{1..5}_1, non-affine InstanceNorm on both
    branches, weighted MSE with weights 1/64, 1/64, 1/32, 1/32, 1.  Gradient flows to `x` only.

    vgg_weights: a torchvision-keyed state dict, a path to vgg19-dcbb9e9d.pth, None (search $UEGAN_VGG19_WEIGHTS and
    ./models/; raise if absent, as the reference's `vgg19(pretrained=True)` would), or the string "seeded" for the
    architecture-exact stand-in `seeded_vgg19_weights` (NOT the pretrained network: a fidelity loss for benchmarks/tests only)."""

    def __init__(self, vgg_weights=None, width_div=1):
        super().__init__()
        if isinstance(vgg_weights, dict):
            sd = vgg_weights
        elif vgg_weights == "seeded":
            sd = seeded_vgg19_weights(width_div=width_div)
        else:
            sd = _find_vgg_weights(vgg_weights)
        self.add_module("vgg", VGG19_relu(sd, width_div, deferred_act_grad=True))
        self.weights = [1.0 / 64, 1.0 / 64, 1.0 / 32, 1.0 / 32, 1.0 / 1]
        self.fused = True           # False: one autograd node per layer, two VGG passes of B (the restructuring's own parity reference)
        self.register_buffer("mean", torch.tensor(IMAGENET_MEAN).view(1, -1, 1, 1))
        self.register_buffer("std", torch.tensor(IMAGENET_STD).view(1, -1, 1, 1))

    def _taps(self, img, scale, shift):
        # (img*scale + shift - mean)/std folded into the NCHW->NHWC conversion kernel
        a = [scale / s for s in IMAGENET_STD]
        b = [(shift - m) / s for m, s in zip(IMAGENET_MEAN, IMAGENET_STD)]
        return self.vgg(ops.to_nhwc(img, a=a, b=b))

    def reference_taps(self, y, input_range01=True):
        """VGG taps of the image the loss compares against, computed ahead of the call (fused path only; `forward(..., y_taps=...)`)"""
        if y.shape[1] != 3:
            y = y.repeat(1, 3, 1, 1)
        scale, shift = (1.0, 0.0) if input_range01 else (0.5, 0.5)
        a = [scale / s for s in IMAGENET_STD]
        b = [(shift - m) / s for m, s in zip(IMAGENET_MEAN, IMAGENET_STD)]
        return fused.vgg_reference_taps(self.vgg, y, a, b)

    def forward(self, x, y, input_range01=True, y_taps=None):
        """x, y: [B,3,H,W] in [0,1] like the reference (trainer.py:108 passes (img+1)/2).  With
        input_range01=False the tensors are the raw [-1,1] images and the (img+1)/2 rescale is fused as well."""
        if x.shape[1] != 3:
            x = x.repeat(1, 3, 1, 1)
            y = y.repeat(1, 3, 1, 1)
        scale, shift = (1.0, 0.0) if input_range01 else (0.5, 0.5)
        if self.fused and self.vgg.deferred_act_grad:
            # both images through the frozen VGG19 as ONE batch of 2B (or x alone against taps of y computed ahead), backward over the x
            # half only (uegan_amd/fused.py)
            a = [scale / s for s in IMAGENET_STD]
            b = [(shift - m) / s for m, s in zip(IMAGENET_MEAN, IMAGENET_STD)]
            return fused.vgg_fidelity_loss(self.vgg, self.weights, x, y, a, b, y_taps=y_taps)
        tx = self._taps(x, scale, shift)
        with torch.no_grad():
            ty = self._taps(y, scale, shift)
        return ops.perceptual_taps_loss(tx, ty, self.weights, in_act=ops.ACT_RELU if self.vgg.deferred_act_grad else ops.ACT_NONE)


class GANLoss(nn.Module):
    """Quality loss over the 5 discriminator scales (losses.py:251-409): 'rahinge' (the default, :348-362) and 'rals' (:363-376)
    compare the real list against the fake list; 'original' / 'ls' / 'hinge' / 'w' take ONE list (`for_real` or `for_fake` selects
    it -- trainer.py:92,95,104 pass neither, so with those modes the reference's trainer raises, and so does this)."""

    def __init__(self, gan_mode, target_real_label=1.0, target_fake_label=0.0, tensor=torch.FloatTensor, opt=None):
        super().__init__()
        if gan_mode not in ("ls", "original", "w", "hinge", "rahinge", "rals"):
            raise ValueError("Unexpected gan_mode {}".format(gan_mode))
        self.gan_mode = gan_mode
        self.real_label, self.fake_label, self.Tensor, self.opt = target_real_label, target_fake_label, tensor, opt

    def __call__(self, real_preds, fake_preds, target_is_real=None, for_real=None, for_fake=None, for_discriminator=True):
        if not isinstance(real_preds, list):
            real_preds, fake_preds = [real_preds], [fake_preds]
        real_preds = [p[-1] if isinstance(p, list) else p for p in real_preds]
        fake_preds = [p[-1] if isinstance(p, list) else p for p in fake_preds]
        m = self.gan_mode
        if m == "rahinge":
            return ops.rahinge(real_preds, fake_preds, for_discriminator)
        if m == "rals":
            return variants.rals(real_preds, fake_preds, for_discriminator)
        if for_real:
            preds = real_preds
        elif for_fake:
            preds = fake_preds
        else:
            raise NotImplementedError("nither for real_preds nor for fake_preds")
        target = self.real_label if target_is_real else self.fake_label
        if m == "original":                                                   # losses.py:313-323
            return variants.pred_loss(preds, variants.PRED_BCE, target)
        if m == "ls":                                                         # :324-332
            return variants.pred_loss(preds, variants.PRED_LS, target)
        if m == "hinge":                                                      # :333-347
            if for_discriminator:
                return variants.pred_loss(preds, variants.PRED_HINGE_REAL if target_is_real else variants.PRED_HINGE_FAKE)
            assert target_is_real, "The generator's hinge loss must be aiming for real"
            return variants.pred_loss(preds, variants.PRED_NEG_MEAN)
        return variants.pred_loss(preds, variants.PRED_NEG_MEAN if target_is_real else variants.PRED_POS_MEAN)      # 'w', :378-392


class MultiscaleRecLoss(nn.Module):
    """Identity loss (losses.py:202-231): the criterion (`l1` / `smoothl1` / `l2`, mean reduction) at `scale` scales with
    AvgPool2d(2,2) between, weights 1, 1/2, 1/4 (the reference's weight list has three entries: scale > 3 behaves like 3);
    multiscale=False: the plain criterion."""

    def __init__(self, scale=3, rec_loss_type="l1", multiscale=True):
        super().__init__()
        if rec_loss_type not in ("l1", "smoothl1", "l2"):
            raise NotImplementedError("Loss [{}] is not implemented".format(rec_loss_type))
        self.multiscale, self.rec_loss_type = multiscale, rec_loss_type
        if multiscale:
            self.weights = [1.0, 1.0 / 2, 1.0 / 4][:scale]
            if not self.weights:       # (scale <= 0: the reference's loop body never runs and it returns the int 0)
                raise NotImplementedError("MultiscaleRecLoss needs scale >= 1")

    def forward(self, input, target):
        return ops.multiscale_rec(input, target, self.rec_loss_type, len(self.weights) if self.multiscale else 1)


class TVLoss(nn.Module):
    """Name kept so `from losses import ... TVLoss` (tester.py:9) resolves; the reference never calls it."""

    def __init__(self, tv_loss_weight=1):
        super().__init__()
        self.tv_loss_weight = tv_loss_weight

    def forward(self, x):
        raise NotImplementedError("TVLoss is dead code in the reference (imported, never called) and is not built")
