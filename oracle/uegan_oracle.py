"""CPU oracle for the UEGAN hot path -- TEST INFRASTRUCTURE ONLY.

This file is a functional, plain-PyTorch-CPU fp32 restatement of the reference's
algorithm for the path named by BASELINE.json `north_star` (G / D / VGG fidelity
loss / relativistic-hinge loss / multiscale-L1 loss / Adam / the train-step
ordering).  It is NOT part of the product: only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it, and
there only as the checker / the timed CPU baseline.  The product path
(`uegan_amd/`) never imports this module and has no CPU fallback.

Parity pinning: every function here is checked in the build container against
the reference's own `models.py` / `losses.py` imported unmodified from
/root/reference (tools/make_golden.py), and against the committed fixtures
produced by that script (tests/golden/*.npz, tests/test_oracle_golden.py).
The *pretrained* VGG19 weights (vgg19-dcbb9e9d.pth) are not available offline,
so VGG parity is pinned on the architecture + seeded random weights only
("parity unpinned" w.r.t. the pretrained network).

Every function cites the reference file:line it follows (paths relative to
/root/reference).  Tensors are NCHW fp32 like the reference.  Parameters travel
in flat dicts keyed by the reference's state-dict names.
"""
import math
import random

import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.2          # models.py:252  nn.LeakyReLU(0.2)
IN_EPS = 1e-5              # nn.InstanceNorm2d default eps (models.py:227, losses.py:18)
SN_EPS = 1e-12             # nn.utils.spectral_norm default eps (models.py:187)

# ----------------------------------------------------------------------------
# layer helpers
# ----------------------------------------------------------------------------

def reflect_conv(x, w, b, stride):
    """models.py:80-84 / 92-94 / 161-162 / 173-174: ReflectionPad2d((k-1)//2) then Conv2d(padding=0)."""
    p = (w.shape[-1] - 1) // 2
    if p > 0:
        x = F.pad(x, (p, p, p, p), mode="reflect")
    return F.conv2d(x, w, b, stride=stride)


def conv_block(P, prefix, x, stride):
    """ConvBlock (models.py:88-101) with norm 'none' (Identity) and LeakyReLU(0.2)."""
    y = reflect_conv(x, P[prefix + ".main.1.weight"], P[prefix + ".main.1.bias"], stride)
    return F.leaky_relu(y, LRELU_SLOPE)


def sn_conv(P, prefix, x, stride=1):
    """SNConv (models.py:77-86) with use_sn=False: reflect pad + conv, no activation."""
    return reflect_conv(x, P[prefix + ".main.1.weight"], P[prefix + ".main.1.bias"], stride)


def calc_mean_std(feat, eps=1e-5):
    """models.py:204-212 (unbiased var + eps, sqrt)."""
    n, c = feat.shape[:2]
    var = feat.reshape(n, c, -1).var(dim=2) + eps
    std = var.sqrt().reshape(n, c, 1, 1)
    mean = feat.reshape(n, c, -1).mean(dim=2).reshape(n, c, 1, 1)
    return mean, std


def gam(P, prefix, x):
    """GAM.forward (models.py:230-237) with bias=False gate convs, fuse bias, norm=True."""
    mean, std = calc_mean_std(x)
    g = F.conv2d(torch.cat([mean, std], dim=1), P[prefix + ".conv.0.weight"])
    g = F.relu(g)
    g = F.conv2d(g, P[prefix + ".conv.2.weight"])
    y = F.conv2d(torch.cat([x, g.expand_as(x)], dim=1), P[prefix + ".fuse.0.weight"], P[prefix + ".fuse.0.bias"])
    return F.instance_norm(y, eps=IN_EPS)            # nn.InstanceNorm2d(out_nc): non-affine, biased var


def upsample_conv(P, prefix, x):
    """nn.Sequential(Interpolate(2,'bilinear',True), SNConv 1x1) (models.py:23-26, 191-201)."""
    x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    return F.conv2d(x, P[prefix + ".1.main.1.weight"], P[prefix + ".1.main.1.bias"])


# ----------------------------------------------------------------------------
# Generator (models.py:10-74)
# ----------------------------------------------------------------------------

def generator_forward(P, x, return_acts=False):
    """Generator.forward, models.py:44-74."""
    x1 = conv_block(P, "enc1", x, 1)
    x2 = conv_block(P, "enc2", x1, 2)
    x3 = conv_block(P, "enc3", x2, 2)
    x4 = conv_block(P, "enc4", x3, 2)
    x5 = conv_block(P, "enc5", x4, 2)
    x5 = gam(P, "ga5", x5)

    y1 = upsample_conv(P, "upsample1", x5)
    y1 = conv_block(P, "dec1", torch.cat([y1, gam(P, "ga4", x4)], dim=1), 1)
    y2 = upsample_conv(P, "upsample2", y1)
    y2 = conv_block(P, "dec2", torch.cat([y2, gam(P, "ga3", x3)], dim=1), 1)
    y3 = upsample_conv(P, "upsample3", y2)
    y3 = conv_block(P, "dec3", torch.cat([y3, gam(P, "ga2", x2)], dim=1), 1)
    y4 = upsample_conv(P, "upsample4", y3)
    y4 = conv_block(P, "dec4", torch.cat([y4, gam(P, "ga1", x1)], dim=1), 1)

    res = sn_conv(P, "dec5.0", y4 * x1)              # models.py:70, 32-36
    res = torch.tanh(sn_conv(P, "dec5.1", res))
    out = torch.clamp(res + x, min=-1.0, max=1.0)    # models.py:72
    if return_acts:
        return out, dict(x1=x1, x2=x2, x3=x3, x4=x4, x5=x5, y1=y1, y2=y2, y3=y3, y4=y4, res=res)
    return out


# ----------------------------------------------------------------------------
# Discriminator (models.py:104-155) with spectral norm (models.py:185-188 ->
# torch.nn.utils.spectral_norm: 1 power iteration per training forward)
# ----------------------------------------------------------------------------
D_KERNELS = (7, 7, 7, 5, 5)


def spectral_norm_weight(P, prefix, train):
    """torch.nn.utils.spectral_norm.SpectralNorm.compute_weight (dim=0, n_power_iterations=1,
    eps=1e-12): in training mode update u,v IN PLACE (no grad), then sigma=u^T W v with u,v
    constants (cloned), weight = weight_orig / sigma."""
    w = P[prefix + ".weight_orig"]
    u = P[prefix + ".weight_u"]
    v = P[prefix + ".weight_v"]
    wm = w.reshape(w.shape[0], -1)
    if train:
        with torch.no_grad():
            v.copy_(F.normalize(torch.mv(wm.t(), u), dim=0, eps=SN_EPS))
            u.copy_(F.normalize(torch.mv(wm, v), dim=0, eps=SN_EPS))
        u = u.clone()
        v = v.clone()
    sigma = torch.dot(u, torch.mv(wm, v))
    return w / sigma


def discriminator_forward(P, x, train=True, return_feats=False):
    """Discriminator.forward, models.py:139-155; dis_conv_block 158-167; dis_pred_conv_block 170-182."""
    preds, feats = [], []
    h = x
    for i in range(1, 6):
        pre = "d%d.0.1" % i
        w = spectral_norm_weight(P, pre, train)
        h = F.leaky_relu(reflect_conv(h, w, P[pre + ".bias"], 2), LRELU_SLOPE)
        feats.append(h)
        preds.append(torch.tanh(reflect_conv(h, P["d%d_pred.0.1.weight" % i], None, 1)))
    if return_feats:
        return preds, feats
    return preds


# ----------------------------------------------------------------------------
# VGG19 features + fidelity ("perceptual") loss (losses.py:12-164)
# ----------------------------------------------------------------------------
VGG_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
# torchvision `features` indices of the 16 convs of cfg "E"
VGG_CONV_IDX = [0, 2, 5, 7, 10, 12, 14, 16, 19, 21, 23, 25, 28, 30, 32, 34]
VGG_TAPS = {0: "relu1_1", 5: "relu2_1", 10: "relu3_1", 19: "relu4_1", 28: "relu5_1"}
VGG_TAP_WEIGHTS = [1.0 / 64, 1.0 / 64, 1.0 / 32, 1.0 / 32, 1.0]       # losses.py:17
IMAGENET_MEAN = (0.485, 0.456, 0.406)                                   # losses.py:19
IMAGENET_STD = (0.229, 0.224, 0.225)                                    # losses.py:20


def vgg_channels(width_div=1):
    return [v if v == "M" else max(v // width_div, 1) for v in VGG_CFG]


def make_vgg_weights(seed=1234, width_div=1, dtype=torch.float32):
    """Seeded stand-in for vgg19-dcbb9e9d.pth (not available offline): per conv
    `torch.Generator().manual_seed(seed+idx)`, randn * sqrt(2/fan_in), small seeded bias.
    Keys follow torchvision: features.{idx}.{weight,bias}."""
    W = {}
    c = 3
    ci = 0
    for v in vgg_channels(width_div):
        if v == "M":
            continue
        idx = VGG_CONV_IDX[ci]
        g = torch.Generator().manual_seed(seed + idx)
        W["features.%d.weight" % idx] = (torch.randn(v, c, 3, 3, generator=g) * math.sqrt(2.0 / (c * 9))).to(dtype)
        W["features.%d.bias" % idx] = (torch.randn(v, generator=g) * 0.05).to(dtype)
        c = v
        ci += 1
    return W


class _RoundBF16(torch.autograd.Function):
    """bf16 STORAGE emulation: the value (forward) and its gradient (backward) are rounded to bfloat16, arithmetic stays fp32"""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


def vgg_taps(V, x, through="relu5_1", bf16_storage=False):
    """VGG19_relu.forward (losses.py:120-164) restricted to what PerceptualLoss consumes:
    torchvision cfg 'E' = 3x3 conv (zero pad 1, bias) + ReLU, MaxPool2d(2,2) at idx 4,9,18,27.
    Returns the 5 taps relu{1..5}_1.  (conv5_2..5_4 are computed by the reference and
    discarded, losses.py:137-140; they do not influence any output.)
    bf16_storage: emulate the build's throughput mode on the CPU -- every stored activation, its gradient and the conv weights
    are rounded to bf16, all sums stay fp32.  Not reference behaviour: a yardstick for how far bf16 storage ALONE moves a result."""
    taps = []
    h = _RoundBF16.apply(x) if bf16_storage else x
    layer = 0
    ci = 0
    for v in VGG_CFG:
        if v == "M":
            h = F.max_pool2d(h, 2, 2)
            layer += 1
            continue
        idx = VGG_CONV_IDX[ci]
        assert idx == layer
        w = V["features.%d.weight" % idx]
        h = F.relu(F.conv2d(h, w.bfloat16().float() if bf16_storage else w, V["features.%d.bias" % idx], padding=1))
        if bf16_storage:
            h = _RoundBF16.apply(h)
        if idx in VGG_TAPS:
            taps.append(h)
            if VGG_TAPS[idx] == through:
                break
        layer += 2
        ci += 1
    return taps


def perceptual_loss(V, x, y, tap_weights=None, bf16_storage=False):
    """PerceptualLoss.__call__ (losses.py:22-36).  x,y in [0,1] NCHW 3ch.  (tap_weights / bf16_storage: test yardsticks, see vgg_taps)"""
    mean = torch.tensor(IMAGENET_MEAN, dtype=x.dtype).view(1, -1, 1, 1)
    std = torch.tensor(IMAGENET_STD, dtype=x.dtype).view(1, -1, 1, 1)
    x = (x - mean) / std
    y = (y - mean) / std
    tx, ty = vgg_taps(V, x, bf16_storage=bf16_storage), vgg_taps(V, y, bf16_storage=bf16_storage)
    loss = 0
    for w, a, b in zip(VGG_TAP_WEIGHTS if tap_weights is None else tap_weights, tx, ty):
        loss = loss + w * F.mse_loss(F.instance_norm(a, eps=IN_EPS), F.instance_norm(b, eps=IN_EPS))
    return loss


# ----------------------------------------------------------------------------
# Quality loss: relativistic average hinge over 5 scales (losses.py:348-362, 393-409)
# ----------------------------------------------------------------------------

def rahinge_loss(real_preds, fake_preds, for_discriminator):
    """GANLoss('rahinge').__call__ on lists: per scale loss (losses.py:348-362), summed over
    scales, shape [1] (losses.py:397-409)."""
    loss = 0
    for r, f in zip(real_preds, fake_preds):
        r_f = r - torch.mean(f)
        f_r = f - torch.mean(r)
        if for_discriminator:
            li = (torch.mean(F.relu(1 - r_f)) + torch.mean(F.relu(1 + f_r))) / 2
        else:
            li = (torch.mean(F.relu(1 + r_f)) + torch.mean(F.relu(1 - f_r))) / 2
        loss = loss + torch.mean(li.view(1, -1), dim=1)
    return loss


# ----------------------------------------------------------------------------
# Identity loss: multiscale L1 (losses.py:202-231)
# ----------------------------------------------------------------------------

def multiscale_rec(pred, gt, scale=3, kind="l1", multiscale=True):
    """MultiscaleRecLoss(scale, rec_loss_type, multiscale).forward (losses.py:202-231): L1Loss / SmoothL1Loss / MSELoss at the scales of
    the weight list [1, 1/2, 1/4][:scale] with AvgPool2d(2, 2) between; multiscale=False: the plain criterion."""
    crit = {"l1": F.l1_loss, "smoothl1": F.smooth_l1_loss, "l2": F.mse_loss}[kind]
    if not multiscale:
        return crit(pred, gt)
    weights = [1.0, 0.5, 0.25][:scale]
    loss = 0
    for i, w in enumerate(weights):
        loss = loss + w * crit(pred, gt)
        if i != len(weights) - 1:
            pred = F.avg_pool2d(pred, 2, 2)
            gt = F.avg_pool2d(gt, 2, 2)
    return loss


def multiscale_l1(pred, gt, scale=3):
    """MultiscaleRecLoss(scale=3,'l1',multiscale=True).forward (losses.py:219-231)."""
    return multiscale_rec(pred, gt, scale, "l1", True)


# ----------------------------------------------------------------------------
# ImagePool (utils.py:23-50) -- host-side history buffer; python `random` call order kept
# ----------------------------------------------------------------------------

class ImagePool:
    def __init__(self, pool_size, rng=random):
        self.pool_size = pool_size
        self.num_imgs = 0
        self.images = []
        self.rng = rng

    def query(self, images):
        if self.pool_size == 0:
            return images
        out = []
        for image in images:
            image = image.detach().unsqueeze(0)
            if self.num_imgs < self.pool_size:
                self.num_imgs += 1
                self.images.append(image)
                out.append(image)
            else:
                if self.rng.uniform(0, 1) > 0.5:
                    rid = self.rng.randint(0, self.pool_size - 1)
                    tmp = self.images[rid].clone()
                    self.images[rid] = image
                    out.append(tmp)
                else:
                    out.append(image)
        return torch.cat(out, 0)


# ----------------------------------------------------------------------------
# Adam with L2-in-gradient weight decay (torch.optim.Adam semantics; trainer.py:337-338)
# ----------------------------------------------------------------------------

def adam_step(params, grads, state, lr, beta1=0.5, beta2=0.999, eps=1e-8, weight_decay=1e-4):
    """params/grads: dict name->tensor. state: dict with 'step', 'm', 'v'. In place."""
    state["step"] += 1
    t = state["step"]
    bc1 = 1 - beta1 ** t
    bc2 = 1 - beta2 ** t
    for k, p in params.items():
        g = grads[k]
        if weight_decay != 0:
            g = g + weight_decay * p
        m = state["m"][k]
        v = state["v"][k]
        m.mul_(beta1).add_(g, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / bc1)


def new_adam_state(params):
    return {"step": 0, "m": {k: torch.zeros_like(p) for k, p in params.items()},
            "v": {k: torch.zeros_like(p) for k, p in params.items()}}


D_BUFFER_SUFFIXES = (".weight_u", ".weight_v")


def trainable(P):
    """Leaves that the optimizer updates (everything except the spectral-norm u/v buffers)."""
    return {k: v for k, v in P.items() if not k.endswith(D_BUFFER_SUFFIXES)}


# ----------------------------------------------------------------------------
# One training iteration (trainer.py:77-119 with the defaults of config.py:11-81)
# ----------------------------------------------------------------------------

class TrainState:
    def __init__(self, G, D, V, pool_size=50, g_lr=1e-4, d_lr=4e-4, lambda_adv=0.1, lambda_percep=1.0,
                 lambda_idt=0.1, rng=random):
        self.G, self.D, self.V = G, D, V
        self.g_lr, self.d_lr = g_lr, d_lr
        self.lambda_adv, self.lambda_percep, self.lambda_idt = lambda_adv, lambda_percep, lambda_idt
        self.pool = ImagePool(pool_size, rng)
        self.g_opt = new_adam_state(trainable(G))
        self.d_opt = new_adam_state(trainable(D))


def _with_grad(P):
    Q = {}
    for k, v in P.items():
        if k.endswith(D_BUFFER_SUFFIXES):
            Q[k] = v
        else:
            Q[k] = v.detach().requires_grad_(True)
    return Q


def train_step(S, real_raw, real_exp, return_grads=False):
    """trainer.py:85-119.  Returns dict of the five logged scalars (and grads if asked).
    Updates S.G / S.D (incl. weight_u/v) / optimizer states / pool in place."""
    out = {}
    Gp = _with_grad(S.G)
    fake_exp = generator_forward(Gp, real_raw)                              # :85
    fake_store = S.pool.query(fake_exp)                                     # :86
    # ---- update D (:89-98)
    Dp = _with_grad(S.D)
    real_preds = discriminator_forward(Dp, real_exp, True)                  # :90
    fake_preds = discriminator_forward(Dp, fake_store.detach(), True)       # :91
    d_loss = rahinge_loss(real_preds, fake_preds, True)                     # :92
    input_preds = discriminator_forward(Dp, real_raw, True)                 # :94 (adv_input=True)
    d_loss = d_loss + rahinge_loss(real_preds, input_preds, True)           # :95
    d_train = trainable(Dp)
    d_grads = torch.autograd.grad(d_loss.sum(), list(d_train.values()))     # :96
    d_grads = dict(zip(d_train.keys(), d_grads))
    with torch.no_grad():
        adam_step(trainable(S.D), d_grads, S.d_opt, S.d_lr)                 # :97
    out["d_loss"] = float(d_loss)                                           # :98
    # ---- update G (:101-119); D params are now the UPDATED ones, their grads are dead work
    Dp = {k: v.detach() for k, v in S.D.items()}
    real_preds = discriminator_forward(Dp, real_exp, True)                  # :102
    fake_preds = discriminator_forward(Dp, fake_exp, True)                  # :103
    g_adv = S.lambda_adv * rahinge_loss(real_preds, fake_preds, False)      # :104
    g_percep = S.lambda_percep * perceptual_loss(S.V, (fake_exp + 1.) / 2., (real_raw + 1.) / 2.)   # :108
    real_exp_idt = generator_forward(Gp, real_exp)                          # :112
    g_idt = S.lambda_idt * multiscale_l1(real_exp_idt, real_exp)            # :113
    g_loss = g_adv + g_percep + g_idt                                       # :106,110,115
    g_train = trainable(Gp)
    g_grads = torch.autograd.grad(g_loss.sum(), list(g_train.values()), allow_unused=True)   # :117
    g_grads = {k: (g if g is not None else torch.zeros_like(p)) for (k, p), g in zip(g_train.items(), g_grads)}
    with torch.no_grad():
        adam_step(trainable(S.G), g_grads, S.g_opt, S.g_lr)                 # :118
    out.update(g_adv=float(g_adv), g_percep=float(g_percep), g_idt=float(g_idt), g_loss=float(g_loss))
    if return_grads:
        out["d_grads"], out["g_grads"] = d_grads, g_grads
        out["fake_exp"] = fake_exp.detach()
    return out


# ----------------------------------------------------------------------------
# Parameter construction (shapes from models.py:12-42, 105-137; init trainer.py:357-390)
# ----------------------------------------------------------------------------

def generator_param_shapes(cd=32):
    S = {}
    def conv(prefix, cin, cout, k, bias=True):
        S[prefix + ".weight"] = (cout, cin, k, k)
        if bias:
            S[prefix + ".bias"] = (cout,)
    conv("enc1.main.1", 3, cd, 7)
    for i, m in enumerate([1, 2, 4, 8]):
        conv("enc%d.main.1" % (i + 2), cd * m, cd * m * 2, 3)
    for i, m in enumerate([16, 8, 4, 2]):
        conv("upsample%d.1.main.1" % (i + 1), cd * m, cd * m // 2, 1)
        conv("dec%d.main.1" % (i + 1), cd * m, cd * m // 2, 3)
    conv("dec5.0.main.1", cd, cd, 3)
    conv("dec5.1.main.1", cd, 3, 7)
    for i, m in zip([5, 4, 3, 2, 1], [16, 8, 4, 2, 1]):
        c = cd * m
        conv("ga%d.conv.0" % i, 2 * c, c // 8, 1, bias=False)
        conv("ga%d.conv.2" % i, c // 8, c, 1, bias=False)
        conv("ga%d.fuse.0" % i, 2 * c, c, 1)
    return S


def discriminator_param_shapes(cd=32):
    S = {}
    cin = 3
    for i, (k, m) in enumerate(zip(D_KERNELS, [1, 2, 4, 8, 16])):
        cout = cd * m
        pre = "d%d.0.1" % (i + 1)
        S[pre + ".bias"] = (cout,)
        S[pre + ".weight_orig"] = (cout, cin, k, k)
        S[pre + ".weight_u"] = (cout,)
        S[pre + ".weight_v"] = (cin * k * k,)
        S["d%d_pred.0.1.weight" % (i + 1)] = (1, cout, k, k)
        cin = cout
    return S


def init_params(shapes, seed, mode="default", gain=0.02):
    """Seeded parameters.  mode 'orthogonal' = init_weights('orthogonal', 0.02), biases 0
    (trainer.py:357-390); mode 'default' = a non-degenerate kaiming-uniform-like init (SURVEY 7:
    orthogonal-0.02 makes G ~ identity and losses ill-conditioned).  u,v ~ normalize(randn)."""
    g = torch.Generator().manual_seed(seed)
    P = {}
    for k, shp in shapes.items():
        if k.endswith(".weight_u") or k.endswith(".weight_v"):
            P[k] = F.normalize(torch.randn(shp, generator=g), dim=0, eps=SN_EPS)
        elif k.endswith(".bias"):
            if mode == "orthogonal":
                P[k] = torch.zeros(shp)
            else:
                P[k] = (torch.rand(shp, generator=g) * 2 - 1) * 0.1
        else:
            fan_in = shp[1] * shp[2] * shp[3]
            if mode == "orthogonal":
                rows, cols = shp[0], fan_in
                a = torch.randn(max(rows, cols), min(rows, cols), generator=g)
                q, r = torch.linalg.qr(a)
                q = q * torch.sign(torch.diagonal(r)).unsqueeze(0)
                if rows < cols:
                    q = q.t()
                P[k] = (gain * q[:rows, :cols]).reshape(shp).contiguous()
            else:
                bound = 1.0 / math.sqrt(fan_in)
                P[k] = (torch.rand(shp, generator=g) * 2 - 1) * bound * math.sqrt(3.0)
    return P


# ----------------------------------------------------------------------------
# evaluation metrics of the inference configuration (numpy; metrics/CalcPSNR.py, metrics/CalcSSIM.py)
# ----------------------------------------------------------------------------

def to_uint8_image(x):
    """tester.py:70-71 `save_image(denorm(fake))`: utils.py:128-130 denorm, then torchvision.utils.save_image's
    `mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(uint8)` (torchvision 0.5, the reference's pinned version)."""
    out = ((x.detach().clone() + 1) / 2.0).clamp_(0, 1)
    return out.mul(255).add_(0.5).clamp_(0, 255).permute(0, 2, 3, 1).to(torch.uint8)


def psnr_u8(img1, img2, crop_border=4):
    """metrics/CalcPSNR.py:85-92 (`10 log10(data_range^2 / mse)`, float64) after the border crop of :56-57; HWC arrays in [0,255]."""
    import numpy as np
    a = np.asarray(img1).astype(np.float64)
    b = np.asarray(img2).astype(np.float64)
    if crop_border:
        a = a[crop_border:-crop_border, crop_border:-crop_border]
        b = b[crop_border:-crop_border, crop_border:-crop_border]
    mse = np.mean((a - b) ** 2, dtype=np.float64)
    if mse == 0:
        return float("inf")
    return float(10 * np.log10(255.0 ** 2 / mse))


def ssim_u8_skimage(img1, img2, crop_border=4):
    """metrics/CalcSSIM.py:63 `ssim_skimage(GT*255, Gen*255, multichannel=True, data_range=255)` after the crop of :56-57.
    skimage (pinned by the reference's environment, absent here -> "parity unpinned" against the package itself) is restated from
    its published algorithm, skimage.metrics.structural_similarity with default arguments: win_size 7, uniform window
    (scipy.ndimage.uniform_filter, mode 'reflect'), K1 0.01, K2 0.03, use_sample_covariance=True (cov_norm = NP/(NP-1)),
    S cropped by (win_size-1)//2 on every side and averaged; multichannel = mean of the per-channel results."""
    import numpy as np
    from scipy.ndimage import uniform_filter
    a = np.asarray(img1).astype(np.float64)
    b = np.asarray(img2).astype(np.float64)
    if crop_border:
        a = a[crop_border:-crop_border, crop_border:-crop_border]
        b = b[crop_border:-crop_border, crop_border:-crop_border]
    win, K1, K2, R = 7, 0.01, 0.03, 255.0
    NP = win ** 2
    cov_norm = NP / (NP - 1.0)
    C1, C2 = (K1 * R) ** 2, (K2 * R) ** 2
    pad = (win - 1) // 2
    vals = []
    for c in range(a.shape[2]):
        X, Y = a[..., c], b[..., c]
        ux, uy = uniform_filter(X, size=win), uniform_filter(Y, size=win)
        uxx, uyy, uxy = uniform_filter(X * X, size=win), uniform_filter(Y * Y, size=win), uniform_filter(X * Y, size=win)
        vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
        S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
        vals.append(S[pad:-pad, pad:-pad].mean(dtype=np.float64))
    return float(np.mean(vals))


# ----------------------------------------------------------------------------
# Data-parallel yardstick: N ranks, replicated weights, per-rank batch + pool, gradients AVERAGED over ranks
# before each Adam update.  Not reference behaviour (the reference's multi-GPU mode is nn.DataParallel): the definition the
# build's RCCL path must equal -- "an N-rank step = one Adam step on the mean of N independent reference steps' gradients".
# ----------------------------------------------------------------------------

# --------------------------------------------------------------------------------------------------------------------
# Input transforms (data_loader.py:74-82 train, :95-100 test).  torchvision is not in this image; its PIL-backend transforms are
# thin calls into Pillow (RandomCrop -> Image.crop, Resize -> Image.resize(BILINEAR), flips -> Image.transpose, ToTensor ->
# uint8 HWC -> CHW float / 255, Normalize -> (t - mean) / std), restated here on Pillow ITSELF: the pixel arithmetic is pinned by
# the reference's own dependency, the random-draw order is a convention (see uegan_amd/data.py).
# --------------------------------------------------------------------------------------------------------------------
def _to_tensor_normalize(img):
    import numpy as np
    t = torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1).contiguous().float().div(255)       # ToTensor (:79)
    return (t - 0.5) / 0.5                                                                              # Normalize(0.5, 0.5) (:80-81)


def train_transform(rgb, top, left, crop, resize, flip_bits):
    """rgb: uint8 [H, W, 3] numpy.  RandomCrop(crop) at (top, left) -> Resize([resize, resize]) -> flips -> ToTensor -> Normalize."""
    from PIL import Image
    img = Image.fromarray(rgb, "RGB").crop((left, top, left + crop, top + crop))                        # :75
    img = img.resize((resize, resize), Image.BILINEAR)                                                  # :76
    if flip_bits & 1:
        img = img.transpose(Image.FLIP_LEFT_RIGHT)                                                      # :77
    if flip_bits & 2:
        img = img.transpose(Image.FLIP_TOP_BOTTOM)                                                      # :78
    return _to_tensor_normalize(img)


def test_transform(rgb, size):
    """Resize([size, size]) of the whole image -> ToTensor -> Normalize (data_loader.py:95-100)"""
    from PIL import Image
    return _to_tensor_normalize(Image.fromarray(rgb, "RGB").resize((size, size), Image.BILINEAR))


# --------------------------------------------------------------------------------------------------------------------
# Non-default flags: restated for the pieces that have no network fixture of their own; the network variants
# are pinned directly by reference-generated fixtures (tools/make_golden_variants.py -> tests/golden/variants_*.npz).
# --------------------------------------------------------------------------------------------------------------------
def gan_loss(mode, real_preds, fake_preds, target_is_real=None, for_real=None, for_fake=None, for_discriminator=True,
             real_label=1.0, fake_label=0.0):
    """GANLoss.__call__ over lists (losses.py:393-409) of GANLoss.loss (losses.py:312-392); returns shape [1]"""
    total = 0
    for r, f in zip(real_preds, fake_preds):
        if mode in ("rahinge", "rals"):
            rf, fr = r - f.mean(), f - r.mean()
            sgn = 1.0 if for_discriminator else -1.0
            if mode == "rahinge":                                                                  # :348-362
                l = (F.relu(1 - sgn * rf).mean() + F.relu(1 + sgn * fr).mean()) / 2
            else:                                                                                  # :363-376
                l = (((rf - sgn) ** 2).mean() + ((fr + sgn) ** 2).mean()) / 2
        else:
            if for_real:
                p = r
            elif for_fake:
                p = f
            else:
                raise NotImplementedError("nither for real_preds nor for fake_preds")
            t = torch.full_like(p, real_label if target_is_real else fake_label)
            if mode == "original":                                                                 # :313-323
                l = F.binary_cross_entropy_with_logits(p, t)
            elif mode == "ls":                                                                     # :324-332
                l = F.mse_loss(p, t)
            elif mode == "hinge":                                                                  # :333-347
                if for_discriminator:
                    l = -torch.min((p - 1) if target_is_real else (-p - 1), torch.zeros_like(p)).mean()
                else:
                    assert target_is_real
                    l = -p.mean()
            else:                                                                                  # wgan, :378-392
                l = -p.mean() if target_is_real else p.mean()
        total = total + l.reshape(1)
    return total


def rmsprop_step(params, grads, square_avg, lr, alpha=0.9, eps=1e-8):
    """torch.optim.RMSprop as constructed at trainer.py:341-342 (weight_decay 0, momentum 0, centered False), in place"""
    for p, g, v in zip(params, grads, square_avg):
        v.mul_(alpha).addcmul_(g, g, value=1 - alpha)
        p.addcdiv_(g, v.sqrt().add_(eps), value=-lr)


def train_step_data_parallel(S, pools, shards):
    """S: one TrainState (the replicated weights / optimizer states; S.pool unused); pools[r], shards[r] = (real_raw, real_exp)
    of rank r.  Per rank the arithmetic is exactly train_step's (trainer.py:85-119); returns per-rank loss dicts."""
    n = len(shards)
    outs = [dict() for _ in range(n)]
    fakes, Gps = [], []
    d_sum, D_after = None, None
    for r, (real_raw, real_exp) in enumerate(shards):
        Gp = _with_grad(S.G)
        fake_exp = generator_forward(Gp, real_raw)
        fake_store = pools[r].query(fake_exp)
        Dp = _with_grad({k: v.clone() for k, v in S.D.items()})           # every rank starts from the same u/v
        real_preds = discriminator_forward(Dp, real_exp, True)
        fake_preds = discriminator_forward(Dp, fake_store.detach(), True)
        d_loss = rahinge_loss(real_preds, fake_preds, True)
        input_preds = discriminator_forward(Dp, real_raw, True)
        d_loss = d_loss + rahinge_loss(real_preds, input_preds, True)
        d_train = trainable(Dp)
        g = dict(zip(d_train.keys(), torch.autograd.grad(d_loss.sum(), list(d_train.values()))))
        d_sum = g if d_sum is None else {k: d_sum[k] + g[k] for k in g}
        D_after = {k: v.detach() for k, v in Dp.items()}                  # u/v after three forwards (identical on every rank)
        outs[r]["d_loss"] = float(d_loss.detach())
        fakes.append(fake_exp)
        Gps.append(Gp)
    for k in S.D:
        if k.endswith(D_BUFFER_SUFFIXES):
            S.D[k] = D_after[k].clone()
    with torch.no_grad():
        adam_step(trainable(S.D), {k: v / n for k, v in d_sum.items()}, S.d_opt, S.d_lr)
    g_sum, D_after = None, None
    for r, (real_raw, real_exp) in enumerate(shards):
        Gp, fake_exp = Gps[r], fakes[r]
        Dp = {k: v.detach().clone() for k, v in S.D.items()}
        real_preds = discriminator_forward(Dp, real_exp, True)
        fake_preds = discriminator_forward(Dp, fake_exp, True)
        g_adv = S.lambda_adv * rahinge_loss(real_preds, fake_preds, False)
        g_percep = S.lambda_percep * perceptual_loss(S.V, (fake_exp + 1.) / 2., (real_raw + 1.) / 2.)
        real_exp_idt = generator_forward(Gp, real_exp)
        g_idt = S.lambda_idt * multiscale_l1(real_exp_idt, real_exp)
        g_loss = g_adv + g_percep + g_idt
        g_train = trainable(Gp)
        gr = torch.autograd.grad(g_loss.sum(), list(g_train.values()), allow_unused=True)
        gr = {k: (g if g is not None else torch.zeros_like(p)) for (k, p), g in zip(g_train.items(), gr)}
        g_sum = gr if g_sum is None else {k: g_sum[k] + gr[k] for k in gr}
        D_after = Dp
        outs[r].update(g_adv=float(g_adv), g_percep=float(g_percep), g_idt=float(g_idt), g_loss=float(g_loss))
    for k in S.D:
        if k.endswith(D_BUFFER_SUFFIXES):
            S.D[k] = D_after[k].clone()
    with torch.no_grad():
        adam_step(trainable(S.G), {k: v / n for k, v in g_sum.items()}, S.g_opt, S.g_lr)
    return outs
