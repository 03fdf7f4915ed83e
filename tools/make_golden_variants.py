#!/usr/bin/env python3
"""Generate tests/golden/variants_*.npz from the REFERENCE's own code (build container only): the non-default flags of
SURVEY.md 8f-4 -- norm_fun / act_fun / use_sn variants of Generator and Discriminator (models.py:249-281), every GANLoss mode
(losses.py:312-392).  Same method as tools/make_golden.py (reference modules imported unmodified, data-only fixtures).
Each network fixture holds: the state dict BEFORE the forward, the input, the train-mode outputs, the gradients of
sum(out * weight_map) w.r.t. the input and every parameter, the buffers AFTER the forward (running statistics, u / v), and the
eval-mode outputs computed afterwards (which read the updated running statistics)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden as MG  # noqa: E402

G_CONFIGS = {"g_bn_swish_sn": ("BatchNorm", "Swish", True), "g_in_selu": ("InstanceNorm", "SELU", False),
             "g_none_relu_sn": ("none", "ReLU", True), "g_none_none": ("none", "none", False)}
D_CONFIGS = {"d_in_selu_rals": ("InstanceNorm", "SELU", False, "rals"), "d_bn_relu_sn_ls": ("BatchNorm", "ReLU", True, "ls"),
             "d_none_swish_sn": ("none", "Swish", True, "rahinge")}


def randomise(net, seed):
    """non-trivial parameters and statistics: affine norm weights away from (1, 0), running statistics away from (0, 1)"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if p.dim() == 1 and name.split(".")[-2] == "2":      # main.2.{weight,bias}: the norm layer
                p.copy_(1.0 + 0.3 * torch.randn(p.shape, generator=g) if name.endswith("weight") else 0.2 * torch.randn(p.shape, generator=g))
            elif p.dim() == 4:
                p.mul_(1.5)
        for name, b in net.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(0.1 * torch.randn(b.shape, generator=g))
            elif name.endswith("running_var"):
                b.copy_(0.5 + torch.rand(b.shape, generator=g))


def run(net, x, wmaps):
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    net.train()
    x = x.clone().requires_grad_(True)
    out = net(x)
    outs = out if isinstance(out, list) else [out]
    loss = sum((o * w).sum() for o, w in zip(outs, wmaps))
    loss.backward()
    arrs = {"x": x.detach(), "gx": x.grad}
    for k, v in sd0.items():
        arrs["sd0." + k] = v
    for i, (o, w) in enumerate(zip(outs, wmaps)):
        arrs["out%d" % i] = o.detach()
        arrs["w%d" % i] = w
    for k, p in net.named_parameters():
        arrs["grad." + k] = p.grad if p.grad is not None else torch.zeros_like(p)
    for k, b in net.named_buffers():
        arrs["buf1." + k] = b.clone()
    net.eval()
    with torch.no_grad():
        oe = net(x.detach())
    for i, o in enumerate(oe if isinstance(oe, list) else [oe]):
        arrs["eval%d" % i] = o
    return arrs


def msrec(losses):
    """MultiscaleRecLoss(scale, rec_loss_type, multiscale) (losses.py:202-231) for every criterion and weight-list length: value and gradient.
    Differences up to ~3 so that SmoothL1Loss has elements on both sides of |d| = 1."""
    g = torch.Generator().manual_seed(23)
    a = torch.randn(2, 3, 16, 24, generator=g) * 1.2
    b = torch.randn(2, 3, 16, 24, generator=g)
    arrs = {"a": a, "b": b}
    cases = [(sc, kind, True) for kind in ("l1", "smoothl1", "l2") for sc in (1, 2, 3, 5)] + [(3, kind, False) for kind in ("l1", "smoothl1", "l2")]
    for ci, (sc, kind, ms) in enumerate(cases):
        crit = losses.MultiscaleRecLoss(scale=sc, rec_loss_type=kind, multiscale=ms)
        x = a.clone().requires_grad_(True)
        loss = crit(x, b)
        loss.backward()
        arrs["c%02d.loss" % ci] = loss.detach().reshape(1)
        arrs["c%02d.ga" % ci] = x.grad
        arrs["c%02d.meta" % ci] = np.array([str(sc), kind, str(ms)])
    arrs["ncases"] = np.array(len(cases))
    # odd sizes are legal for the single-scale forms (no pooling)
    a2 = torch.randn(1, 3, 7, 9, generator=g)
    b2 = torch.randn(1, 3, 7, 9, generator=g)
    x = a2.clone().requires_grad_(True)
    loss = losses.MultiscaleRecLoss(rec_loss_type="smoothl1", multiscale=False)(x, b2)
    loss.backward()
    arrs.update(odd_a=a2, odd_b=b2, odd_loss=loss.detach().reshape(1), odd_ga=x.grad)
    # ... and for the pooled forms AvgPool2d(2, 2) floors (7 x 9 -> 3 x 4 -> 1 x 2; 10 x 6 -> 5 x 3 -> 2 x 1): the reference's own values
    a3 = torch.randn(2, 3, 10, 6, generator=g) * 1.2
    b3 = torch.randn(2, 3, 10, 6, generator=g)
    rag = [("7x9", a2 * 1.2, b2), ("10x6", a3, b3)]
    nr = 0
    for tag, ra, rb in rag:
        for kind in ("l1", "smoothl1", "l2"):
            for sc in (2, 3):
                x = ra.clone().requires_grad_(True)
                loss = losses.MultiscaleRecLoss(scale=sc, rec_loss_type=kind, multiscale=True)(x, rb)
                loss.backward()
                arrs["r%02d.a" % nr], arrs["r%02d.b" % nr] = ra, rb
                arrs["r%02d.loss" % nr], arrs["r%02d.ga" % nr] = loss.detach().reshape(1), x.grad
                arrs["r%02d.meta" % nr] = np.array([str(sc), kind, tag])
                nr += 1
    arrs["nragged"] = np.array(nr)
    MG.npz("variants_msrec.npz", **arrs)


def main():
    models, losses, _, _ = MG.import_reference()
    if "--msrec" in sys.argv:        # only the MultiscaleRecLoss fixture (added after the others were committed)
        msrec(losses)
        return
    msrec(losses)
    for name, (norm, act, sn) in G_CONFIGS.items():
        torch.manual_seed(101)
        net = models.Generator(8, norm, act, sn)
        randomise(net, 7)
        g = torch.Generator().manual_seed(3)
        x = (torch.rand(2, 3, 32, 32, generator=g) * 2 - 1)
        w = torch.randn(2, 3, 32, 32, generator=g)
        MG.npz("variants_" + name + ".npz", **run(net, x, [w]))
    for name, (norm, act, sn, adv) in D_CONFIGS.items():
        torch.manual_seed(202)
        net = models.Discriminator(8, norm, act, sn, adv)
        randomise(net, 9)
        g = torch.Generator().manual_seed(5)
        x = (torch.rand(2, 3, 96, 96, generator=g) * 2 - 1)
        with torch.no_grad():
            shapes = [tuple(o.shape) for o in net(x)]
        net.load_state_dict({k: v for k, v in net.state_dict().items()})
        # (the probe forward above advanced u / v and the running statistics: rebuild for a clean 'before' state)
        torch.manual_seed(202)
        net = models.Discriminator(8, norm, act, sn, adv)
        randomise(net, 9)
        ws = [torch.randn(s, generator=g) for s in shapes]
        MG.npz("variants_" + name + ".npz", **run(net, x, ws))

    # every GANLoss mode on random prediction lists (5 scales): values and gradients
    g = torch.Generator().manual_seed(11)
    shapes = [(2, 1, 12, 12), (2, 1, 6, 6), (2, 1, 3, 3), (2, 1, 2, 2), (2, 1, 1, 1)]
    real = [torch.randn(s, generator=g) for s in shapes]
    fake = [torch.randn(s, generator=g) * 1.3 + 0.2 for s in shapes]
    arrs = {}
    for i, (r, f) in enumerate(zip(real, fake)):
        arrs["real%d" % i], arrs["fake%d" % i] = r, f
    cases = [("rals", None, None, None, True), ("rals", None, None, None, False), ("rahinge", None, None, None, True)]
    for mode in ("original", "ls", "hinge", "w"):
        for target in (True, False):
            for side in ("real", "fake"):
                for for_d in ((True, False) if mode == "hinge" else (True,)):
                    if mode == "hinge" and not for_d and not target:
                        continue                      # asserts in the reference
                    cases.append((mode, target, side == "real", side == "fake", for_d))
    for ci, (mode, target, for_real, for_fake, for_d) in enumerate(cases):
        crit = losses.GANLoss(mode)
        rr = [t.clone().requires_grad_(True) for t in real]
        ff = [t.clone().requires_grad_(True) for t in fake]
        loss = crit(rr, ff, target, for_real, for_fake, for_discriminator=for_d)
        loss.sum().backward()
        tag = "c%02d" % ci
        arrs[tag + ".loss"] = loss.detach().reshape(-1)
        arrs[tag + ".meta"] = np.array([mode, str(target), str(for_real), str(for_fake), str(for_d)])
        for i in range(5):
            arrs[tag + ".greal%d" % i] = rr[i].grad if rr[i].grad is not None else torch.zeros_like(rr[i])
            arrs[tag + ".gfake%d" % i] = ff[i].grad if ff[i].grad is not None else torch.zeros_like(ff[i])
    arrs["ncases"] = np.array(len(cases))
    MG.npz("variants_ganloss.npz", **arrs)


if __name__ == "__main__":
    main()
