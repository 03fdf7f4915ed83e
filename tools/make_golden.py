#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own code (build container only).

Imports /root/reference/models.py and losses.py unmodified (a stub `torchvision.models.vgg19`
with the cfg-"E" layout is injected because torchvision is absent), runs
them on seeded inputs / parameters, cross-checks this repo's CPU oracle (oracle/uegan_oracle.py)
against them, and writes inputs + expected outputs as fixtures.  Nothing from /root/reference is
copied: the fixtures hold data only.  The train-step driver below is a transcription of
trainer.py:85-119 over the reference's own nn.Modules and torch.optim.Adam (trainer.py itself is
not importable here: tensorflow / torchvision.utils / cv2 at import time).
"""
import ast
import os
import random
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import uegan_oracle as O  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def import_reference():
    def vgg19(pretrained=False):
        L = []
        c = 3
        for v in O.VGG_CFG:
            if v == "M":
                L.append(nn.MaxPool2d(2, 2))
            else:
                L += [nn.Conv2d(c, v // vgg19.width_div, 3, padding=1), nn.ReLU(inplace=True)]
                c = v // vgg19.width_div
        m = nn.Module()
        m.features = nn.Sequential(*L)
        return m
    vgg19.width_div = 1
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvm.vgg19 = vgg19
    tv.models = tvm
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tvm
    sys.path.insert(0, REF)
    import models
    import losses
    # ImagePool: utils.py is not importable (tensorflow, scipy.misc); exec only that class.
    src = open(os.path.join(REF, "utils.py")).read()
    tree = ast.parse(src)
    node = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "ImagePool"][0]
    ns = {"torch": torch, "random": random}
    exec(compile(ast.Module([node], []), "utils.py::ImagePool", "exec"), ns)
    return models, losses, ns["ImagePool"], vgg19


def npz(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print("wrote", path, "%.2f MB" % (os.path.getsize(path) / 1e6))


def maxrel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def load_ref_G(models, P, cd):
    G = models.Generator(cd, "none", "LeakyReLU", False)
    G.load_state_dict({k: v.clone() for k, v in P.items()})
    return G


def load_ref_D(models, P, cd):
    D = models.Discriminator(cd, "none", "LeakyReLU", True, "rahinge")
    missing = D.load_state_dict({k: v.clone() for k, v in P.items()}, strict=True)
    return D


def load_ref_percep(losses, vgg19, V, width_div):
    vgg19.width_div = width_div
    Pm = losses.PerceptualLoss()
    sd = {}
    for name, p in Pm.vgg.named_parameters():
        # names look like relu1_1.0.weight -> torchvision features.0.weight
        idx = name.split(".")[1]
        p.data.copy_(V["features.%s.%s" % (idx, name.split(".")[2])])
    vgg19.width_div = 1
    return Pm


def checksum(t):
    t = t.detach().double().reshape(-1)
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()] + t[:13].tolist())


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    models, losses, RefImagePool, vgg19 = import_reference()
    report = []

    # ------------------------------------------------------------------ G forward/backward (cd=8, 96^2 and 32^2)
    cd = 8
    for mode, seed in (("default", 11), ("orthogonal", 12)):
        P = O.init_params(O.generator_param_shapes(cd), seed, mode)
        G = load_ref_G(models, P, cd)
        g = torch.Generator().manual_seed(100 + seed)
        x = (torch.rand(2, 3, 96, 96, generator=g) * 2 - 1)
        xs = (torch.rand(1, 3, 32, 32, generator=g) * 2 - 1)
        r = torch.randn(2, 3, 96, 96, generator=g)
        x.requires_grad_(True)
        out = G(x)
        (out * r).sum().backward()
        ref_grads = {k: p.grad.clone() for k, p in G.named_parameters()}
        gx = x.grad.clone()
        # hooks for activations
        Po = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        xo = x.detach().clone().requires_grad_(True)
        out_o, acts = O.generator_forward(Po, xo, return_acts=True)
        (out_o * r).sum().backward()
        e = maxrel(out_o, out)
        eg = max(maxrel(Po[k].grad if Po[k].grad is not None else torch.zeros_like(Po[k]), ref_grads[k]) for k in ref_grads
                 if ref_grads[k].abs().max() > 1e-6)
        report.append("G %s: oracle-vs-ref out %.2e grads %.2e gx %.2e" % (mode, e, eg, maxrel(xo.grad, gx)))
        assert e < 1e-5 and eg < 1e-4, report[-1]
        with torch.no_grad():
            out_s = G(xs)
        sel = ["enc1.main.1.weight", "enc3.main.1.weight", "dec1.main.1.weight", "dec4.main.1.bias", "dec5.1.main.1.weight",
               "upsample2.1.main.1.weight", "ga3.fuse.0.weight", "ga1.fuse.0.weight", "dec5.0.main.1.weight", "enc5.main.1.bias"]
        # orthogonal set: parameters are regenerated by the tests from the seeded recipe
        # (oracle.init_params) and verified against these checksums -- keeps the fixture small
        arrs = {"param/" + k: v for k, v in P.items()} if mode == "default" else {}
        arrs["param_seed"] = np.array(seed)
        arrs["param_checksums"] = np.array([checksum(P[k])[:3] for k in sorted(P)])
        arrs.update(x=x.detach(), xs=xs, r=r, out=out.detach(), out_s=out_s, gx=gx)
        arrs.update({"act/" + k: v.detach() for k, v in acts.items() if k in ("x1", "x5", "y1", "y4", "res")})
        arrs.update({"grad/" + k: ref_grads[k] for k in sel})
        arrs.update({"gradnorm/" + k: ref_grads[k].norm() for k in ref_grads})
        npz("g_cd8_%s.npz" % mode, **arrs)

    # ------------------------------------------------------------------ D forward/backward (cd=8, 96^2), train + eval, u/v evolution
    P = O.init_params(O.discriminator_param_shapes(cd), 21, "default")
    D = load_ref_D(models, P, cd)
    D.train()
    g = torch.Generator().manual_seed(121)
    x = (torch.rand(2, 3, 96, 96, generator=g) * 2 - 1).requires_grad_(True)
    rs = [torch.randn(2, 1, 96 // 2 ** (i + 1), 96 // 2 ** (i + 1), generator=g) for i in range(5)]
    preds = D(x)
    sum((p * r).sum() for p, r in zip(preds, rs)).backward()
    ref_grads = {k: p.grad.clone() for k, p in D.named_parameters()}
    gx = x.grad.clone()
    uv1 = {k: v.clone() for k, v in D.state_dict().items() if k.endswith(("_u", "_v"))}
    with torch.no_grad():
        preds2 = D(x)            # second training forward: u,v advance again
    uv2 = {k: v.clone() for k, v in D.state_dict().items() if k.endswith(("_u", "_v"))}
    D.eval()
    with torch.no_grad():
        preds_eval = D(x)
    Po = {k: (v.clone().requires_grad_(True) if not k.endswith(("_u", "_v")) else v.clone()) for k, v in P.items()}
    xo = x.detach().clone().requires_grad_(True)
    po = O.discriminator_forward(Po, xo, True)
    sum((p * r).sum() for p, r in zip(po, rs)).backward()
    e = max(maxrel(a, b) for a, b in zip(po, preds))
    eg = max(maxrel(Po[k].grad, ref_grads[k]) for k in ref_grads)
    eu = max(maxrel(Po[k], uv1[k]) for k in uv1)
    with torch.no_grad():
        po2 = O.discriminator_forward(Po, xo, True)
        poe = O.discriminator_forward(Po, xo, False)
    e2 = max(maxrel(a, b) for a, b in zip(po2, preds2))
    ee = max(maxrel(a, b) for a, b in zip(poe, preds_eval))
    report.append("D: oracle-vs-ref preds %.2e grads %.2e u/v %.2e 2nd-fwd %.2e eval %.2e gx %.2e" % (e, eg, eu, e2, ee, maxrel(xo.grad, gx)))
    assert max(e, e2, ee, eu) < 1e-5 and eg < 1e-4, report[-1]
    arrs = {"param/" + k: v for k, v in P.items()}
    arrs.update(x=x.detach(), gx=gx)
    for i in range(5):
        arrs["r%d" % i] = rs[i]
        arrs["pred%d" % i] = preds[i].detach()
        arrs["pred2_%d" % i] = preds2[i]
        arrs["pred_eval%d" % i] = preds_eval[i]
    arrs.update({"uv1/" + k: v for k, v in uv1.items()})
    arrs.update({"uv2/" + k: v for k, v in uv2.items()})
    arrs.update({"grad/" + k: v for k, v in ref_grads.items() if "d1." in k or "d3." in k or "pred" in k or "d5.0.1.bias" in k})
    arrs.update({"gradnorm/" + k: v.norm() for k, v in ref_grads.items()})
    npz("d_cd8.npz", **arrs)

    # ------------------------------------------------------------------ losses on seeded tensors
    g = torch.Generator().manual_seed(31)
    A = losses.GANLoss("rahinge", tensor=torch.FloatTensor)
    reals = [torch.tanh(torch.randn(2, 1, s, s, generator=g)).requires_grad_(True) for s in (48, 24, 12, 6, 3)]
    fakes = [torch.tanh(torch.randn(2, 1, s, s, generator=g) - 0.3).requires_grad_(True) for s in (48, 24, 12, 6, 3)]
    arrs = {}
    for name, ford in (("d", True), ("g", False)):
        for t in reals + fakes:
            t.grad = None
        l = A(reals, fakes, None, None, for_discriminator=ford)
        assert tuple(l.shape) == (1,)
        l.sum().backward()
        lo = O.rahinge_loss(reals, fakes, ford)
        assert abs(float(lo) - float(l)) < 1e-6
        arrs["rahinge_%s" % name] = l.detach()
        for i in range(5):
            arrs["rahinge_%s_greal%d" % (name, i)] = reals[i].grad.clone()
            arrs["rahinge_%s_gfake%d" % (name, i)] = fakes[i].grad.clone()
    for i in range(5):
        arrs["real%d" % i] = reals[i].detach()
        arrs["fake%d" % i] = fakes[i].detach()
    I = losses.MultiscaleRecLoss(3, "l1", True)
    a = (torch.rand(2, 3, 96, 96, generator=g) * 2 - 1).requires_grad_(True)
    b = (torch.rand(2, 3, 96, 96, generator=g) * 2 - 1)
    l = I(a, b)
    l.backward()
    assert abs(float(O.multiscale_l1(a, b)) - float(l)) < 1e-6
    arrs.update(msl1_a=a.detach(), msl1_b=b, msl1=l.detach(), msl1_ga=a.grad.clone())
    # perceptual loss: width/8 VGG with committed weights
    V = O.make_vgg_weights(seed=1234, width_div=8)
    Pm = load_ref_percep(losses, vgg19, V, 8)
    px = torch.rand(2, 3, 96, 96, generator=g).requires_grad_(True)
    py = (px.detach() + 0.1 * torch.randn(2, 3, 96, 96, generator=g)).clamp(0, 1)
    l = Pm(px, py)
    l.backward()
    pxo = px.detach().clone().requires_grad_(True)
    lo = O.perceptual_loss(V, pxo, py)
    lo.backward()
    report.append("percep(w/8): ref %.6e oracle %.6e grad rel %.2e" % (float(l), float(lo), maxrel(pxo.grad, px.grad)))
    assert abs(float(lo) - float(l)) / abs(float(l)) < 1e-5 and maxrel(pxo.grad, px.grad) < 1e-4
    arrs.update(percep_x=px.detach(), percep_y=py, percep=l.detach(), percep_gx=px.grad.clone())
    arrs.update({"vgg8/" + k: v for k, v in V.items()})
    with torch.no_grad():
        feats = Pm.vgg((px - Pm.mean) / Pm.std)
    for k in ("relu1_1", "relu3_1", "relu5_1"):
        arrs["vgg8_tap/" + k] = feats[k]
    npz("losses.npz", **arrs)

    # full-width VGG: checksums only (weights regenerated from the seeded recipe by the tests)
    Vf = O.make_vgg_weights(seed=1234, width_div=1)
    Pf = load_ref_percep(losses, vgg19, Vf, 1)
    px = torch.rand(1, 3, 64, 64, generator=g).requires_grad_(True)
    py = (px.detach() + 0.1 * torch.randn(1, 3, 64, 64, generator=g)).clamp(0, 1)
    l = Pf(px, py)
    l.backward()
    lo = O.perceptual_loss(Vf, px.detach(), py)
    assert abs(float(lo) - float(l)) / abs(float(l)) < 1e-5
    npz("percep_full.npz", x=px.detach(), y=py, percep=l.detach(), gx=px.grad.clone(),
        wsum=np.array([checksum(Vf[k])[:3] for k in sorted(Vf)]))

    # ------------------------------------------------------------------ full train steps (cd=8 and cd=32), reference modules
    for cd_t, tag, modes in ((8, "cd8", ("default", "orthogonal")), (32, "cd32", ("default",))):
        for mode in modes:
            PG = O.init_params(O.generator_param_shapes(cd_t), 41, mode)
            PD = O.init_params(O.discriminator_param_shapes(cd_t), 42, mode)
            V = O.make_vgg_weights(seed=1234, width_div=8)
            G = load_ref_G(models, PG, cd_t)
            D = load_ref_D(models, PD, cd_t)
            G.train(); D.train()
            Pm = load_ref_percep(losses, vgg19, V, 8)
            I = losses.MultiscaleRecLoss(3, "l1", True)
            A = losses.GANLoss("rahinge", tensor=torch.FloatTensor)
            g_opt = torch.optim.Adam(G.parameters(), lr=1e-4, betas=[0.5, 0.999], weight_decay=0.0001)
            d_opt = torch.optim.Adam(D.parameters(), lr=4e-4, betas=[0.5, 0.999], weight_decay=0.0001)
            random.seed(1990)
            pool = RefImagePool(3)          # small pool so the swap branch is exercised by step 2
            S = O.TrainState({k: v.clone() for k, v in PG.items()}, {k: v.clone() for k, v in PD.items()}, V, pool_size=3,
                             rng=random.Random(1990))
            gi = torch.Generator().manual_seed(1990)
            arrs = {}
            nsteps = 3
            for step in range(nsteps):
                real_raw = torch.rand(2, 3, 96, 96, generator=gi) * 2 - 1
                real_exp = torch.rand(2, 3, 96, 96, generator=gi) * 2 - 1
                # --- transcription of trainer.py:85-119
                fake_exp = G(real_raw)
                fake_exp_store = pool.query(fake_exp)
                d_opt.zero_grad()
                real_exp_preds = D(real_exp)
                fake_exp_preds = D(fake_exp_store.detach())
                d_loss = A(real_exp_preds, fake_exp_preds, None, None, for_discriminator=True)
                input_preds = D(real_raw)
                d_loss += A(real_exp_preds, input_preds, None, None, for_discriminator=True)
                d_loss.backward()
                d_grad_norms = {k: p.grad.norm().item() for k, p in D.named_parameters()}
                d_opt.step()
                g_opt.zero_grad()
                real_exp_preds = D(real_exp)
                fake_exp_preds = D(fake_exp)
                g_adv = 0.1 * A(real_exp_preds, fake_exp_preds, None, None, for_discriminator=False)
                g_loss = g_adv
                g_percep = 1.0 * Pm((fake_exp + 1.) / 2., (real_raw + 1.) / 2.)
                g_loss = g_loss + g_percep
                real_exp_idt = G(real_exp)
                g_idt = 0.1 * I(real_exp_idt, real_exp)
                g_loss = g_loss + g_idt
                g_loss.backward()
                g_grad_norms = {k: (p.grad.norm().item() if p.grad is not None else 0.0) for k, p in G.named_parameters()}
                g_opt.step()
                ref = dict(d_loss=float(d_loss), g_adv=float(g_adv), g_percep=float(g_percep), g_idt=float(g_idt), g_loss=float(g_loss))
                # --- oracle on the same inputs
                o = O.train_step(S, real_raw, real_exp, return_grads=True)
                for k in ref:
                    rel = abs(o[k] - ref[k]) / (abs(ref[k]) + 1e-8)
                    assert rel < 2e-4 or abs(o[k] - ref[k]) < 2e-7, (tag, mode, step, k, o[k], ref[k])
                # weights: error relative to (max|w| + lr): at orthogonal-0.02 init some tensors (zero
                # biases with 1e-29 grads) move by ~1e-25 per step and a pure relative metric is noise.
                ew = max(float((S.G[k] - v).abs().max() / (v.abs().max() + 1e-4)) for k, v in G.state_dict().items())
                ed = max(float((S.D[k] - v).abs().max() / (v.abs().max() + 4e-4)) for k, v in D.state_dict().items())
                report.append("train %s %s step %d: %s | oracle-vs-ref weights G %.2e D %.2e" % (
                    tag, mode, step, " ".join("%s=%.6f" % kv for kv in ref.items()), ew, ed))
                assert ew < 1e-3 and ed < 1e-3, report[-1]
                arrs["raw%d" % step] = real_raw
                arrs["exp%d" % step] = real_exp
                arrs["losses%d" % step] = np.array([ref[k] for k in ("d_loss", "g_adv", "g_percep", "g_idt", "g_loss")])
                arrs["fake%d" % step] = fake_exp.detach() if cd_t == 8 else checksum(fake_exp)
                arrs["dgradnorm%d" % step] = np.array([d_grad_norms[k] for k in sorted(d_grad_norms)])
                arrs["ggradnorm%d" % step] = np.array([g_grad_norms[k] for k in sorted(g_grad_norms)])
                if cd_t == 8:
                    for k, v in G.state_dict().items():
                        if (step == nsteps - 1 and mode == "default") or k in ("enc1.main.1.weight", "dec5.1.main.1.weight"):
                            arrs["G%d/%s" % (step, k)] = v.clone()
                    for k, v in D.state_dict().items():
                        if (step == nsteps - 1 and mode == "default") or k.startswith("d1.") or k.endswith(("_u", "_v")):
                            arrs["D%d/%s" % (step, k)] = v.clone()
                if True:
                    arrs["Gsum%d" % step] = np.array([checksum(v)[:3] for k, v in sorted(G.state_dict().items())])
                    arrs["Dsum%d" % step] = np.array([checksum(v)[:3] for k, v in sorted(D.state_dict().items())])
            if cd_t == 8 and mode == "default":
                arrs.update({"G_init/" + k: v for k, v in PG.items()})
                arrs.update({"D_init/" + k: v for k, v in PD.items()})
            arrs["init_seeds"] = np.array([41, 42])
            arrs["G_init_checksums"] = np.array([checksum(PG[k])[:3] for k in sorted(PG)])
            arrs["D_init_checksums"] = np.array([checksum(PD[k])[:3] for k in sorted(PD)])
            arrs["gnames"] = np.array(sorted(dict(G.named_parameters()).keys()))
            arrs["dnames"] = np.array(sorted(dict(D.named_parameters()).keys()))
            npz("train_%s_%s.npz" % (tag, mode), **arrs)

    print("\n".join(report))
    open(os.path.join(OUT, "REPORT.txt"), "w").write("\n".join(report) + "\n")


if __name__ == "__main__":
    main()
