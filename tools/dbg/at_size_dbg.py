import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from oracle import uegan_oracle as O
from uegan_amd import fused, losses, models, ops, trainer
from test_oracle_at_size import _images, elem_rel
dev = torch.device("cuda:0")
ops.set_compute_dtype(torch.float32)
def relmax(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))
which = sys.argv[1] if len(sys.argv) > 1 else "all"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
if which in ("d", "all"):
    PD = O.init_params(O.discriminator_param_shapes(32), 42, "default")
    exp, fake, raw = _images(1, S, 5), _images(1, S, 6), _images(1, S, 7)
    Pr = {k: (v.clone() if k.endswith(O.D_BUFFER_SUFFIXES) else v.clone().requires_grad_(True)) for k, v in PD.items()}
    rp = O.discriminator_forward(Pr, exp, True); fp = O.discriminator_forward(Pr, fake, True)
    loss_r = O.rahinge_loss(rp, fp, True)
    ip = O.discriminator_forward(Pr, raw, True)
    loss_r = loss_r + O.rahinge_loss(rp, ip, True)
    loss_r.backward()
    def fresh():
        D = models.Discriminator(32, "none", "LeakyReLU", True, "rahinge"); D.load_state_dict(PD); return D.to(dev).train()
    D1, D2 = fresh(), fresh()
    A = losses.GANLoss("rahinge")
    xs = [exp.to(dev), fake.to(dev), raw.to(dev)]
    pe, pf, pr = D1(xs[0]), D1(xs[1]), D1(xs[2])
    l1 = A(pe, pf, None, None, for_discriminator=True) + A(pe, pr, None, None, for_discriminator=True)
    l1.backward()
    l2 = fused.discriminator_loss(D2, xs, [(0, 1), (0, 2)], True)
    l2.backward()
    print("D losses oracle %.7f module %.7f fused %.7f" % (float(loss_r), float(l1), float(l2)))
    for (k, p), q in zip(D1.named_parameters(), D2.parameters()):
        print("%-28s module-vs-oracle relmax %.2e elem %.2e | fused-vs-oracle relmax %.2e elem %.2e | fused-vs-module %.2e" % (
            k, relmax(p.grad, Pr[k].grad), elem_rel(p.grad, Pr[k].grad), relmax(q.grad, Pr[k].grad), elem_rel(q.grad, Pr[k].grad), relmax(q.grad, p.grad)))
if which in ("vgg", "all"):
    V = O.make_vgg_weights(seed=1234, width_div=1)
    x, y = _images(1, S, 21), _images(1, S, 22)
    xr = x.clone().requires_grad_(True)
    lr = O.perceptual_loss(V, (xr + 1.) / 2., (y + 1.) / 2.)
    lr.backward()
    xr64 = x.double().clone().requires_grad_(True)
    V64 = {k: v.double() for k, v in V.items()}
    l64 = O.perceptual_loss(V64, (xr64 + 1.) / 2., (y.double() + 1.) / 2.)
    l64.backward()
    print("oracle f32 vs f64: loss %.3e  grad relmax %.2e elem %.2e" % (abs(float(lr) - float(l64)) / float(l64), relmax(xr.grad, xr64.grad), elem_rel(xr.grad, xr64.grad.float())))
    P = losses.PerceptualLoss(vgg_weights=V, width_div=1).to(dev)
    for fz in (True, False):
        P.fused = fz
        xd = x.to(dev).requires_grad_(True)
        l = P(xd, y.to(dev), input_range01=False)
        l.backward()
        g = xd.grad.cpu()
        print("fused=%s loss rel %.3e (vs f64 %.3e)  grad vs f32 oracle: relmax %.2e elem %.2e ; vs f64 oracle relmax %.2e elem %.2e cos %.8f" % (
            fz, abs(float(l) - float(lr)) / float(lr), abs(float(l) - float(l64)) / float(l64), relmax(g, xr.grad), elem_rel(g, xr.grad),
            relmax(g, xr64.grad), elem_rel(g, xr64.grad.float()),
            float((g.double() * xr64.grad).sum() / g.double().norm() / xr64.grad.norm())))
