import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from oracle import uegan_oracle as O
from test_oracle_at_size import _images, elem_rel
def relmax(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))
S = 256
PD = O.init_params(O.discriminator_param_shapes(32), 42, "default")
exp, fake, raw = _images(1, S, 5), _images(1, S, 6), _images(1, S, 7)
def run(dt):
    Pr = {k: (v.clone().to(dt) if k.endswith(O.D_BUFFER_SUFFIXES) else v.clone().to(dt).requires_grad_(True)) for k, v in PD.items()}
    rp = O.discriminator_forward(Pr, exp.to(dt), True); fp = O.discriminator_forward(Pr, fake.to(dt), True)
    l = O.rahinge_loss(rp, fp, True)
    ip = O.discriminator_forward(Pr, raw.to(dt), True)
    l = l + O.rahinge_loss(rp, ip, True)
    l.backward()
    return float(l), {k: v.grad for k, v in Pr.items() if v.grad is not None}, [p.detach() for p in rp]
l32, g32, p32 = run(torch.float32)
l64, g64, p64 = run(torch.float64)
print("loss", l32, l64)
for k in g32:
    print("%-26s f32-vs-f64 relmax %.2e elem %.2e" % (k, relmax(g32[k], g64[k]), elem_rel(g32[k], g64[k].float())))
for i, (a, b) in enumerate(zip(p32, p64)):
    print("pred", i, relmax(a, b), float(b.abs().max()), float(b.abs().mean()))
