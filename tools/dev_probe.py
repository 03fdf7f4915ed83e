import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from helpers import golden, tens, use_backend
from oracle import uegan_oracle as O
from uegan_amd import ops
import test_train_step as TT
dev = use_backend("gpu")
ops.set_compute_dtype(torch.float32)
z = golden("train_cd32_default.npz")
PG = O.init_params(O.generator_param_shapes(32), 41, "default")
PD = O.init_params(O.discriminator_param_shapes(32), 42, "default")
T, G, D = TT._build(32, PG, PD, dev)
for step in range(3):
    T.train_step(tens(z, "raw%d" % step, dev), tens(z, "exp%d" % step, dev))
    worst = (0, "")
    for net, key in ((G, "ggradnorm%d"), (D, "dgradnorm%d")):
        named = dict(net.named_parameters()); ref = z[key % step]
        for i, k in enumerate(sorted(named)):
            if k.endswith(TT.DEAD): continue
            n = float(named[k].grad.norm()); r = abs(n - ref[i]) / (ref[i] + 1e-12)
            if r > worst[0]: worst = (r, k)
    wsum = (0, "")
    for net, tag in ((G, "G"), (D, "D")):
        sd = net.state_dict(); ref = z["%ssum%d" % (tag, step)]
        for i, k in enumerate(sorted(sd.keys())):
            if k.endswith(TT.DEAD): continue
            t = sd[k].double().cpu(); scale = ref[i][1] + 1e-6
            r = max(abs(float(t.sum()) - ref[i][0]) / scale, abs(float(t.abs().sum()) - ref[i][1]) / scale)
            if r > wsum[0]: wsum = (r, k)
    got = T.loss_items()
    lr = max(abs(got[k] - r) / abs(r) for k, r in zip(TT.NAMES, z["losses%d" % step]))
    print("step", step, "gradnorm worst", worst, "weight-sum worst", wsum, "loss rel", lr)
