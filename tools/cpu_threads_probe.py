import sys, time, random, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import uegan_oracle as O
for nt in (8, 16, 32, 64):
    torch.set_num_threads(nt)
    PG = O.init_params(O.generator_param_shapes(32), 41, "default"); PD = O.init_params(O.discriminator_param_shapes(32), 42, "default")
    V = O.make_vgg_weights(seed=1234, width_div=1)
    St = O.TrainState(PG, PD, V, pool_size=50, rng=random.Random(1990))
    g = torch.Generator().manual_seed(1990)
    ts = []
    for it in range(2):
        raw = torch.rand(2, 3, 512, 512, generator=g) * 2 - 1; exp = torch.rand(2, 3, 512, 512, generator=g) * 2 - 1
        t = time.time(); O.train_step(St, raw, exp); ts.append(time.time() - t)
    print(nt, "threads:", ["%.1f s" % t for t in ts], flush=True)
