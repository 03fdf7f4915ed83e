import sys, random; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from uegan_amd import models, losses, trainer, ops
from oracle import uegan_oracle as O
dev=torch.device('cuda:0')
z=np.load('tests/golden/train_cd8_orthogonal.npz'); zl=np.load('tests/golden/losses.npz')
V={k[5:]:torch.from_numpy(zl[k]) for k in zl.files if k.startswith('vgg8/')}
PG=O.init_params(O.generator_param_shapes(8),41,'orthogonal'); PD=O.init_params(O.discriminator_param_shapes(8),42,'orthogonal')
G=models.Generator(8,'none','LeakyReLU',False); D=models.Discriminator(8,'none','LeakyReLU',True,'rahinge')
G.load_state_dict(PG); D.load_state_dict(PD)
T=trainer.Trainer(G.to(dev),D.to(dev),losses.PerceptualLoss(vgg_weights=V,width_div=8).to(dev),pool_size=3,rng=random.Random(1990))
S=O.TrainState({k:v.clone() for k,v in PG.items()},{k:v.clone() for k,v in PD.items()},V,pool_size=3,rng=random.Random(1990))
for step in range(2):
    raw=torch.from_numpy(z['raw%d'%step]); exp=torch.from_numpy(z['exp%d'%step])
    T.train_step(raw.to(dev),exp.to(dev)); o=O.train_step(S,raw,exp,return_grads=True)
    print("step",step,T.loss_items(), {k:o[k] for k in ('d_loss','g_adv','g_percep','g_idt')})
    names=[n for n,p in G.named_parameters()]
    off=0
    for n,p in G.named_parameters():
        g=T.g_optimizer.flat_grad[off:off+p.numel()].view_as(p).cpu(); off+=p.numel()
        og=o['g_grads'][n]
        if n in ('dec5.1.main.1.weight','dec5.0.main.1.weight','dec4.main.1.weight','enc1.main.1.weight','dec5.1.main.1.bias'):
            print("  %-24s |g| ours %.3e oracle %.3e maxdiff %.3e  wd*|w| %.3e  wdiff %.3e"%(n, g.abs().max(), og.abs().max(), (g-og).abs().max(), 1e-4*p.abs().max().item(), (p.detach().cpu()-S.G[n]).abs().max()))
