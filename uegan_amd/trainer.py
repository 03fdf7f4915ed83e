"""Step driver: the reference's training iteration (trainer.py:77-119 + optimizer setup 335-338 + ImagePool,
utils.py:23-50) over the MI355X-native modules, one process per GPU.

Data parallelism: every rank holds replicated weights / spectral-norm vectors and its own unpaired
batch and ImagePool.  Per step there are exactly two collectives -- an RCCL all-reduce (sum) of the flat fp32 D-gradient
bucket after `d_loss.backward()` and of the flat G-gradient bucket after `g_loss.backward()`; the 1/world scaling is
folded into the fused Adam kernel.  The D all-reduce + D Adam are overlapped with the G-side work that does not
depend on D (VGG(fake), VGG(raw), G(real_exp) + identity loss), as the dependency analysis of trainer.py:85-119 allows;
the order of the five D forwards (spectral-norm u/v advance per call) and the scalar sum adv + percep + idt are kept.
"""
import random

import torch
import torch.distributed as dist

from . import ops
from .losses import GANLoss, MultiscaleRecLoss, PerceptualLoss


class ImagePool:
    """History buffer of generated images (utils.py:23-50); same `random` call order as the reference."""

    def __init__(self, pool_size, rng=random):
        self.pool_size = pool_size
        self.rng = rng
        if self.pool_size > 0:
            self.num_imgs = 0
            self.images = []

    def query(self, images):
        if self.pool_size == 0:
            return images
        return_images = []
        for image in images:
            image = torch.unsqueeze(image.detach(), 0)
            if self.num_imgs < self.pool_size:
                self.num_imgs = self.num_imgs + 1
                self.images.append(image)
                return_images.append(image)
            else:
                p = self.rng.uniform(0, 1)
                if p > 0.5:
                    random_id = self.rng.randint(0, self.pool_size - 1)
                    tmp = self.images[random_id].clone()
                    self.images[random_id] = image
                    return_images.append(tmp)
                else:
                    return_images.append(image)
        return torch.cat(return_images, 0)


class _Frozen:
    """Temporarily mark parameters as not requiring grad (D during the G update: its weight gradients are zeroed at
    trainer.py:89 before ever being used, so they are dead work)."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]

    def __enter__(self):
        for p in self.params:
            p.requires_grad_(False)

    def __exit__(self, *a):
        for p in self.params:
            p.requires_grad_(True)


class GradBucket:
    """Flat fp32 gradient bucket all-reduced once per optimizer step (RCCL `nccl` backend on GPUs, gloo in CPU tests)."""

    def __init__(self, flat, group=None):
        self.flat = flat
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.work = None

    def start(self):
        if self.world > 1:
            self.work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        return 1.0 / self.world


def lambda_rule(epoch, lr_num_epochs_decay=50, lr_decay_ratio=50):
    """trainer.py:347-349"""
    return 1.0 - max(0, epoch + 1 - lr_num_epochs_decay) / lr_decay_ratio


class Trainer:
    def __init__(self, G, D, percep=None, pool_size=50, g_lr=1e-4, d_lr=4e-4, beta1=0.5, beta2=0.999, lambda_adv=0.1, lambda_percep=1.0,
                 lambda_idt=0.1, adv_input=True, group=None, rng=random, broadcast_init=True):
        self.G, self.D = G, D
        self.criterionPercep = percep if percep is not None else PerceptualLoss().to(next(G.parameters()).device)
        self.criterionIdt = MultiscaleRecLoss(scale=3, rec_loss_type="l1", multiscale=True)
        self.criterionGAN = GANLoss("rahinge")
        self.lambda_adv, self.lambda_percep, self.lambda_idt, self.adv_input = lambda_adv, lambda_percep, lambda_idt, adv_input
        self.g_lr0, self.d_lr0 = g_lr, d_lr
        self.group = group
        distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        if distributed and broadcast_init:
            for t in list(G.parameters()) + list(D.parameters()) + list(D.buffers()):
                dist.broadcast(t.data, src=0, group=group)
            ops.invalidate_weight_caches()
        self.g_optimizer = ops.FusedAdamL2(G.parameters(), g_lr, (beta1, beta2), 1e-8, 1e-4)
        self.d_optimizer = ops.FusedAdamL2(D.parameters(), d_lr, (beta1, beta2), 1e-8, 1e-4)
        self.g_bucket = GradBucket(self.g_optimizer.flat_grad, group)
        self.d_bucket = GradBucket(self.d_optimizer.flat_grad, group)
        self.fake_exp_pool = ImagePool(pool_size, rng)
        self.losses = {}

    def set_epoch(self, epoch):
        f = lambda_rule(epoch)
        self.g_optimizer.lr = self.g_lr0 * f
        self.d_optimizer.lr = self.d_lr0 * f

    def train_step(self, real_raw, real_exp):
        """One iteration of trainer.py:77-119. Returns device scalars (no host sync)."""
        G, D = self.G, self.D
        G.train()
        D.train()
        fake_exp = G(real_raw)                                                            # :85
        fake_exp_store = self.fake_exp_pool.query(fake_exp)                               # :86

        # ---------------- update D (:89-98)
        self.d_optimizer.zero_grad()
        real_exp_preds = D(real_exp)                                                      # :90
        fake_exp_preds = D(fake_exp_store.detach())                                       # :91
        d_loss = self.criterionGAN(real_exp_preds, fake_exp_preds, None, None, for_discriminator=True)
        if self.adv_input:
            input_preds = D(real_raw)                                                     # :94
            d_loss = d_loss + self.criterionGAN(real_exp_preds, input_preds, None, None, for_discriminator=True)
        d_loss.backward()                                                                 # :96
        self.d_bucket.start()            # RCCL all-reduce of the D bucket runs while the D-independent G work is issued

        # ---------------- update G (:101-119)
        self.g_optimizer.zero_grad()
        g_percep_loss = self.lambda_percep * self.criterionPercep(fake_exp, real_raw, input_range01=False)   # :108
        real_exp_idt = G(real_exp)                                                        # :112
        g_idt_loss = self.lambda_idt * self.criterionIdt(real_exp_idt, real_exp)          # :113

        self.d_optimizer.step(self.d_bucket.finish())                                     # :97 (after the all-reduce)
        with _Frozen(D):
            real_exp_preds = D(real_exp)                                                  # :102 (updated D)
            fake_exp_preds = D(fake_exp)                                                  # :103
        g_adv_loss = self.lambda_adv * self.criterionGAN(real_exp_preds, fake_exp_preds, None, None, for_discriminator=False)  # :104
        g_loss = g_adv_loss + g_percep_loss + g_idt_loss                                  # :106,110,115 (same sum order)
        g_loss.backward()                                                                 # :117
        self.g_bucket.start()
        self.g_optimizer.step(self.g_bucket.finish())                                     # :118
        self.losses = dict(d_loss=d_loss.detach(), g_adv=g_adv_loss.detach(), g_percep=g_percep_loss.detach(),
                           g_idt=g_idt_loss.detach(), g_loss=g_loss.detach())
        self.fake_exp, self.real_exp_idt = fake_exp.detach(), real_exp_idt.detach()
        return self.losses

    def loss_items(self):
        """Single end-of-step readback of the five logged scalars (the reference syncs five times, trainer.py:98-119)."""
        return {k: float(v.reshape(-1)[0]) for k, v in self.losses.items()}
