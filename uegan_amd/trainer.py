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

from . import fused, ops, variants
from .losses import GANLoss, MultiscaleRecLoss, PerceptualLoss


class ImagePool:
    """History buffer of generated images (utils.py:23-50) as a device-resident ring: `pool_size` image slots allocated once,
    one gather (slots / batch -> returned batch) and one scatter (batch -> slots) per query (ops.copy_images), no per-image
    tensors.  The slot bookkeeping runs on the host and draws from `rng` exactly like the reference -- `uniform(0, 1)` per
    image once the pool is full, then `randint(0, pool_size - 1)` when it exceeds 0.5 -- so a seeded run returns the same images."""

    def __init__(self, pool_size, rng=random):
        self.pool_size = pool_size
        self.rng = rng
        self.num_imgs = 0
        self.slots = None            # [pool_size, C, H, W] fp32, allocated on the first query

    def query(self, images):
        if self.pool_size == 0:
            return images
        images = images.detach()
        if images.dtype != torch.float32:
            raise TypeError("ImagePool holds float32 images")
        images = images.contiguous()
        if self.slots is None:
            self.slots = torch.empty((self.pool_size,) + tuple(images.shape[1:]), dtype=images.dtype, device=images.device)
        elif tuple(self.slots.shape[1:]) != tuple(images.shape[1:]) or self.slots.device != images.device:
            raise RuntimeError("ImagePool: image shape/device changed (%s on %s, pool holds %s on %s)"
                               % (tuple(images.shape[1:]), images.device, tuple(self.slots.shape[1:]), self.slots.device))
        # Sequential semantics of the reference loop, resolved to indices: `written` maps a slot to the batch image stored into
        # it EARLIER in this query (a later image of the same batch that draws the slot gets that image, not the old content).
        written = {}
        src = []                     # per returned image: slot index, or ~i for image i of this batch
        for i in range(images.shape[0]):
            if self.num_imgs < self.pool_size:
                written[self.num_imgs] = i
                self.num_imgs += 1
                src.append(~i)
            elif self.rng.uniform(0, 1) > 0.5:
                k = self.rng.randint(0, self.pool_size - 1)
                src.append(~written[k] if k in written else k)
                written[k] = i
            else:
                src.append(~i)
        if all(s == ~i for i, s in enumerate(src)):
            out = images             # nothing came from the pool (values equal to the reference's torch.cat copy)
        else:
            out = torch.empty_like(images)
            ops.copy_images(out, self.slots, images, list(range(len(src))), src)      # reads the slots BEFORE the scatter below
        if written:
            ks = sorted(written)
            ops.copy_images(self.slots, self.slots, images, ks, [~written[k] for k in ks])
        return out


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
    """Flat fp32 gradient bucket of one optimizer, all-reduced once per optimizer step in CHUNKS (RCCL `nccl` backend on GPUs, gloo
    in CPU tests): contiguous parameter ranges of the bucket, each started as soon as the last of its parameters' gradients has
    been written -- while the rest of the backward sweep is still running.  The weight-gradient kernels write straight into the
    bucket (ops.GradSink) and tell the bucket (`notify`); `start()` launches whatever is left, `finish()` waits for everything.

    chunk_bounds: parameter indices (into optimizer.params) at which a new chunk begins; [] = one chunk.  Early starts need every
    parameter to be written exactly once per backward sweep (Trainer(fused_passes=True)); otherwise chunks go out at start()."""

    def __init__(self, optimizer_or_flat, group=None, chunk_bounds=(), early=True):
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.works = []
        self.early = early
        self.timing = None               # a list: finish() records event pairs around its waits (exposed_wait_ms)
        if torch.is_tensor(optimizer_or_flat):            # a bare flat tensor: one chunk, no notifications
            self.flat, self.chunks, self.param_chunk, self.pending0 = optimizer_or_flat, [(0, optimizer_or_flat.numel())], [], [0]
        else:
            opt = optimizer_or_flat
            self.flat = opt.flat_grad
            n = len(opt.params)
            cuts = [0] + sorted(b for b in set(chunk_bounds) if 0 < b < n) + [n]
            offs = list(opt._offsets) + [self.flat.numel()]
            self.chunks = [(offs[a], offs[b]) for a, b in zip(cuts[:-1], cuts[1:])]
            self.param_chunk = [0] * n
            self.pending0 = [0] * len(self.chunks)
            for ci, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
                for i in range(a, b):
                    self.param_chunk[i] = ci
                    self.pending0[ci] += 1
            if self.world > 1:
                for i, p in enumerate(opt.params):
                    p._uegan_sink.owner, p._uegan_sink.index = self, i
        self.arm()

    def arm(self):
        """call after the optimizer's zero_grad(): a new backward sweep begins"""
        self.pending = list(self.pending0)
        self.started = [False] * len(self.chunks)
        self.launch_log = []             # (chunk index, parameters of OTHER chunks still unwritten at that moment) in launch order: > 0 = went out
                                         # under the rest of the backward sweep (tests/test_dist.py asserts the order and the early starts)

    def _launch(self, ci):
        a, b = self.chunks[ci]
        self.started[ci] = True
        self.launch_log.append((ci, sum(self.pending)))
        if b > a:
            self.works.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def notify(self, param_index):
        if self.world == 1 or not self.early:
            return
        ci = self.param_chunk[param_index]
        self.pending[ci] -= 1
        if self.pending[ci] == 0 and not self.started[ci]:
            self._launch(ci)

    def touched_again(self, param_index):
        """a later contribution to a parameter whose first one was already reported.  Early starts rely on 'every parameter is written
        exactly once per backward sweep': a second write after the chunk went out would race with the all-reduce in flight on the same
        bucket range and silently corrupt the gradients -- refuse instead."""
        if self.world > 1 and self.early and self.started[self.param_chunk[param_index]]:
            raise RuntimeError("GradBucket: parameter %d received a second gradient contribution after its all-reduce chunk was launched; "
                               "build the bucket with early=False when parameters are used more than once per backward sweep" % param_index)

    def start(self):
        if self.world > 1:
            for ci in range(len(self.chunks)):
                if not self.started[ci]:
                    self._launch(ci)

    def finish(self):
        """wait for every chunk in flight (for RCCL: the CURRENT stream waits for ProcessGroupNCCL's stream, the host does not block).
        With `self.timing` a list, an event pair on the current stream brackets the waits: its elapsed time is the all-reduce tail that
        nothing on this stream covered (bench.py --gpus N reports the sum per step)."""
        timed = self.timing is not None and self.works and self.flat.is_cuda
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for w in self.works:
            w.wait()
        if timed:
            e1.record()
            self.timing.append((e0, e1))
        self.works = []
        return 1.0 / self.world

    def exposed_wait_ms(self):
        """sum over the recorded finish() calls (synchronise the device first); clears the record"""
        ms = sum(a.elapsed_time(b) for a, b in (self.timing or []))
        if self.timing is not None:
            self.timing = []
        return ms


def _chunk_bounds(module, optimizer, starts):
    """indices (into optimizer.params) of the first parameter of each named sub-module in `starts`"""
    ids = {id(p): i for i, p in enumerate(optimizer.params)}
    out = []
    for name in starts:
        ps = list(getattr(module, name).parameters())
        if ps and id(ps[0]) in ids:
            out.append(ids[id(ps[0])])
    return out


def lambda_rule(epoch, lr_num_epochs_decay=50, lr_decay_ratio=50):
    """trainer.py:347-349"""
    return 1.0 - max(0, epoch + 1 - lr_num_epochs_decay) / lr_decay_ratio


class LambdaLR:
    """torch.optim.lr_scheduler.LambdaLR as the reference drives it (trainer.py:344-351 construction, :131-134
    `step(epoch=current_epoch)`): lr = base_lr * lr_lambda(epoch).  Construction performs torch's initial step (epoch 0).
    state_dict() carries torch's keys (`lr_lambdas` saved as [None] because the rule is a plain function)."""

    def __init__(self, optimizer, lr_lambda=lambda_rule, before_change=None):
        """before_change: called before the optimizer's lr is rewritten (the Trainer passes its sync(): a generator update left
        pending by the data-parallel overlap must be applied at the learning rate of the epoch it belongs to)"""
        self.optimizer, self.lr_lambda, self.before_change = optimizer, lr_lambda, before_change
        if optimizer.initial_lr is None:
            optimizer.initial_lr = optimizer.lr
        self.base_lrs = [optimizer.initial_lr]
        self.last_epoch = 0
        self._step_count = 1
        self._apply()

    def _apply(self):
        if self.before_change is not None:
            self.before_change()
        self._last_lr = [b * self.lr_lambda(self.last_epoch) for b in self.base_lrs]
        self.optimizer.lr = self._last_lr[0]

    def step(self, epoch=None):
        self._step_count += 1
        self.last_epoch = self.last_epoch + 1 if epoch is None else epoch
        self._apply()

    def get_last_lr(self):
        return list(self._last_lr)

    def state_dict(self):
        return {"base_lrs": list(self.base_lrs), "last_epoch": self.last_epoch, "_step_count": self._step_count,
                "_get_lr_called_within_step": False, "_last_lr": list(self._last_lr), "lr_lambdas": [None]}

    def load_state_dict(self, sd):
        self.base_lrs = [float(b) for b in sd["base_lrs"]]
        self.last_epoch = int(sd["last_epoch"])
        self._step_count = int(sd.get("_step_count", self.last_epoch + 1))
        self._last_lr = [float(v) for v in sd.get("_last_lr", [b * self.lr_lambda(self.last_epoch) for b in self.base_lrs])]
        # (like torch: loading does not touch the optimizer's lr -- that comes from the optimizer's own state dict)


class Trainer:
    def __init__(self, G, D, percep=None, pool_size=50, g_lr=1e-4, d_lr=4e-4, beta1=0.5, beta2=0.999, lambda_adv=0.1, lambda_percep=1.0,
                 lambda_idt=0.1, adv_input=True, group=None, rng=random, broadcast_init=True, fused_passes=True, adv_loss_type="rahinge",
                 optimizer_type="adam", alpha=0.9, defer_g_update=None, overlap=True, early_taps=False, loss_scale=None, early_sn=False):
        """fused_passes: run the repeated network applications of a step as single batched passes (uegan_amd/fused.py: one
        generator pass for :85 + :112, one discriminator pass per optimizer step with the loss fused behind it, one VGG pass for
        both fidelity-loss images).  False: one module call per reference line, exactly as trainer.py:85-119 is written -- the
        same arithmetic through the drop-in module API (the two settings are compared in tests/test_fused.py)."""
        self.G, self.D = G, D
        # loss_scale: both backward sweeps start from loss * loss_scale and the optimizer kernels divide it out again (folded into their
        # grad_scale like 1/world).  Needed by the float16 storage mode only: fp16 keeps 11 significant bits but only 5 exponent bits, and
        # the gradients of a mean over ~10^7 pixels (1e-9 .. 1e-3) sit below its normal range; every backward op is linear in the incoming
        # gradient, so the scale is exact up to rounding.  Default: 2^14 in float16 mode (measured: DESIGN.md section 4), 1 otherwise.
        # loss_scale="dynamic": the usual guard for long fp16 runs -- after each backward sweep the flat gradient bucket is tested for inf / nan
        # (one tiny reduction + ONE host read per sweep); an overflowing sweep's optimizer step is skipped and the scale halved, 1000 clean
        # steps in a row double it (up to 2^24).  Ranks agree through a 1-element all-reduce.  A fixed scale never syncs the host.
        self.dynamic_scale = isinstance(loss_scale, str)
        if self.dynamic_scale and loss_scale != "dynamic":
            raise ValueError("loss_scale: a number, None or 'dynamic'")
        self.loss_scale = (65536.0 if self.dynamic_scale else float(loss_scale)) if loss_scale is not None else \
            (16384.0 if ops.get_compute_dtype() == torch.float16 else 1.0)
        self.scale_growth_interval, self._clean_steps, self.skipped_steps = 1000, 0, 0
        self._scale_given = loss_scale is not None       # (train_step refuses float16 storage with the implicit scale 1 of another dtype)
        self._g_pending_scale = self.loss_scale
        # the fused passes batch several applications of one network (exact without batch statistics) and fuse the 'rahinge' loss
        # behind the discriminator: non-default flags run one module call per reference line instead
        default_flags = getattr(G, "default_flags", True) and getattr(D, "default_flags", True) and adv_loss_type == "rahinge"
        self.fused_passes = fused_passes = fused_passes and default_flags
        self.criterionPercep = percep if percep is not None else PerceptualLoss().to(next(G.parameters()).device)
        self.criterionIdt = MultiscaleRecLoss(scale=3, rec_loss_type="l1", multiscale=True)
        self.criterionGAN = GANLoss(adv_loss_type)                                        # trainer.py:44
        self.lambda_adv, self.lambda_percep, self.lambda_idt, self.adv_input = lambda_adv, lambda_percep, lambda_idt, adv_input
        self.g_lr0, self.d_lr0 = g_lr, d_lr
        self.group = group
        distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        if distributed and broadcast_init:
            src = dist.get_global_rank(group, 0) if group is not None else 0           # (group rank 0: a sub-group need not hold global rank 0)
            for t in list(G.parameters()) + list(D.parameters()) + list(D.buffers()):
                dist.broadcast(t.data, src=src, group=group)
            ops.invalidate_weight_caches()
        # Data parallel: the generator's last all-reduce chunks and its optimizer step are left PENDING at the end of train_step and
        # applied at the start of the next one, after that step's G-independent input work (NCHW -> NHWC of both image sets) has been
        # queued: RCCL finishes the reduction on its own stream under it (trainer.py:85,118).  Nothing reads a stale
        # G: sync() runs before G's next use in train_step, before any direct G(x) call and before state_dict() (hooks below).
        # overlap: the D-independent generator losses on a second stream beside the discriminator update (train_step); False: one stream
        self.overlap = bool(overlap) and next(G.parameters()).is_cuda
        # early_taps: real_raw's VGG taps computed at the start of the step beside the generator's forward (train_step): bit-identical, but two
        # VGG passes of B instead of one of 2B and a slower generator forward -- measured 427 / 405 vs 432 / 431 img/s, so off unless asked for
        self.early_taps = bool(early_taps)
        # early_sn: the D update's spectral-norm rounds at the start of the step on the second stream (train_step).  Same-box A/B (bench.py --early-sn):
        # 32.75 vs 32.93 ms when the host runs ahead, 33.23 vs 33.01 with the per-step loss readback (more host work in front of the step's first
        # launches, where the GPU is waiting for the host) -- off unless asked for
        self.early_sn = bool(early_sn)
        self._side = None
        self.defer_g_update = distributed if defer_g_update is None else bool(defer_g_update)
        self._g_pending = False
        G.register_forward_pre_hook(lambda m, a: self.sync())
        G.register_state_dict_pre_hook(lambda m, prefix, keep_vars: self.sync())
        if optimizer_type == "adam":                                                      # trainer.py:335-338
            self.g_optimizer = ops.FusedAdamL2(G.parameters(), g_lr, (beta1, beta2), 1e-8, 1e-4)
            self.d_optimizer = ops.FusedAdamL2(D.parameters(), d_lr, (beta1, beta2), 1e-8, 1e-4)
        elif optimizer_type == "rmsprop":                                                 # :339-342
            self.g_optimizer = variants.FusedRMSprop(G.parameters(), g_lr, alpha)
            self.d_optimizer = variants.FusedRMSprop(D.parameters(), d_lr, alpha)
        else:
            raise NotImplementedError("=== Optimizer [{}] is not found ===".format(optimizer_type))
        self.g_optimizer.before_access = self.sync      # state_dict() / load_state_dict() / `lr = ...` from outside first apply a pending update
        self.lr_scheduler_g = LambdaLR(self.g_optimizer, lambda_rule, before_change=self.sync)      # trainer.py:344-351
        self.lr_scheduler_d = LambdaLR(self.d_optimizer, lambda_rule)
        ops.invalidate_weight_caches()      # weights may have been (re-)initialised through `.data` since the last forward
        # All-reduce chunks follow the order in which the backward sweeps finish parameter groups (parameters are laid out in
        # registration order: G = enc1-5 | upsample1-4 | dec1-4, dec5 | ga5-1; D = d1, d1_pred, ..., d5, d5_pred; the sweeps run
        # decoder -> attention -> encoder and d5 -> d1): D's d5 (70 % of its bytes) goes out first, G's decoder / attention /
        # upsample chunks go out while the encoder's gradients are still being computed.  SURVEY.md 5(ii), 8e.
        self.g_bucket = GradBucket(self.g_optimizer, group, _chunk_bounds(G, self.g_optimizer, ("enc5", "upsample1", "dec1", "ga5")), early=fused_passes)
        self.d_bucket = GradBucket(self.d_optimizer, group, _chunk_bounds(D, self.d_optimizer, ("d4", "d5")), early=fused_passes)
        self.fake_exp_pool = ImagePool(pool_size, rng)
        self.losses = {}
        self._sn_pre = self._sn_done = None
        # modules whose forward depends on .training (spectral norm's power iteration, batch statistics): train_step's fast path around nn.Module.train()
        from .models import SpectralNormConv2d
        self._mode_modules = [[m for m in net.modules() if isinstance(m, (SpectralNormConv2d, variants._Norm2d)) or type(m).__name__.startswith("BatchNorm")]
                              for net in (G, D)]

    def set_epoch(self, epoch):
        """trainer.py:131-134: `lr_scheduler_{g,d}.step(epoch=current_epoch)` at the first step of every epoch.  A generator update
        left pending by the previous step (data parallel) belongs to the OLD epoch: it is applied first, at the old learning rate."""
        self.sync()
        self.lr_scheduler_g.step(epoch=epoch)
        self.lr_scheduler_d.step(epoch=epoch)

    # ---- the reference's checkpoint dict (trainer.py:186-208 save, :402-423 resume; tester.py:133-146 reads G_net)
    def checkpoint(self, epoch):
        self.sync()
        ck = {"G_net": self.G.state_dict(), "D_net": self.D.state_dict(), "epoch": epoch,
              "g_optimizer": self.g_optimizer.state_dict(), "d_optimizer": self.d_optimizer.state_dict(),
              "lr_scheduler_g": self.lr_scheduler_g.state_dict(), "lr_scheduler_d": self.lr_scheduler_d.state_dict()}
        if self.dynamic_scale:
            # one OPTIONAL key beyond the reference's dict (trainer.py:199-207; its loaders ignore unknown keys): a resumed fp16 run continues
            # at the scale it had reached instead of restarting at 2^16
            ck["loss_scale_state"] = {"loss_scale": float(self.loss_scale), "clean_steps": int(self._clean_steps), "skipped_steps": int(self.skipped_steps)}
        return ck

    def _side_stream(self):
        if self._side is None:
            self._side = torch.cuda.Stream(device=next(self.G.parameters()).device)
        return self._side

    def _sweep_ok(self, optimizer):
        """dynamic loss scale: True when this sweep's (all-reduced) gradients are finite on every rank; adjusts the scale otherwise"""
        if not self.dynamic_scale:
            return True
        ok = torch.isfinite(optimizer.flat_grad).all().to(torch.float32).reshape(1)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        if bool(ok.item()):
            return True
        self.loss_scale = max(self.loss_scale * 0.5, 1.0)
        self._clean_steps = 0
        self.skipped_steps += 1
        return False

    def sync(self):
        """apply the generator update left pending by the last train_step (data parallel; see __init__)"""
        if self._g_pending:
            self._g_pending = False
            world_scale = self.g_bucket.finish()
            if self._sweep_ok(self.g_optimizer):
                self.g_optimizer.step(world_scale / self._g_pending_scale)                # trainer.py:118
                if self.dynamic_scale:
                    self._clean_steps += 1
                    if self._clean_steps >= self.scale_growth_interval:
                        self.loss_scale, self._clean_steps = min(self.loss_scale * 2.0, 2.0 ** 24), 0

    def save_checkpoint(self, path, epoch):
        """Not a collective: the replicas are bit-identical (all-reduced gradients, deterministic power iteration -- uegan_specnorm_multi
        has a fixed summation order, tests/test_dist.py asserts torch.equal on u / v), so group rank 0 writes and every other rank
        returns at once; calling it on rank 0 only (the usual `if rank == 0: save`) is fine."""
        self.sync()
        distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1
        if distributed and dist.get_rank(self.group) != 0:
            return
        torch.save(self.checkpoint(epoch), path)

    def load_checkpoint(self, path_or_dict, map_location=None):
        ck = path_or_dict if isinstance(path_or_dict, dict) else torch.load(path_or_dict, map_location=map_location, weights_only=True)
        self.sync()
        self.G.load_state_dict(ck["G_net"])
        self.D.load_state_dict(ck["D_net"])
        self.g_optimizer.load_state_dict(ck["g_optimizer"])
        self.d_optimizer.load_state_dict(ck["d_optimizer"])
        self.lr_scheduler_g.load_state_dict(ck["lr_scheduler_g"])
        self.lr_scheduler_d.load_state_dict(ck["lr_scheduler_d"])
        st = ck.get("loss_scale_state")
        if st is not None and self.dynamic_scale:
            self.loss_scale, self._clean_steps = float(st["loss_scale"]), int(st.get("clean_steps", 0))
            self.skipped_steps = int(st.get("skipped_steps", 0))
            self._g_pending_scale = self.loss_scale
        ops.invalidate_weight_caches()
        return ck.get("epoch")

    def train_step(self, real_raw, real_exp):
        """One iteration of trainer.py:77-119. Returns device scalars (no host sync)."""
        G, D = self.G, self.D
        if ops.get_compute_dtype() == torch.float16 and self.loss_scale == 1.0 and not self._scale_given:
            # the default scale was chosen from the compute dtype at construction: float16 selected afterwards would train with raw gradients,
            # most of which sit below fp16's normal range (the generator's gradient loses 11 % of its norm)
            raise RuntimeError("Trainer was built before set_compute_dtype(torch.float16): float16 storage needs a loss scale -- construct the "
                               "Trainer after selecting the dtype, or pass loss_scale=2**14 / 'dynamic' (loss_scale=1.0 explicitly to insist)")
        # (nn.Module.train() walks the whole module tree: ~60 modules each, every step, on the host path between the previous step's loss readback and
        # this step's first launch -- the GPU is idle there.  trainer.py:77-78 calls it every step, so a submodule a caller put into eval() -- e.g. to
        # freeze a spectral-norm power iteration -- must be switched back: the few modules whose behaviour depends on the flag are checked)
        if not G.training or not all(m.training for m in self._mode_modules[0]):
            G.train()
        if not D.training or not all(m.training for m in self._mode_modules[1]):
            D.train()
        fz = self.fused_passes
        self.criterionPercep.fused = fz
        side = self._side_stream() if (fz and self.overlap) else None
        y_taps = None
        self._sn_pre = None
        if side is not None and self.early_sn:
            # the D update's spectral-norm rounds (20 small dependent launches that need nothing but D's weights) on the second stream NOW, beside the
            # generator's forward, instead of in front of the discriminator's first convolution
            sn_start = torch.cuda.Event()
            sn_start.record()
            with torch.cuda.stream(side):
                side.wait_event(sn_start)
                self._sn_pre = fused.discriminator_sn(D, 3 if self.adv_input else 2, keep_uv=True)
                self._sn_done = torch.cuda.Event()
                self._sn_done.record(side)
            # (allocated under the side stream, consumed by D's forward and backward on the training stream: tell the caching allocator, so that the blocks
            # are not handed to the side stream's later VGG passes while queued training-stream kernels still read them, whatever happens to self._sn_pre)
            for layer in self._sn_pre:
                for t in layer:
                    if t is not None:
                        t.record_stream(torch.cuda.current_stream())
        if side is not None and self.early_taps:
            # real_raw's VGG taps (:108's second argument: no gradient, no dependence on G) at the very start of the step, on the second
            # stream beside the generator's forward -- an MFMA-bound pass beside an HBM-bound one
            step_start = torch.cuda.Event()
            step_start.record()
            with torch.cuda.stream(side):
                side.wait_event(step_start)
                y_taps = self.criterionPercep.reference_taps(real_raw, input_range01=False)
        if fz:
            # this step's G-independent input work first, THEN the previous step's pending generator update (its all-reduce tail ran
            # on RCCL's stream meanwhile), then :85 and :112 in one generator pass (same weights: G is not updated before :118)
            xin = G.input_pair(real_raw, real_exp)
            self.sync()
            fake_exp, real_exp_idt = G.forward_pair(real_raw, real_exp, xin=xin)
        else:
            self.sync()
            fake_exp = G(real_raw)                                                        # :85
        fake_exp_store = self.fake_exp_pool.query(fake_exp)                               # :86

        # The D-independent part of the generator's loss -- both VGG19 passes and the identity loss (:108, :113) -- on a second stream
        # beside the discriminator update: D's kernels run on small maps with grids that leave most CUs idle, the VGG kernels fill them.
        # Autograd runs each node's backward on its forward's stream, so the two backward sweeps overlap the same way.
        # (measured, 16 x 512^2 bf16: 42.5 -> 40.4 ms/step; putting D on a high-priority stream instead, or the weight gradients on a
        # stream of their own, added nothing.)  Every tensor that crosses the streams lives until the end of the step, and the side stream
        # starts each step by waiting for this one, so the caching allocator never hands a block to one stream while the other uses it.
        self.g_optimizer.zero_grad()
        self.g_bucket.arm()
        if side is not None:
            fwd_done = torch.cuda.Event()
            fwd_done.record()
            with torch.cuda.stream(side):
                side.wait_event(fwd_done)
                percep = self.criterionPercep(fake_exp, real_raw, input_range01=False, y_taps=y_taps)    # :108
                idt = self.criterionIdt(real_exp_idt, real_exp)                           # :113
                side_done = torch.cuda.Event()
                side_done.record(side)
        d_loss, adv = self._d_update_and_adv(real_raw, real_exp, fake_exp, fake_exp_store, fz)
        if side is None:
            percep = self.criterionPercep(fake_exp, real_raw, input_range01=False)        # :108
            if not fz:
                real_exp_idt = G(real_exp)                                                # :112
            idt = self.criterionIdt(real_exp_idt, real_exp)                               # :113
        else:
            torch.cuda.current_stream().wait_event(side_done)
        if fz:
            # :104-115: g_loss = lambda_adv*adv + lambda_percep*percep + lambda_idt*idt (same order) in one tiny kernel
            g_loss, parts = ops.loss_sum([adv, percep, idt], [self.lambda_adv, self.lambda_percep, self.lambda_idt])
            g_adv_loss, g_percep_loss, g_idt_loss = parts[0], parts[1], parts[2]
        else:
            g_adv_loss = self.lambda_adv * adv                                            # :104
            g_percep_loss = self.lambda_percep * percep                                   # :108
            g_idt_loss = self.lambda_idt * idt                                            # :113
            g_loss = g_adv_loss + g_percep_loss + g_idt_loss                              # :106,110,115 (same sum order)
        self._g_pending_scale = self.loss_scale                                           # (the deferred step divides by THIS sweep's scale)
        (g_loss if self.loss_scale == 1.0 else g_loss * self.loss_scale).backward()       # :117
        self.g_bucket.start()
        self._g_pending = True
        self.losses = dict(d_loss=d_loss.detach(), g_adv=g_adv_loss.detach(), g_percep=g_percep_loss.detach(),
                           g_idt=g_idt_loss.detach(), g_loss=g_loss.detach())
        # the five logged scalars in one device vector: loss_items() is then ONE copy + ONE host sync.  The copy into pinned memory
        # and its event are queued HERE, in front of the generator's optimizer step: the losses do not depend on it, so loss_items() returns while
        # Adam and the weight repacking are still running and the host starts issuing the next step under them.
        self._loss_vec = ops.gather_scalars([self.losses[k] for k in self.LOSS_KEYS])
        if self._loss_vec.is_cuda:
            if getattr(self, "_loss_host", None) is None:
                self._loss_host = torch.empty((len(self.LOSS_KEYS),), dtype=torch.float32).pin_memory()
                self._loss_evt = torch.cuda.Event()
            self._loss_host.copy_(self._loss_vec, non_blocking=True)
            self._loss_evt.record()
        if not self.defer_g_update:
            self.sync()                                                                   # :118
        self.fake_exp, self.real_exp_idt = fake_exp.detach(), real_exp_idt.detach()
        return self.losses

    def _d_update_and_adv(self, real_raw, real_exp, fake_exp, fake_exp_store, fz):
        G, D = self.G, self.D
        # ---------------- update D (:89-98)
        self.d_optimizer.zero_grad()
        self.d_bucket.arm()
        if fz:
            # D(real_exp), D(fake_store), D(real_raw) (:90,91,94) as one batched pass; both GANLoss terms (:92,95) fused behind it
            groups = [real_exp, fake_exp_store.detach()] + ([real_raw] if self.adv_input else [])
            if self._sn_pre is not None:
                torch.cuda.current_stream().wait_event(self._sn_done)
            d_loss = fused.discriminator_loss(D, groups, [(0, 1), (0, 2)] if self.adv_input else [(0, 1)], True, sn=self._sn_pre)
        else:
            real_exp_preds = D(real_exp)                                                  # :90
            fake_exp_preds = D(fake_exp_store.detach())                                   # :91
            d_loss = self.criterionGAN(real_exp_preds, fake_exp_preds, None, None, for_discriminator=True)
            if self.adv_input:
                input_preds = D(real_raw)                                                 # :94
                d_loss = d_loss + self.criterionGAN(real_exp_preds, input_preds, None, None, for_discriminator=True)
        d_scale = self.loss_scale
        (d_loss if d_scale == 1.0 else d_loss * d_scale).backward()                       # :96
        self.d_bucket.start()            # (chunks not yet in flight) the D all-reduce runs while the D-independent G work is issued
        # the part of the adversarial pass (:102-103) that needs neither the reduced gradient nor the updated D -- the layout conversion of its
        # two image sets -- goes out BEFORE the wait for the all-reduce: the tail of D's last chunk (d1 - d3) is then covered on the training
        # stream too, not only by the second stream's VGG passes
        adv_x = fused.discriminator_input([real_exp, fake_exp]) if fz else None
        world_scale = self.d_bucket.finish()
        if self._sweep_ok(self.d_optimizer):
            self.d_optimizer.step(world_scale / d_scale)                                  # :97 (after the all-reduce)
        with _Frozen(D):
            if fz:
                adv = fused.discriminator_loss(D, [real_exp, fake_exp], [(0, 1)], False, x_nhwc=adv_x)  # :102-104 (updated D)
            else:
                real_exp_preds = D(real_exp)                                              # :102 (updated D)
                fake_exp_preds = D(fake_exp)                                              # :103
                adv = self.criterionGAN(real_exp_preds, fake_exp_preds, None, None, for_discriminator=False)
        return d_loss, adv

    LOSS_KEYS = ("d_loss", "g_adv", "g_percep", "g_idt", "g_loss")

    def loss_items(self):
        """Single end-of-step readback of the five logged scalars: one device-to-host copy of a 5-float vector, one host sync (the reference
        syncs five times, trainer.py:98-119)."""
        if getattr(self, "_loss_vec", None) is None:
            return {}
        if self._loss_vec.is_cuda and getattr(self, "_loss_host", None) is not None:
            self._loss_evt.synchronize()
            vals = self._loss_host.tolist()
        else:
            vals = self._loss_vec.tolist()
        return dict(zip(self.LOSS_KEYS, vals))
