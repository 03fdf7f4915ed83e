"""The reference's non-default flags: norm_fun BatchNorm | InstanceNorm and act_fun ReLU | Swish | SELU | none
for the conv blocks (models.py:88-101, 158-167, 249-281), the adversarial losses besides 'rahinge' (losses.py:312-392) and RMSprop
(trainer.py:339-342).  The default configuration never comes through here: its activations live in the conv epilogues and its
passes are fused (uegan_amd/fused.py); these variants run layer by layer -- conv kernel, then ONE normalise+activate kernel
(uegan_affine_act_*), with the coefficient arithmetic on the [B, C] statistics done in torch on the host side of the launch.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .ops import _chk, _dt, _p, _ptr_table, _stream, lib

ACT_SIGMOID, ACT_SWISH, ACT_SELU = 4, 5, 6
ACT_CODES = {"LeakyReLU": ops.ACT_LRELU, "ReLU": ops.ACT_RELU, "Swish": ACT_SWISH, "SELU": ACT_SELU, "none": ops.ACT_NONE}
EPILOGUE_ACTS = ("LeakyReLU", "ReLU", "none")       # the conv kernels' epilogues evaluate these themselves


# --------------------------------------------------------------------------------------------------------------------
# normalisation (+ activation)
# --------------------------------------------------------------------------------------------------------------------
class _NormAct(torch.autograd.Function):
    """y = act(norm(x)) on an NHWC (channel-padded) tensor.  kind: None (activation only) | 'in' | 'bn'; batch_stats: statistics of
    this input (training, or no running statistics) instead of the running ones.  gamma / beta: [C] parameters."""

    @staticmethod
    def forward(ctx, x, gamma, beta, kind, act, batch_stats, running_mean, running_var, momentum, eps):
        x = x.contiguous()
        B, H, W, Cp = x.shape
        HW = H * W
        dev = x.device
        st = _stream()
        _chk(x)
        scale = shift = mu = r = None
        if kind is not None:
            Cr = gamma.shape[0]
            if batch_stats:
                stats = torch.empty((2, B, Cp), dtype=torch.float32, device=dev)
                tmp = torch.empty((lib().uegan_reduce_workspace_floats(B, HW, Cp),), dtype=torch.float32, device=dev)
                L.check(lib().uegan_moments(_dt(x), _p(x), _p(stats[0]), _p(stats[1]), _p(tmp), B, HW, Cp, st))
                mean_bc, var_bc = stats[0], stats[1]
                if kind == "in":
                    mu, var = mean_bc, var_bc
                    n = HW
                    new_mean, new_var = mean_bc.mean(0), (var_bc * (n / max(n - 1, 1))).mean(0)      # F.instance_norm: per-instance update, averaged
                else:
                    mu_c = mean_bc.mean(0)
                    var_c = (var_bc + (mean_bc - mu_c) ** 2).mean(0)                                   # law of total variance: no cancellation
                    mu, var = mu_c.expand(B, Cp), var_c.expand(B, Cp)
                    n = B * HW
                    new_mean, new_var = mu_c, var_c * (n / max(n - 1, 1))
                if running_mean is not None and momentum is not None:
                    with torch.no_grad():
                        running_mean.mul_(1 - momentum).add_(new_mean[:Cr], alpha=momentum)
                        running_var.mul_(1 - momentum).add_(new_var[:Cr], alpha=momentum)
            else:
                pad = (0, Cp - Cr)
                mu = torch.nn.functional.pad(running_mean.float(), pad).expand(B, Cp)
                var = torch.nn.functional.pad(running_var.float(), pad, value=1.0).expand(B, Cp)
            r = torch.rsqrt(var + eps)
            g = torch.nn.functional.pad(gamma.detach().float(), (0, Cp - Cr))
            b = torch.nn.functional.pad(beta.detach().float(), (0, Cp - Cr))
            scale = (g * r).contiguous()
            shift = (b - mu * scale).contiguous()
        y = torch.empty_like(x)
        L.check(lib().uegan_affine_act_fwd(_dt(x), act, _p(x), _p(scale), _p(shift), _p(y), B, HW, Cp, st))
        ctx.kind, ctx.act, ctx.batch_stats = kind, act, batch_stats
        ctx.save_for_backward(x, scale, shift, mu, r, gamma)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, scale, shift, mu, r, gamma = ctx.saved_tensors
        gy = gy.contiguous()
        B, H, W, Cp = x.shape
        HW = H * W
        dev = x.device
        st = _stream()
        kind, act = ctx.kind, ctx.act
        ca = cb = cc = dgamma = dbeta = None
        if kind is not None:
            Cr = gamma.shape[0]
            sums = torch.empty((B, Cp, 2), dtype=torch.float32, device=dev)
            tmp = torch.empty((lib().uegan_reduce_workspace_floats(B, HW, Cp),), dtype=torch.float32, device=dev)
            L.check(lib().uegan_affine_act_bwd_sums(_dt(x), act, _p(gy), _p(x), _p(scale), _p(shift), _p(sums), _p(tmp), B, HW, Cp, st))
            s0, s1 = sums[..., 0], sums[..., 1]
            t = r * (s1 - mu * s0)                          # sum g * xhat per (b, c)
            dbeta, dgamma = s0.sum(0)[:Cr], t.sum(0)[:Cr]
            g = torch.nn.functional.pad(gamma.detach().float(), (0, Cp - Cr))
            ca = (g * r).contiguous()
            if ctx.batch_stats:
                if kind == "in":
                    m0, m1 = s0 / HW, t / HW
                else:
                    m0, m1 = (s0.sum(0) / (B * HW)).expand(B, Cp), (t.sum(0) / (B * HW)).expand(B, Cp)
                cb = (-ca * r * m1).contiguous()
                cc = (-ca * m0 + ca * r * mu * m1).contiguous()
        gx = torch.empty_like(x)
        L.check(lib().uegan_affine_act_bwd_apply(_dt(x), act, _p(gy), _p(x), _p(scale), _p(shift), _p(ca), _p(cb), _p(cc), _p(gx), B, HW, Cp, st))
        return gx, dgamma, dbeta, None, None, None, None, None, None, None


class _Norm2d(nn.Module):
    """nn.BatchNorm2d / nn.InstanceNorm2d with affine=True, track_running_stats=True (models.py:272-277): same parameters,
    buffers and state-dict keys; the arithmetic happens in NormAct together with the block's activation."""
    kind = None

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def forward(self, x):
        raise RuntimeError("applied through NormAct (with the block's activation), not called directly")


class BatchNorm2d(_Norm2d):
    kind = "bn"


class InstanceNorm2d(_Norm2d):
    kind = "in"


class ActTag(nn.Module):
    """parameter-free stand-in at the activation's position in `main` (keeps the reference's module indices)"""

    def __init__(self, name):
        super().__init__()
        if name not in ACT_CODES:
            raise NotImplementedError("activation function [%s] is not found" % name)
        self.name, self.code = name, ACT_CODES[name]

    def forward(self, x):
        return x


class NormAct(nn.Module):
    """norm (a _Norm2d or None) followed by the activation `act` (an ActTag), one kernel each way"""

    def __init__(self, norm, act):
        super().__init__()
        object.__setattr__(self, "_norm", norm)         # (owned by the block's `main`: not registered twice)
        self.code = act.code

    def forward(self, x):
        n = self._norm
        if n is None:
            return _NormAct.apply(x, None, None, None, self.code, False, None, None, None, 0.0)
        if n.training and n.kind == "bn":
            n.num_batches_tracked += 1          # (nn.InstanceNorm2d never advances its counter)
        return _NormAct.apply(x, n.weight, n.bias, n.kind, self.code, n.training, n.running_mean, n.running_var, n.momentum, n.eps)


# --------------------------------------------------------------------------------------------------------------------
# adversarial losses
# --------------------------------------------------------------------------------------------------------------------
class _Rals(torch.autograd.Function):
    """GANLoss('rals') over lists of prediction maps (losses.py:363-376, 393-409); returns shape [1]."""

    @staticmethod
    def forward(ctx, for_discriminator, nscales, *maps):
        reals = [m.contiguous() for m in maps[:nscales]]
        fakes = [m.contiguous() for m in maps[nscales:]]
        for a, b in zip(reals, fakes):
            if a.dtype != torch.float32 or b.dtype != torch.float32 or a.numel() != b.numel():
                raise RuntimeError("rals: maps must be float32 with matching sizes")
        dev = reals[0].device
        loss = torch.empty((1,), dtype=torch.float32, device=dev)
        tmp = torch.empty((lib().uegan_rahinge_workspace_floats(nscales),), dtype=torch.float32, device=dev)
        n = (C.c_int64 * nscales)(*[a.numel() for a in reals])
        _chk(*reals, *fakes)
        L.check(lib().uegan_rals_fwd(nscales, _ptr_table(reals), _ptr_table(fakes), n, 1 if for_discriminator else 0, _p(loss), _p(tmp), _stream()))
        ctx.for_d, ctx.nscales = for_discriminator, nscales
        ctx.save_for_backward(tmp, *reals, *fakes)
        return loss

    @staticmethod
    def backward(ctx, g):
        tmp = ctx.saved_tensors[0]
        ns = ctx.nscales
        reals, fakes = ctx.saved_tensors[1:1 + ns], ctx.saved_tensors[1 + ns:]
        g = g.contiguous().float()
        greal = [torch.empty_like(a) if ctx.needs_input_grad[2 + i] else None for i, a in enumerate(reals)]
        gfake = [torch.empty_like(b) if ctx.needs_input_grad[2 + ns + i] else None for i, b in enumerate(fakes)]
        n = (C.c_int64 * ns)(*[a.numel() for a in reals])
        L.check(lib().uegan_rals_bwd(ns, _ptr_table(reals), _ptr_table(fakes), n, 1 if ctx.for_d else 0, _p(tmp), _p(g), _ptr_table(greal),
                                     _ptr_table(gfake), _stream()))
        return (None, None) + tuple(greal) + tuple(gfake)


def rals(real_preds, fake_preds, for_discriminator):
    return _Rals.apply(bool(for_discriminator), len(real_preds), *real_preds, *fake_preds)


PRED_BCE, PRED_LS, PRED_HINGE_REAL, PRED_HINGE_FAKE, PRED_NEG_MEAN, PRED_POS_MEAN = range(6)


class _PredLoss(torch.autograd.Function):
    """sum over scales of mean(term(pred)): the non-relativistic modes of GANLoss.loss on ONE prediction list; returns shape [1]."""

    @staticmethod
    def forward(ctx, term, target, *maps):
        ps = [m.contiguous() for m in maps]
        if any(p.dtype != torch.float32 for p in ps):
            raise RuntimeError("prediction maps must be float32")
        ns = len(ps)
        dev = ps[0].device
        loss = torch.empty((1,), dtype=torch.float32, device=dev)
        tmp = torch.empty((lib().uegan_pred_loss_workspace_floats(ns),), dtype=torch.float32, device=dev)
        n = (C.c_int64 * ns)(*[p.numel() for p in ps])
        _chk(*ps)
        L.check(lib().uegan_pred_loss_fwd(term, float(target), ns, _ptr_table(ps), n, _p(loss), _p(tmp), _stream()))
        ctx.term, ctx.target = term, float(target)
        ctx.save_for_backward(*ps)
        return loss

    @staticmethod
    def backward(ctx, g):
        ps = ctx.saved_tensors
        ns = len(ps)
        g = g.contiguous().float()
        gs = [torch.empty_like(p) for p in ps]
        n = (C.c_int64 * ns)(*[p.numel() for p in ps])
        L.check(lib().uegan_pred_loss_bwd(ctx.term, ctx.target, ns, _ptr_table(list(ps)), n, _p(g), _ptr_table(gs), _stream()))
        return (None, None) + tuple(gs)


def pred_loss(preds, term, target=0.0):
    return _PredLoss.apply(int(term), float(target), *preds)


# --------------------------------------------------------------------------------------------------------------------
# RMSprop
# --------------------------------------------------------------------------------------------------------------------
class FusedRMSprop(ops.FusedAdamL2):
    """torch.optim.RMSprop(params, lr, alpha) as the reference constructs it (trainer.py:339-342: eps 1e-8, weight_decay 0, momentum 0,
    centered False), one launch over the flat gradient bucket; state_dict() speaks torch.optim.RMSprop's format."""

    def __init__(self, params, lr, alpha=0.99, eps=1e-8):
        super().__init__(params, lr, betas=(0.0, 0.0), eps=eps, weight_decay=0.0)
        self.alpha = alpha

    def step(self, grad_scale=1.0):
        self.step_count += 1
        for p, v in zip(self.params, self._views):
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                raise RuntimeError("FusedRMSprop: a parameter's .grad no longer aliases the flat bucket")
        L.check(lib().uegan_rmsprop_step(_p(self.desc_dev), len(self.params), self.max_n, self.lr, self.alpha, self.eps, grad_scale, _stream()))
        ops.invalidate_weight_caches(self.params)

    @property
    def param_groups(self):
        g = {"lr": self.lr, "momentum": 0, "alpha": self.alpha, "eps": self.eps, "centered": False, "weight_decay": 0,
             "params": list(range(len(self.params)))}
        if self.initial_lr is not None:
            g["initial_lr"] = self.initial_lr
        return [g]

    def state_dict(self):
        self._flush()
        state = {}
        if self.step_count > 0:
            for i, (p, off) in enumerate(zip(self.params, self._offsets)):
                state[i] = {"step": self.step_count, "square_avg": self.v[off:off + p.numel()].view_as(p).clone()}
        return {"state": state, "param_groups": self.param_groups}

    def load_state_dict(self, sd):
        self._flush()
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.params):
            raise ValueError("loaded state dict has a different number of parameter groups / parameters")
        g = groups[0]
        if g.get("momentum", 0) or g.get("centered", False) or g.get("weight_decay", 0):
            raise NotImplementedError("RMSprop with momentum / centered / weight_decay is not what the reference constructs (trainer.py:341-342)")
        self.lr, self.alpha, self.eps = float(g["lr"]), float(g["alpha"]), float(g["eps"])
        if "initial_lr" in g:
            self.initial_lr = float(g["initial_lr"])
        self.v.zero_()
        steps = set()
        for key, st in sd["state"].items():
            i = int(key)
            p, off = self.params[i], self._offsets[i]
            if tuple(st["square_avg"].shape) != tuple(p.shape):
                raise ValueError("optimizer state %d has shape %s, parameter has %s" % (i, tuple(st["square_avg"].shape), tuple(p.shape)))
            self.v[off:off + p.numel()].view_as(p).copy_(st["square_avg"])
            steps.add(int(float(st["step"])))
        if len(sd["state"]) not in (0, len(self.params)) or len(steps) > 1:
            raise ValueError("FusedRMSprop keeps ONE step counter: every parameter must carry the same `step` (got %s)" % sorted(steps))
        self.step_count = steps.pop() if steps else 0
