"""torch.autograd.Function wrappers over the C ABI of libuegan_hip.so.

PyTorch is plumbing here: device memory (caching allocator), the current HIP stream and the autograd graph.
Every arithmetic step of the hot path is a hand-written gfx950 kernel reached through ctypes (uegan_amd/_lib.py).
Activations inside the networks are explicit NHWC tensors ([B,H,W,C]) in the compute dtype (fp32 or bf16);
module boundaries keep the reference's NCHW fp32 contract (data_loader.py:79-81).
"""
import ctypes as C
import weakref

import torch

from . import _lib as L

ACT_NONE, ACT_LRELU, ACT_RELU, ACT_TANH = 0, 1, 2, 3
PAD_ZERO, PAD_REFLECT = 0, 1
IN_EPS = 1e-5
SN_EPS = 1e-12

_compute_dtype = torch.float32
_weight_epoch = [0]


def set_compute_dtype(dt):
    """Storage dtype of activations / packed weights inside G, D and VGG: torch.float32 (parity mode), torch.bfloat16 (throughput mode;
    fp32 accumulation, fp32 master weights / statistics / losses) or torch.float16 (the same bytes and MFMA rate with 11 instead of 8
    significant bits: libuegan_hip_f16.so, the same kernel sources with fp16 as the 16-bit storage format -- what the generator's O(1)
    activations want; training in it needs Trainer(loss_scale=...): fp16's exponent range does not hold raw gradients).
    Process-wide; do not switch between a forward and its backward."""
    global _compute_dtype
    if dt not in (torch.float32, torch.bfloat16, torch.float16):
        raise ValueError("compute dtype must be float32, bfloat16 or float16")
    if dt != torch.float32:
        L.use_half_format("fp16" if dt == torch.float16 else "bf16")
    _compute_dtype = dt      # (the packed-weight caches are keyed on the dtype: nothing to invalidate)


def get_compute_dtype():
    return _compute_dtype


_precise = [False]


def set_precise(flag):
    """`precise` mode of the 16-bit storage dtypes (meant for torch.float16): the generator's full-resolution tensors (the image, x1, the attention
    branch ga1, y4 * x1, dec5.0's result) travel as hi + lo PAIRS of 16-bit planes and the weights of the five thin layers as pairs too, the residual
    + clamp is formed from dec5.1's fp32 result (uegan_conv2d_fwd_ex) -- the enhanced pixels then sit inside north_star's 1e-3 of the fp32 reference
    (the backward pass is unchanged: it reads the hi planes).  No effect in float32 mode.  Process-wide."""
    _precise[0] = bool(flag)


# the generator's product (models.py:69) and residual + clamp (models.py:70-72) formed by the epilogues of dec4 / dec5.1 in the plain 16-bit modes
# (uegan_conv2d_fwd_ex); False: the separate elementwise kernels of round 5 (A/B: bench.py --no-fuse-epilogues).  The precise mode always fuses.
fuse_epilogues = [True]


def precise():
    """is the precise mode in force for the current compute dtype?"""
    return _precise[0] and _compute_dtype != torch.float32


def invalidate_weight_caches(params=None):
    """Call after weights were modified behind autograd's back (`.data` edits, in-place kernels): the packed bf16/fp32 copies the
    conv kernels read are re-made on next use.  With `params` only those tensors' copies are invalidated (what FusedAdamL2.step
    does for the tensors it updated -- the frozen VGG19 and the other network keep theirs); without, every cache in the process."""
    if params is None:
        _weight_epoch[0] += 1
    else:
        for p in params:
            p._uegan_epoch = getattr(p, "_uegan_epoch", 0) + 1


class GradSink:
    """Where a parameter's gradient lives inside an optimizer's flat fp32 bucket (FusedAdamL2): the weight-gradient kernels write
    (first touch after zero_grad) or accumulate (later touches) there directly and hand autograd `None`, so there is no
    per-parameter `grad += dw` kernel and no temporary."""
    __slots__ = ("view", "dirty", "owner", "index")

    def __init__(self, view):
        self.view, self.dirty = view, False
        self.owner, self.index = None, -1        # trainer.GradBucket: told when this parameter's gradient has been written

    def mark(self):
        """the gradient (or its first contribution) is in the bucket"""
        first = not self.dirty
        self.dirty = True
        if self.owner is not None:
            if first:
                self.owner.notify(self.index)
            else:
                self.owner.touched_again(self.index)      # (raises if this parameter's all-reduce chunk is already in flight)


def _sink_of(p):
    if p is None or not p.requires_grad:
        return None
    s = getattr(p, "_uegan_sink", None)
    if s is None or p.grad is None or p.grad.data_ptr() != s.view.data_ptr():
        return None
    return s


def _dt(t):
    if t.dtype == torch.float32:
        return 0
    if t.dtype == torch.bfloat16 or t.dtype == torch.float16:      # code 1 = "the 16-bit storage format" of the loaded build
        if (t.dtype == torch.float16) != (L.half_format() == "fp16"):
            raise TypeError("a %s tensor reached the %s build of the kernel library (uegan_amd.set_compute_dtype selects it)"
                            % (t.dtype, L.half_format()))
        return 1
    raise TypeError("unsupported dtype %s" % t.dtype)


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    if L.is_emulated():
        return None
    return torch.cuda.current_stream().cuda_stream


def _chk(*ts):
    emu = L.is_emulated()
    for t in ts:
        if t is None:
            continue
        if not emu and not t.is_cuda:
            raise RuntimeError("uegan_amd ops need CUDA/HIP tensors (there is no CPU path)")
        if not t.is_contiguous():
            raise RuntimeError("uegan_amd internal error: non-contiguous tensor reached a kernel")


def lib():
    return L.load()


def chunk_elems(dtype):
    """elements per 16-byte chunk: every NHWC tensor the kernels see has a channel count that is a multiple of this"""
    return 8 if dtype in (torch.bfloat16, torch.float16) else 4


def cpad(c, dtype):
    e = chunk_elems(dtype)
    return (c + e - 1) // e * e


# --------------------------------------------------------------------------------------------------------------------
# layout boundary
# --------------------------------------------------------------------------------------------------------------------
def _farr(vals):
    return None if vals is None else (C.c_float * len(vals))(*[float(v) for v in vals])


class _ToNHWC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype, a, b, pair=False):
        x = x.contiguous()
        if x.dtype != torch.float32:
            raise TypeError("module inputs must be float32 NCHW (data_loader.py:79-81)")
        B, Cc, H, W = x.shape
        Cp = cpad(Cc, dtype)                      # zero-padded to one 16-byte chunk (3 -> 8 bf16 / 4 fp32)
        y = torch.empty((B, H, W, Cp), dtype=dtype, device=x.device)
        _chk(x, y)
        fn = lib().uegan_nchw_to_nhwc_pair if pair else lib().uegan_nchw_to_nhwc      # (pair: the image's lo plane in its own spare channels)
        L.check(fn(_dt(y), _p(x), _p(y), B, Cc, Cp, H, W, _farr(a), _farr(b), _stream()))
        ctx.a, ctx.C = a, Cc
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        B, H, W, Cp = g.shape
        gx = torch.empty((B, ctx.C, H, W), dtype=torch.float32, device=g.device)
        L.check(lib().uegan_nhwc_to_nchw(_dt(g), _p(g), _p(gx), B, ctx.C, Cp, H, W, _farr(ctx.a), _stream()))
        return gx, None, None, None, None


class _ToNCHW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, channels):
        x = x.contiguous()
        B, H, W, Cp = x.shape
        Cc = Cp if channels is None else channels
        y = torch.empty((B, Cc, H, W), dtype=torch.float32, device=x.device)
        _chk(x, y)
        L.check(lib().uegan_nhwc_to_nchw(_dt(x), _p(x), _p(y), B, Cc, Cp, H, W, None, _stream()))
        ctx.dtype, ctx.Cp = x.dtype, Cp
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        B, Cc, H, W = g.shape
        gx = torch.empty((B, H, W, ctx.Cp), dtype=ctx.dtype, device=g.device)
        L.check(lib().uegan_nchw_to_nhwc(_dt(gx), _p(g), _p(gx), B, Cc, ctx.Cp, H, W, None, None, _stream()))
        return gx, None


class _ToNHWCPair(torch.autograd.Function):
    """two NCHW fp32 batches -> one NHWC tensor [Ba + Bb, H, W, Cp] (a batch concatenation that never exists in NCHW)"""

    @staticmethod
    def forward(ctx, xa, xb, dtype, pair=False):
        y = raw_to_nhwc([xa, xb], dtype, pair=pair)
        ctx.Ba, ctx.C = xa.shape[0], xa.shape[1]
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        ga = raw_to_nchw_grad(g[:ctx.Ba], ctx.C) if ctx.needs_input_grad[0] else None
        gb = raw_to_nchw_grad(g[ctx.Ba:], ctx.C) if ctx.needs_input_grad[1] else None
        return ga, gb, None, None


def to_nhwc(x, dtype=None, a=None, b=None, pair=False):
    """pair: channels [C, 2C) of the padded pixel receive the lo plane of the image (uegan_nchw_to_nhwc_pair; 16-bit dtypes)"""
    return _ToNHWC.apply(x, dtype or _compute_dtype, a, b, pair)


def to_nhwc_pair(xa, xb, dtype=None, pair=False):
    """two image sets -> one batch-concatenated NHWC tensor (`pair` as in to_nhwc: unrelated to the two sets)"""
    return _ToNHWCPair.apply(xa, xb, dtype or _compute_dtype, pair)


def to_nchw(x, channels=None):
    """NHWC (possibly channel-padded) -> NCHW fp32 keeping the first `channels` channels"""
    return _ToNCHW.apply(x, channels)


# --------------------------------------------------------------------------------------------------------------------
# convolution
# --------------------------------------------------------------------------------------------------------------------
class PackedWeight:
    """Cache of the two packed copies (OHWI for forward/wgrad, IHWO for dgrad) of one OIHW fp32 weight."""

    def __init__(self):
        self.key = None
        self.ohwi = None
        self.ihwo = None
        self.ohwi_lo = None       # pair=True: the lo part of the OHWI copy (uegan_pack_weights_pair)
        self.version = 0          # bumped whenever the packed buffers are rewritten IN PLACE (PackTable.repack)

    def get(self, w, dtype, cin_pad, cout_pad, key_src=None, cin_used=None, pair=False, dup=False):
        """cin_used: pack only the first cin_used input channels of w (a column slice of the master weight); pair: also the lo part of the forward copy
        (self.ohwi_lo); dup 1: input channels [Cin, 2 Cin) of the forward copies repeat [0, Cin) (the source carries its own lo plane there), dup 2: they hold
        the LO part of [0, Cin) (the weight pair inside one matrix, for a kernel that reads the source's channels twice: stride-2 forwards)"""
        src = w if key_src is None else key_src
        key = (src.data_ptr(), src._version, _weight_epoch[0], getattr(src, "_uegan_epoch", 0), dtype, tuple(w.shape), str(w.device), cin_pad,
               cout_pad, cin_used, bool(pair), int(dup))
        if key != self.key:
            wd = w.detach()
            if wd.dtype != torch.float32:
                raise TypeError("master weights must be float32")
            wd = wd.contiguous()
            co, ci_total, kh, kw = wd.shape
            ci = ci_total if cin_used is None else cin_used
            kp = lib().uegan_packed_k(kh * kw * cin_pad)
            kp2 = lib().uegan_packed_k(kh * kw * cout_pad)
            self.ohwi = torch.empty((cout_pad, kp), dtype=dtype, device=wd.device)
            self.ihwo = torch.empty((cin_pad, kp2), dtype=dtype, device=wd.device)
            self.ohwi_lo = torch.empty((cout_pad, kp), dtype=dtype, device=wd.device) if pair else None
            _chk(wd)
            L.check(lib().uegan_pack_weights_pair(_dt(self.ohwi), _p(wd), co, ci, ci_total, kh, kw, cout_pad, cin_pad, _p(self.ohwi),
                                                  _p(self.ihwo), _p(self.ohwi_lo), int(dup), _stream()))
            self.key = key
            # remembered on the master tensor: the optimizer that updates it re-packs all of its copies in one launch (repack_all)
            self.args = (wd.data_ptr(), co, ci, ci_total, kh, kw, cout_pad, cin_pad, kp, kp2, _p(self.ohwi_lo) or 0, int(dup))
            self.src = weakref.ref(src)
            packs = getattr(src, "_uegan_packs", None)
            if packs is None:
                packs = src._uegan_packs = []
            if not any(r() is self for r in packs):
                packs.append(weakref.ref(self))
        return self.ohwi, self.ihwo

    def ohwi_for(self, ihwo):
        """the forward (OHWI) pack that belongs to the data-gradient pack `ihwo` a graph node saved -- None when the packs have been re-made since"""
        return self.ohwi if self.ihwo is ihwo else None

    def _fresh_key(self):
        """the key get() would compute now for the arguments of the last pack (after an in-place optimizer step on the master)"""
        src = self.src()
        k = self.key
        return (src.data_ptr(), src._version, _weight_epoch[0], getattr(src, "_uegan_epoch", 0)) + k[4:]      # (dtype ... pair, dup unchanged)


class PackTable:
    """Device table for uegan_pack_weights_multi over the packed copies of a set of master weights; rebuilt when a copy was
    (re)allocated or a new one appeared."""

    def __init__(self):
        self.sig, self.table, self.total, self.n, self.dtype = None, None, 0, 0, None

    def repack(self, params):
        packs = []
        for p in params:
            for r in getattr(p, "_uegan_packs", ()):
                pw = r()
                if pw is not None and pw.key is not None and pw.src() is p and pw.ohwi.dtype == _compute_dtype and pw.args[0] == p.data_ptr():
                    packs.append(pw)
        if not packs:
            return
        sig = tuple((id(pw), pw.ohwi.data_ptr(), pw.ihwo.data_ptr()) + pw.args for pw in packs)
        if sig != self.sig:
            ents = (L.PackEntry * len(packs))()
            start = 0
            for e, pw in zip(ents, packs):
                wptr, co, ci, ci_total, kh, kw, cout_pad, cin_pad, kp, kp2, lo_ptr, dup = pw.args
                e.w_oihw, e.w_ohwi, e.w_ihwo, e.start = wptr, pw.ohwi.data_ptr(), pw.ihwo.data_ptr(), start
                e.w_ohwi_lo, e.flags = (lo_ptr or None), dup
                e.Cout, e.Cin, e.Cin_total, e.KH, e.KW, e.Cout_pad, e.Cin_pad, e.Kp, e.Kp2 = co, ci, ci_total, kh, kw, cout_pad, cin_pad, kp, kp2
                start += cout_pad * kp + cin_pad * kp2
            host = torch.frombuffer(bytearray(bytes(ents)), dtype=torch.uint8)
            self.table = host.to(packs[0].ohwi.device)
            self.sig, self.total, self.n, self.dtype = sig, start, len(packs), _dt(packs[0].ohwi)
        L.check(lib().uegan_pack_weights_multi(self.dtype, _p(self.table), self.n, self.total, _stream()))
        for pw in packs:
            pw.key = pw._fresh_key()
            pw.version += 1


class ConvCfg:
    """Static configuration of one conv layer.  `in_act` / `premasked` implement deferred activation gradients (an exact
    restructuring, include/uegan_hip.h uegan_conv2d_dgrad_act): `in_act` = the activation that produced this conv's input
    x1 -- the data gradient is multiplied by act'(x1) in the dgrad epilogue; `premasked` = every consumer of this conv's
    output does that for it, so backward() skips its own act_bwd pass.  Only a module that owns the whole chain may set them
    (losses.VGG19_relu with deferred_act_grad=True); the defaults are plain autograd semantics."""
    __slots__ = ("stride", "pad_mode", "act", "packed", "packed_il", "in_act", "premasked", "cin_used")

    def __init__(self, stride, pad_mode, act, cin_used=None):
        self.stride, self.pad_mode, self.act = stride, pad_mode, act
        self.packed = PackedWeight()
        self.packed_il = PackedWeight()     # stride-2 forwards on a weight pair: [hi | lo] per tap in one matrix (ConvExtras.pair_w)
        self.in_act, self.premasked = ACT_NONE, False
        self.cin_used = cin_used        # the conv uses only the first cin_used input channels of its weight tensor (models.GAM)


class SNCall:
    """Spectral-norm state of ONE forward call: sigma (device fp32 [2] = sigma, 1/sigma) and the u, v used."""
    __slots__ = ("sigma", "u", "v")

    def __init__(self, sigma, u, v):
        self.sigma, self.u, self.v = sigma, u, v


def _desc(x1, x2, weight, cfg):
    B, H, W, C1 = x1.shape
    C2 = 0 if x2 is None else x2.shape[3]
    co, ci_total, kh, kw = weight.shape
    ci = ci_total if cfg.cin_used is None else cfg.cin_used
    e = chunk_elems(x1.dtype)
    if C1 % e or C2 % e:
        raise RuntimeError("conv: NHWC tensors must carry channel counts padded to multiples of %d (got %d, %d)" % (e, C1, C2))
    if not (ci == C1 + C2 or (C2 == 0 and cpad(ci, x1.dtype) == C1)):
        raise RuntimeError("conv: weight expects %d input channels, got %d" % (ci, C1 + C2))
    pad = (kh - 1) // 2
    Ho = (H + 2 * pad - kh) // cfg.stride + 1
    Wo = (W + 2 * pad - kw) // cfg.stride + 1
    if cfg.pad_mode == PAD_REFLECT and (pad >= H or pad >= W):
        raise RuntimeError("Padding size should be less than the corresponding input dimension (pad %d, input %dx%d)" % (pad, H, W))
    return L.ConvDesc(_dt(x1), B, H, W, C1, C2, Ho, Wo, cpad(co, x1.dtype), kh, kw, cfg.stride, pad, cfg.pad_mode, cfg.act, ci, co, ci_total)


class ConvExtras:
    """Extras of ONE forward convolution (uegan_conv2d_fwd_ex): the request (constructor) and, after the call, what the kernel produced.
      x1_lo / x2_lo   lo planes of the sources (hi + lo pairs, see set_precise)
      pair_w          multiply by the weights as a hi + lo pair;  dup_cin: the source carries its own lo plane in channels [Cin, 2 Cin) (to_nhwc(pair=True))
      want_lo         -> y_lo: the lo plane of the result
      mul / mul_lo    -> prod (/ prod_lo when want_mul_lo): act(...) * (mul + mul_lo) formed from the fp32 result (models.py:69)
      res             (xa,) or (xa, xb): NCHW fp32 image sets -> res_out: clamp(act(...) + x, -1, 1) per set (models.py:70-72), from the fp32 result
    taken: did a kernel honour the request?  False: the plain convolution ran and every output field is None -- a caller that only asked for an
    epilogue (mul / res) then runs the separate kernel; a request that involves pairs raises instead (there is no plain equivalent)."""
    __slots__ = ("x1_lo", "x2_lo", "pair_w", "dup_cin", "want_lo", "mul", "mul_lo", "want_mul_lo", "res", "y_lo", "prod", "prod_lo", "res_out", "taken")

    def __init__(self, x1_lo=None, x2_lo=None, pair_w=False, dup_cin=False, want_lo=False, mul=None, mul_lo=None, want_mul_lo=False, res=None):
        self.x1_lo, self.x2_lo, self.pair_w, self.dup_cin, self.want_lo = x1_lo, x2_lo, pair_w, dup_cin, want_lo
        self.mul, self.mul_lo, self.want_mul_lo, self.res = mul, mul_lo, want_mul_lo, res
        self.y_lo = self.prod = self.prod_lo = self.res_out = None
        self.taken = False

    def needs_pairs(self):
        return self.x1_lo is not None or self.x2_lo is not None or self.pair_w or self.want_lo or self.want_mul_lo or self.mul_lo is not None


def _conv_fwd_ex(d, x1, x2, ohwi, ohwi_lo, biasc, scale, y, ex, stats, interleaved=False):
    """uegan_conv2d_fwd_ex for one layer: fills ex's output fields (and stats.value); False: nothing was launched.
    interleaved: ohwi is the [hi | lo] matrix of a stride-2 forward (ConvCfg.packed_il)"""
    dev, dt = x1.device, x1.dtype
    e = L.ConvEx()
    e.w_interleaved = 1 if interleaved else 0
    shape = (d.B, d.Ho, d.Wo, d.Cout)
    e.x1_lo, e.x2_lo, e.w_lo = _p(ex.x1_lo), _p(ex.x2_lo), _p(ohwi_lo)
    y_lo = torch.empty(shape, dtype=dt, device=dev) if ex.want_lo else None
    prod = torch.empty(shape, dtype=dt, device=dev) if ex.mul is not None else None
    prod_lo = torch.empty(shape, dtype=dt, device=dev) if (ex.mul is not None and ex.want_mul_lo) else None
    e.y_lo, e.mul, e.mul_lo, e.y_mul, e.y_mul_lo = _p(y_lo), _p(ex.mul), _p(ex.mul_lo), _p(prod), _p(prod_lo)
    outs = None
    if ex.res is not None:
        xs = [x.detach().contiguous() for x in ex.res]
        if any(x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != d.Cout_w or tuple(x.shape[2:]) != (d.Ho, d.Wo) for x in xs) or \
                sum(x.shape[0] for x in xs) != d.B or len(xs) > 2:
            raise RuntimeError("conv residual epilogue: one or two float32 NCHW image sets covering the batch expected")
        outs = [torch.empty_like(x) for x in xs]
        e.res_x, e.res_out, e.res_split = _p(xs[0]), _p(outs[0]), xs[0].shape[0]
        if len(xs) == 2:
            e.res_x2, e.res_out2 = _p(xs[1]), _p(outs[1])
        _chk(*xs)
    _chk(ex.x1_lo, ex.x2_lo, ex.mul, ex.mul_lo)
    st = ws = None
    if stats is not None:
        wsb = lib().uegan_conv2d_fwd_ex_workspace_bytes(C.byref(d), C.byref(e))
        if wsb:
            st = torch.empty((2, d.B, d.Cout), dtype=torch.float32, device=dev)
            ws = torch.empty(((wsb + 3) // 4,), dtype=torch.float32, device=dev)
            e.mean, e.rstd, e.stats_workspace, e.stats_workspace_bytes, e.eps = _p(st[0]), _p(st[1]), _p(ws), wsb, IN_EPS
    taken = C.c_int(0)
    L.check(lib().uegan_conv2d_fwd_ex(C.byref(d), C.byref(e), _p(x1), _p(x2), _p(ohwi), _p(biasc), _p(scale), _p(y), C.byref(taken), _stream()))
    if not taken.value:
        return False
    ex.taken, ex.y_lo, ex.prod, ex.prod_lo, ex.res_out = True, y_lo, prod, prod_lo, (tuple(outs) if outs is not None else None)
    if stats is not None:
        stats.value = st if (taken.value & 2) else None
    return True


split_k = [True]      # small-grid forwards (single-image inference) may split their K loop over several workgroups per tile (uegan_conv2d_fwd_splitk)


def _fwd_plain(d, x1, x2, ohwi, biasc, scale, y):
    """uegan_conv2d_fwd -- with a workspace for a split-K launch where the library would use one (the deep layers of a single-image forward)"""
    if split_k[0] and x1.dtype != torch.float32 and d.B * d.Ho * d.Wo <= 65536 and d.Cout >= 64:
        wsb = lib().uegan_conv2d_fwd_splitk_workspace_bytes(C.byref(d))
        if wsb:
            ws = torch.empty(((wsb + 3) // 4,), dtype=torch.float32, device=x1.device)
            L.check(lib().uegan_conv2d_fwd_splitk(C.byref(d), _p(x1), _p(x2), _p(ohwi), _p(biasc), _p(scale), _p(y), _p(ws), wsb, _stream()))
            return
    L.check(lib().uegan_conv2d_fwd(C.byref(d), _p(x1), _p(x2), _p(ohwi), _p(biasc), _p(scale), _p(y), _stream()))


class _ConvFn(torch.autograd.Function):
    """y = act(scale * conv(pad(cat[x1,x2]), W) + b)  -- uegan_conv2d_fwd / dgrad / wgrad."""

    @staticmethod
    def forward(ctx, x1, x2, weight, bias, cfg, sn, wkey, n_out=1, stats=None, ex=None):
        x1 = x1.contiguous()
        x2 = None if x2 is None else x2.contiguous()
        d = _desc(x1, x2, weight, cfg)
        pair_w = ex is not None and ex.pair_w
        ohwi, ihwo = cfg.packed.get(weight, x1.dtype, d.C1 + d.C2, d.Cout, wkey, cfg.cin_used, pair=pair_w and cfg.stride != 2,
                                    dup=1 if (ex is not None and ex.dup_cin) else 0)
        y = torch.empty((d.B, d.Ho, d.Wo, d.Cout), dtype=x1.dtype, device=x1.device)
        biasc = None if bias is None else bias.detach().contiguous()
        scale = None if sn is None else sn.sigma[1:]
        _chk(x1, x2, y, biasc)
        done = False
        if ex is not None and pair_w and cfg.stride == 2 and x1.dtype != torch.float32:
            # a stride-2 forward on a weight pair: the pair lives in ONE matrix, per tap the hi part of the input channels' weights and then their lo part
            # (the kernel reads the source's channels twice); the plain packs above still serve the backward
            w_il, _ = cfg.packed_il.get(weight, x1.dtype, 2 * (d.C1 + d.C2), d.Cout, wkey, cfg.cin_used, dup=2)
            done = _conv_fwd_ex(d, x1, x2, w_il, None, biasc, scale, y, ex, stats, interleaved=True)
        elif ex is not None:
            done = x1.dtype != torch.float32 and _conv_fwd_ex(d, x1, x2, ohwi, cfg.packed.ohwi_lo if pair_w else None, biasc, scale, y, ex, stats)
            if not done and ex.needs_pairs():
                raise RuntimeError("conv: no kernel takes this layer (%dx%d, %d + %d -> %d channels on %dx%d) with hi + lo pairs (the precise mode covers "
                                   "the conv_dim = 32 generator's full-resolution layers in a 16-bit storage dtype)"
                                   % (d.KH, d.KW, d.C1, d.C2, d.Cout, d.H, d.W))
        wsb = lib().uegan_conv2d_fwd_stats_workspace_bytes(C.byref(d)) if (stats is not None and not done) else 0
        if done:
            pass
        elif wsb:
            # the InstanceNorm that follows wants the per-(image, channel) moments of y: the streaming kernel's epilogue emits them (StatsHolder)
            st = torch.empty((2, d.B, d.Cout), dtype=torch.float32, device=x1.device)
            ws = torch.empty(((wsb + 3) // 4,), dtype=torch.float32, device=x1.device)
            produced = C.c_int(0)
            L.check(lib().uegan_conv2d_fwd_stats(C.byref(d), _p(x1), _p(x2), _p(ohwi), _p(biasc), _p(scale), _p(y), _p(st[0]), _p(st[1]), IN_EPS, _p(ws), wsb,
                                                 C.byref(produced), _stream()))
            stats.value = st if produced.value else None
        else:
            if stats is not None:
                stats.value = None
            _fwd_plain(d, x1, x2, ohwi, biasc, scale, y)
        ctx.cfg, ctx.sn, ctx.d, ctx.ihwo = cfg, sn, d, ihwo
        ctx.pack_version = cfg.packed.version
        ctx.has_x2, ctx.has_bias = x2 is not None, bias is not None
        ctx.wsink, ctx.bsink = _sink_of(weight), _sink_of(bias)
        ctx.save_for_backward(x1, x2, y, weight)
        if n_out == 1:
            return y
        # the activation has n_out consumers: hand each its own alias, so that backward() receives their gradients SEPARATELY and
        # sums them inside the activation-backward kernel (autograd would otherwise add them with n_out - 1 elementwise kernels)
        return tuple(y.view_as(y) for _ in range(n_out))

    @staticmethod
    def backward(ctx, *gs):
        x1, x2, y, weight = ctx.saved_tensors
        cfg, sn, d = ctx.cfg, ctx.sn, ctx.d
        gs = [g.contiguous() for g in gs if g is not None]
        if not gs:
            raise RuntimeError("conv backward without any output gradient")
        st = _stream()
        if len(gs) > 3:
            raise RuntimeError("conv: at most 3 consumers per activation")
        g = gs[0]
        if len(gs) > 1 or (cfg.act != ACT_NONE and not cfg.premasked):
            dz = torch.empty_like(g)
            act = ACT_NONE if cfg.premasked else cfg.act
            L.check(lib().uegan_act_bwd3(_dt(g), act, _p(g), _p(gs[1]) if len(gs) > 1 else None, _p(gs[2]) if len(gs) > 2 else None, _p(y),
                                         _p(dz), g.numel(), st))
        else:
            dz = g
        scale = None if sn is None else sn.sigma[1:]
        dx1 = dx2 = dw = db = None
        if ctx.needs_input_grad[0] or (ctx.has_x2 and ctx.needs_input_grad[1]):
            if cfg.packed.ihwo is ctx.ihwo and cfg.packed.version != ctx.pack_version:
                # torch raises its "modified by an inplace operation" version error for `out = net(x); optimizer.step(); out.backward()`;
                # the optimizer's one-launch re-pack rewrites the packed copy this graph saved, so the same pattern is refused here
                raise RuntimeError("conv backward: the layer's weight was updated by an optimizer step after this forward (the packed "
                                   "copy saved for the data gradient has been rewritten in place); run backward() before step()")
            dx1 = torch.empty_like(x1)
            dx2 = torch.empty_like(x2) if ctx.has_x2 else None
            dwsb = lib().uegan_conv2d_dgrad_workspace_bytes(C.byref(d))     # > 0: small reflect-padded map, pad-grid dgrad + fold
            dws = torch.empty((dwsb + 3) // 4, dtype=torch.float32, device=g.device) if dwsb else None
            if cfg.in_act != ACT_NONE and not ctx.has_x2:
                L.check(lib().uegan_conv2d_dgrad_act(C.byref(d), _p(dz), _p(ctx.ihwo), _p(scale), _p(dx1), _p(dws), dwsb, cfg.in_act, _p(x1), st))
            else:
                if cfg.in_act != ACT_NONE:
                    raise RuntimeError("conv: in_act (deferred activation gradient) needs a single-input conv")
                L.check(lib().uegan_conv2d_dgrad_ws(C.byref(d), _p(dz), _p(ctx.ihwo), _p(scale), _p(dx1), _p(dx2), _p(dws), dwsb, st))
        if ctx.needs_input_grad[2] or (ctx.has_bias and ctx.needs_input_grad[3]):
            wsb = lib().uegan_conv2d_wgrad_workspace_bytes(C.byref(d))
            ws = torch.empty((max(wsb, 4) + 3) // 4, dtype=torch.float32, device=g.device)
            wsink, bsink = ctx.wsink, ctx.bsink
            # straight into the optimizer's flat bucket when both gradients live there (beta = 1 after the first touch); the
            # spectral-norm correction below works in place on THIS call's gradient, so it needs the first touch
            sink = (wsink is not None and ctx.needs_input_grad[2] and (not ctx.has_bias or (bsink is not None and bsink.dirty == wsink.dirty))
                    and (sn is None or not wsink.dirty))
            if sink:
                dw, db, acc = wsink.view, (bsink.view if ctx.has_bias else None), (3 if wsink.dirty else 0)
            else:
                # (a column-slice conv writes only its own columns: the rest of the gradient is zero)
                dw = (torch.zeros if cfg.cin_used is not None else torch.empty)(weight.shape, dtype=torch.float32, device=g.device)
                db = torch.empty((d.Cout_w,), dtype=torch.float32, device=g.device) if ctx.has_bias else None
                acc = 0
            L.check(lib().uegan_conv2d_wgrad_acc(C.byref(d), _p(x1), _p(x2), _p(dz), _p(scale), _p(dw), _p(db), _p(ws), wsb, acc, st))
            if sn is not None:
                wd = weight.detach()
                rows, cols = wd.shape[0], wd[0].numel()
                tmp = torch.empty((lib().uegan_specnorm_grad_workspace_floats(),), dtype=torch.float32, device=g.device)
                L.check(lib().uegan_specnorm_grad(_p(dw), _p(wd), _p(sn.u), _p(sn.v), _p(sn.sigma), _p(dw), rows, cols, _p(tmp), st))
            if sink:
                wsink.mark()
                if ctx.has_bias:
                    bsink.mark()
                dw = db = None
        return dx1, dx2, dw, db, None, None, None, None, None, None


class StatsHolder:
    """conv2d(..., stats=holder): after the call holder.value is a [2, B, C] fp32 tensor (mean, rstd of InstanceNorm: biased variance, eps 1e-5)
    of the conv's result when the kernel that took the layer emitted the moments on its way out (uegan_conv2d_fwd_stats), else None"""
    __slots__ = ("value",)

    def __init__(self):
        self.value = None


def conv2d(x1, x2, weight, bias, cfg, sn=None, wkey=None, n_out=1, stats=None, ex=None):
    """n_out > 1: returns n_out aliases of the output, one per consumer (their gradients are summed inside the activation backward);
    ex: a ConvExtras (hi + lo pairs, product / residual epilogue: uegan_conv2d_fwd_ex) -- its outputs are plain tensors outside the graph"""
    return _ConvFn.apply(x1, x2, weight, bias, cfg, sn, wkey, n_out, stats, ex)


def specnorm_sigma(weight_orig, u, v, do_iter):
    """One power iteration (in place on u, v when do_iter) and sigma; returns SNCall (torch spectral_norm)."""
    wd = weight_orig.detach()
    rows, cols = wd.shape[0], wd[0].numel()
    sigma = torch.empty((2,), dtype=torch.float32, device=wd.device)
    tmp = torch.empty((lib().uegan_specnorm_multi_workspace_floats(rows, cols),), dtype=torch.float32, device=wd.device)
    _chk(wd, u, v)
    L.check(lib().uegan_specnorm_sigma(_p(wd), _p(u), _p(v), rows, cols, 1 if do_iter else 0, SN_EPS, _p(sigma), _p(tmp), _stream()))
    need_grad = torch.is_grad_enabled() and weight_orig.requires_grad
    return SNCall(sigma, u.clone() if need_grad else u, v.clone() if need_grad else v)


# --------------------------------------------------------------------------------------------------------------------
# resampling / pooling / norm / elementwise
# --------------------------------------------------------------------------------------------------------------------
class _Upsample2x(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        B, H, W, Cc = x.shape
        y = torch.empty((B, 2 * H, 2 * W, Cc), dtype=x.dtype, device=x.device)
        _chk(x)
        L.check(lib().uegan_upsample2x_fwd(_dt(x), _p(x), _p(y), B, H, W, Cc, _stream()))
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        B, H2, W2, Cc = g.shape
        gx = torch.empty((B, H2 // 2, W2 // 2, Cc), dtype=g.dtype, device=g.device)
        L.check(lib().uegan_upsample2x_bwd(_dt(g), _p(g), _p(gx), B, H2 // 2, W2 // 2, Cc, _stream()))
        return gx


class _MaxPool2x2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, in_act):
        ctx.in_act = in_act
        x = x.contiguous()
        B, H, W, Cc = x.shape
        y = torch.empty((B, H // 2, W // 2, Cc), dtype=x.dtype, device=x.device)
        _chk(x)
        L.check(lib().uegan_maxpool2x2_fwd(_dt(x), _p(x), _p(y), B, H, W, Cc, _stream()))
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = g.contiguous()
        B, H, W, Cc = x.shape
        gx = torch.empty_like(x)
        L.check(lib().uegan_maxpool2x2_bwd_act(_dt(x), ctx.in_act, _p(x), _p(g), _p(gx), B, H, W, Cc, _stream()))
        return gx, None


class _InstNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pre=None, x_lo=None, lo_out=None):
        """pre: [2, B, C] (mean, rstd) of x already known (StatsHolder.value): only the normalising pass runs.  x_lo (with pre): x is a hi + lo pair,
        and so is the result -- its lo plane goes into the one-element list lo_out"""
        x = x.contiguous()
        B, H, W, Cc = x.shape
        y = torch.empty_like(x)
        _chk(x, x_lo)
        if x_lo is not None:
            if pre is None:
                raise RuntimeError("instnorm of a hi + lo pair needs the moments from the producing convolution")
            stats = pre
            ylo = torch.empty_like(x)
            L.check(lib().uegan_instnorm_apply_pair(_dt(x), _p(x), _p(x_lo), _p(y), _p(ylo), _p(stats[0]), _p(stats[1]), B, H * W, Cc, _stream()))
            lo_out.append(ylo)
        elif pre is not None:
            stats = pre
            L.check(lib().uegan_instnorm_apply(_dt(x), _p(x), _p(y), _p(stats[0]), _p(stats[1]), B, H * W, Cc, _stream()))
        else:
            stats = torch.empty((2, B, Cc), dtype=torch.float32, device=x.device)
            tmp = torch.empty((lib().uegan_reduce_workspace_floats(B, H * W, Cc),), dtype=torch.float32, device=x.device)
            L.check(lib().uegan_instnorm_fwd(_dt(x), _p(x), _p(y), _p(stats[0]), _p(stats[1]), _p(tmp), B, H * W, Cc, IN_EPS, _stream()))
        ctx.save_for_backward(y, stats)
        return y

    @staticmethod
    def backward(ctx, g):
        y, stats = ctx.saved_tensors
        g = g.contiguous()
        B, H, W, Cc = y.shape
        dx = torch.empty_like(y)
        tmp = torch.empty((lib().uegan_reduce_workspace_floats(B, H * W, Cc),), dtype=torch.float32, device=y.device)
        L.check(lib().uegan_instnorm_bwd(_dt(y), _p(g), _p(y), _p(stats[1]), _p(dx), _p(tmp), B, H * W, Cc, _stream()))
        return dx, None, None, None


class _Mul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, act_a, act_b, given=None):
        """given: the product, already formed by the epilogue of the convolution that produced `a` (ConvExtras.prod): no kernel runs here"""
        ctx.acts = (act_a, act_b)
        a, b = a.contiguous(), b.contiguous()
        _chk(a, b)
        if given is not None:
            y = given.view_as(given)
        else:
            y = torch.empty_like(a)
            L.check(lib().uegan_mul_fwd(_dt(a), _p(a), _p(b), _p(y), a.numel(), _stream()))
        ctx.save_for_backward(a, b)
        return y

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        da, db = torch.empty_like(a), torch.empty_like(b)
        L.check(lib().uegan_mul_bwd_act(_dt(a), ctx.acts[0], ctx.acts[1], _p(g), _p(a), _p(b), _p(da), _p(db), a.numel(), _stream()))
        return da, db, None, None, None


class _ResidualClamp(torch.autograd.Function):
    """out(NCHW fp32) = clamp(res(NHWC) + x(NCHW fp32), -1, 1)   models.py:72"""

    @staticmethod
    def forward(ctx, res, x, res_act=ACT_NONE, given=None):
        """given: (out,) already written by the epilogue of the convolution that produced res (ConvExtras.res_out): no kernel runs here"""
        ctx.res_act = res_act
        res, x = res.contiguous(), x.contiguous()
        B, H, W, Cp = res.shape
        Cc = x.shape[1]
        _chk(res, x)
        if given is not None:
            out = given[0].view_as(given[0])
        else:
            out = torch.empty((B, Cc, H, W), dtype=torch.float32, device=res.device)
            L.check(lib().uegan_residual_clamp_fwd(_dt(res), _p(res), _p(x), _p(out), B, Cc, Cp, H, W, _stream()))
        ctx.save_for_backward(res, x)
        return out

    @staticmethod
    def backward(ctx, g):
        res, x = ctx.saved_tensors
        g = g.contiguous()
        B, H, W, Cp = res.shape
        Cc = x.shape[1]
        dres = torch.empty_like(res)
        dx = torch.empty_like(x) if ctx.needs_input_grad[1] else None
        L.check(lib().uegan_residual_clamp_bwd_act(_dt(res), ctx.res_act, _p(g), _p(res), _p(x), _p(dres), _p(dx), B, Cc, Cp, H, W, _stream()))
        return dres, dx, None, None


class _ResidualClampPair(torch.autograd.Function):
    """models.py:72 for a generator pass over two image sets at once: res [Ba + Bb, H, W, Cp] -> clamp(res[:Ba] + xa), clamp(res[Ba:] + xb)
    as two separate NCHW fp32 tensors (each is consumed by its own losses)"""

    @staticmethod
    def forward(ctx, res, xa, xb, res_act=ACT_NONE, given=None):
        """given: (oa, ob) already written by dec5.1's epilogue (ConvExtras.res_out)"""
        ctx.res_act = res_act
        res, xa, xb = res.contiguous(), xa.contiguous(), xb.contiguous()
        Bt, H, W, Cp = res.shape
        Ba, Cc = xa.shape[0], xa.shape[1]
        _chk(res, xa, xb)
        if given is not None:
            oa, ob = given[0].view_as(given[0]), given[1].view_as(given[1])
        else:
            oa = torch.empty_like(xa)
            ob = torch.empty_like(xb)
            st = _stream()
            L.check(lib().uegan_residual_clamp_fwd(_dt(res), _p(res), _p(xa), _p(oa), Ba, Cc, Cp, H, W, st))
            L.check(lib().uegan_residual_clamp_fwd(_dt(res), _p(res[Ba:]), _p(xb), _p(ob), Bt - Ba, Cc, Cp, H, W, st))
        ctx.save_for_backward(res, xa, xb)
        return oa, ob

    @staticmethod
    def backward(ctx, ga, gb):
        res, xa, xb = ctx.saved_tensors
        Bt, H, W, Cp = res.shape
        Ba, Cc = xa.shape[0], xa.shape[1]
        dres = torch.empty_like(res)
        st = _stream()
        dxa = dxb = None
        for g, x, r, dr, nb, idx in ((ga, xa, res, dres, Ba, 1), (gb, xb, res[Ba:], dres[Ba:], Bt - Ba, 2)):
            if g is None:
                dr.zero_()             # this output was not used by any loss
                continue
            g = g.contiguous()
            dx = torch.empty_like(x) if ctx.needs_input_grad[idx] else None
            L.check(lib().uegan_residual_clamp_bwd_act(_dt(res), ctx.res_act, _p(g), _p(r), _p(x), _p(dr), _p(dx), nb, Cc, Cp, H, W, st))
            if idx == 1:
                dxa = dx
            else:
                dxb = dx
        return dres, dxa, dxb, None, None


def residual_clamp_pair(res, xa, xb, res_act=ACT_NONE, given=None):
    return _ResidualClampPair.apply(res, xa, xb, res_act, given)


def upsample2x(x):
    return _Upsample2x.apply(x)


def maxpool2x2(x, in_act=ACT_NONE):
    """in_act: the activation whose (deferred) gradient the pool's backward applies for x's producer (see ConvCfg)."""
    return _MaxPool2x2.apply(x, in_act)


def instnorm(x, pre=None):
    return _InstNorm.apply(x, pre)


def instnorm_pair(x, x_lo, pre):
    """InstanceNorm of the pair (x, x_lo) with its moments `pre` -> (y, y_lo); y_lo is a plain tensor outside the graph (the backward reads y)"""
    lo = []
    y = _InstNorm.apply(x, pre, x_lo, lo)
    return y, lo[0]


def mul(a, b, act_a=ACT_NONE, act_b=ACT_NONE, given=None):
    """act_a / act_b: the activation whose (deferred) gradient the backward applies for a's / b's producer (see ConvCfg); given: see _Mul"""
    return _Mul.apply(a, b, act_a, act_b, given)


def residual_clamp(res, x, res_act=ACT_NONE, given=None):
    """res_act: the activation that produced res, its gradient deferred to this op's backward (see ConvCfg); given: see _ResidualClamp"""
    return _ResidualClamp.apply(res, x, res_act, given)


# --------------------------------------------------------------------------------------------------------------------
# raw launchers (no autograd): building blocks of the whole-network functions in uegan_amd/fused.py, which sequence the
# kernels of a fixed network explicitly instead of recording one autograd node per layer
# --------------------------------------------------------------------------------------------------------------------
def _sub_desc(d, nb):
    """the same convolution on the first nb images of the batch"""
    if nb is None or nb == d.B:
        return d
    d2 = L.ConvDesc.from_buffer_copy(d)
    d2.B = nb
    return d2


def raw_conv_fwd(x1, x2, weight, bias, cfg, scale=None, wkey=None, pool=False, n_full=None, n_idx=0, stats=None):
    """y = act(scale * conv(pad(cat[x1, x2]), W) + b) -> (y, desc, w_ihwo); pool=True: -> (y, desc, w_ihwo, maxpool2x2(y)) with the
    pooled tensor written by the convolution's epilogue where the kernel can (uegan_conv2d_fwd_pool).  n_full (with pool): only the first
    n_full images need y itself -- y[n_full:] is UNDEFINED afterwards (uegan_conv2d_fwd_pool_part), the pooled tensor is complete.
    n_idx > 0 (with pool): -> (y, desc, w_ihwo, pooled, idx) with idx (uint8, pooled shape) the window position of each maximum for the first
    n_idx images (uegan_conv2d_fwd_pool_idx): raw_maxpool_bwd_idx routes the gradient with it instead of with y"""
    d = _desc(x1, x2, weight, cfg)
    ohwi, ihwo = cfg.packed.get(weight, x1.dtype, d.C1 + d.C2, d.Cout, wkey, cfg.cin_used)
    y = torch.empty((d.B, d.Ho, d.Wo, d.Cout), dtype=x1.dtype, device=x1.device)
    biasc = None if bias is None else bias.detach()
    _chk(x1, x2, y, biasc)
    if pool:
        yp = torch.empty((d.B, d.Ho // 2, d.Wo // 2, d.Cout), dtype=x1.dtype, device=x1.device)
        nf = d.B if n_full is None else int(n_full)
        if n_idx:
            idx = torch.empty((d.B, d.Ho // 2, d.Wo // 2, d.Cout), dtype=torch.uint8, device=x1.device)
            L.check(lib().uegan_conv2d_fwd_pool_idx(C.byref(d), _p(x1), _p(x2), _p(ohwi), _p(biasc), _p(scale), _p(y), _p(yp), _p(idx), nf, int(n_idx), _stream()))
            return y, d, ihwo, yp, idx
        L.check(lib().uegan_conv2d_fwd_pool_part(C.byref(d), _p(x1), _p(x2), _p(ohwi), _p(biasc), _p(scale), _p(y), _p(yp), nf, _stream()))
        return y, d, ihwo, yp
    wsb = lib().uegan_conv2d_fwd_stats_workspace_bytes(C.byref(d)) if stats is not None else 0
    if wsb:       # (stats: a StatsHolder -- the InstanceNorm moments of y from the kernel's epilogue where it has them, see conv2d)
        st = torch.empty((2, d.B, d.Cout), dtype=torch.float32, device=x1.device)
        ws = torch.empty(((wsb + 3) // 4,), dtype=torch.float32, device=x1.device)
        produced = C.c_int(0)
        L.check(lib().uegan_conv2d_fwd_stats(C.byref(d), _p(x1), _p(x2), _p(ohwi), _p(biasc), _p(scale), _p(y), _p(st[0]), _p(st[1]), IN_EPS, _p(ws), wsb,
                                             C.byref(produced), _stream()))
        stats.value = st if produced.value else None
        return y, d, ihwo
    if stats is not None:
        stats.value = None
    _fwd_plain(d, x1, x2, ohwi, biasc, scale, y)
    return y, d, ihwo


def raw_conv_dgrad(d, dz, ihwo, scale=None, in_act=ACT_NONE, x_act=None, nb=None, two=False):
    """data gradient on the first `nb` images (default: all): dx1 (, dx2) = dgrad(dz) [* act'(x_act)]"""
    d = _sub_desc(d, nb)
    dev = dz.device
    dx1 = torch.empty((d.B, d.H, d.W, d.C1), dtype=dz.dtype, device=dev)
    dx2 = torch.empty((d.B, d.H, d.W, d.C2), dtype=dz.dtype, device=dev) if two else None
    dwsb = lib().uegan_conv2d_dgrad_workspace_bytes(C.byref(d))
    dws = torch.empty((dwsb + 3) // 4, dtype=torch.float32, device=dev) if dwsb else None
    st = _stream()
    if in_act != ACT_NONE:
        if two:
            raise RuntimeError("conv: in_act (deferred activation gradient) needs a single-input conv")
        L.check(lib().uegan_conv2d_dgrad_act(C.byref(d), _p(dz), _p(ihwo), _p(scale), _p(dx1), _p(dws), dwsb, in_act, _p(x_act), st))
    else:
        L.check(lib().uegan_conv2d_dgrad_ws(C.byref(d), _p(dz), _p(ihwo), _p(scale), _p(dx1), _p(dx2), _p(dws), dwsb, st))
    return dx1, dx2


def raw_conv_wgrad(d, x1, x2, dz, weight, bias, scale=None, nb=None):
    """weight (and bias) gradient over the first `nb` images, written into the parameters' gradient sinks when they have one
    (beta = 1 after the first touch) -- returns (dw, db) tensors for autograd, None where the gradient went into a sink."""
    d = _sub_desc(d, nb)
    dev = dz.device
    st = _stream()
    wsb = lib().uegan_conv2d_wgrad_workspace_bytes(C.byref(d))
    ws = torch.empty((max(wsb, 4) + 3) // 4, dtype=torch.float32, device=dev)
    wsink, bsink = _sink_of(weight), _sink_of(bias)
    has_bias = bias is not None
    sink = wsink is not None and (not has_bias or (bsink is not None and bsink.dirty == wsink.dirty))
    if sink:
        dw, db, acc = wsink.view, (bsink.view if has_bias else None), (3 if wsink.dirty else 0)
    else:
        dw = torch.empty(weight.shape, dtype=torch.float32, device=dev)
        db = torch.empty((d.Cout_w,), dtype=torch.float32, device=dev) if has_bias else None
        acc = 0
    L.check(lib().uegan_conv2d_wgrad_acc(C.byref(d), _p(x1), _p(x2), _p(dz), _p(scale), _p(dw), _p(db), _p(ws), wsb, acc, st))
    if sink:
        wsink.mark()
        if has_bias:
            bsink.mark()
        return None, None
    return dw, db


def raw_act_bwd(g, y, act):
    dz = torch.empty_like(g)
    L.check(lib().uegan_act_bwd(_dt(g), act, _p(g), _p(y), _p(dz), g.numel(), _stream()))
    return dz


def raw_to_nhwc(xs, dtype, a=None, b=None, pair=False):
    """several NCHW fp32 image batches of one shape -> ONE NHWC tensor [sum(B_i), H, W, Cp] (batch-concatenated)"""
    xs = [x.detach().contiguous() for x in xs]
    B0, Cc, H, W = xs[0].shape
    for x in xs:
        if x.dtype != torch.float32 or tuple(x.shape[1:]) != (Cc, H, W):
            raise TypeError("module inputs must be float32 NCHW batches of one image shape (data_loader.py:79-81)")
    Cp = cpad(Cc, dtype)
    Bt = sum(x.shape[0] for x in xs)
    y = torch.empty((Bt, H, W, Cp), dtype=dtype, device=xs[0].device)
    _chk(y, *xs)
    off = 0
    fn = lib().uegan_nchw_to_nhwc_pair if pair else lib().uegan_nchw_to_nhwc
    for x in xs:
        L.check(fn(_dt(y), _p(x), _p(y[off:]), x.shape[0], Cc, Cp, H, W, _farr(a), _farr(b), _stream()))
        off += x.shape[0]
    return y


def raw_to_nchw_grad(g_nhwc, channels, a=None):
    """gradient of raw_to_nhwc's conversion for one source: NHWC -> NCHW fp32, times the forward's per-channel scale a"""
    B, H, W, Cp = g_nhwc.shape
    gx = torch.empty((B, channels, H, W), dtype=torch.float32, device=g_nhwc.device)
    L.check(lib().uegan_nhwc_to_nchw(_dt(g_nhwc), _p(g_nhwc), _p(gx), B, channels, Cp, H, W, _farr(a), _stream()))
    return gx


def copy_images(dst, src_a, src_b, dst_idx, src_idx):
    """dst[dst_idx[i]] = src_a[src_idx[i]] if src_idx[i] >= 0 else src_b[~src_idx[i]] (whole images; stacks of contiguous fp32
    images of one shape).  utils.py:41-46 as one gather / one scatter (include/uegan_hip.h uegan_copy_images)."""
    for t in (dst, src_a, src_b):
        if t.dtype != torch.float32 or tuple(t.shape[1:]) != tuple(dst.shape[1:]):
            raise RuntimeError("copy_images: float32 image stacks of one image shape expected")
    _chk(dst, src_a, src_b)
    elems = dst[0].numel()
    n = len(dst_idx)
    for s0 in range(0, n, 64):
        di, si = dst_idx[s0:s0 + 64], src_idx[s0:s0 + 64]
        for d, s in zip(di, si):
            if not (0 <= d < dst.shape[0]) or not (0 <= (s if s >= 0 else ~s) < (src_a if s >= 0 else src_b).shape[0]):
                raise IndexError("copy_images: index out of range")
        L.check(lib().uegan_copy_images(_p(dst), _p(src_a), _p(src_b), (C.c_int32 * len(di))(*di), (C.c_int32 * len(si))(*si), len(di), elems,
                                        _stream()))


# --------------------------------------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------------------------------------
def _ptr_table(ts):
    return (C.c_void_p * len(ts))(*[_p(t) for t in ts])


class _RaHinge(torch.autograd.Function):
    """GANLoss('rahinge') over lists of prediction maps (losses.py:348-362, 393-409); returns shape [1]."""

    @staticmethod
    def forward(ctx, for_discriminator, nscales, *maps):
        reals = [m.contiguous() for m in maps[:nscales]]
        fakes = [m.contiguous() for m in maps[nscales:]]
        for r, f in zip(reals, fakes):
            if r.dtype != torch.float32 or f.dtype != torch.float32 or r.numel() != f.numel():
                raise RuntimeError("rahinge: maps must be float32 with matching sizes")
        dev = reals[0].device
        loss = torch.empty((1,), dtype=torch.float32, device=dev)
        tmp = torch.empty((lib().uegan_rahinge_workspace_floats(nscales),), dtype=torch.float32, device=dev)
        n = (C.c_int64 * nscales)(*[r.numel() for r in reals])
        _chk(*reals, *fakes)
        L.check(lib().uegan_rahinge_fwd(nscales, _ptr_table(reals), _ptr_table(fakes), n, 1 if for_discriminator else 0, _p(loss), _p(tmp),
                                        _stream()))
        ctx.for_d, ctx.nscales = for_discriminator, nscales
        ctx.save_for_backward(tmp, *reals, *fakes)
        return loss

    @staticmethod
    def backward(ctx, g):
        tmp = ctx.saved_tensors[0]
        ns = ctx.nscales
        reals, fakes = ctx.saved_tensors[1:1 + ns], ctx.saved_tensors[1 + ns:]
        g = g.contiguous().float()
        greal = [torch.empty_like(r) if ctx.needs_input_grad[2 + i] else None for i, r in enumerate(reals)]
        gfake = [torch.empty_like(f) if ctx.needs_input_grad[2 + ns + i] else None for i, f in enumerate(fakes)]
        n = (C.c_int64 * ns)(*[r.numel() for r in reals])
        L.check(lib().uegan_rahinge_bwd(ns, _ptr_table(reals), _ptr_table(fakes), n, 1 if ctx.for_d else 0, _p(tmp), _p(g), _ptr_table(greal),
                                        _ptr_table(gfake), _stream()))
        return (None, None) + tuple(greal) + tuple(gfake)


def rahinge(real_preds, fake_preds, for_discriminator):
    return _RaHinge.apply(bool(for_discriminator), len(real_preds), *real_preds, *fake_preds)


REC_KINDS = {"l1": 0, "smoothl1": 1, "l2": 2}


class _MsRec(torch.autograd.Function):
    """MultiscaleRecLoss.forward (losses.py:219-231): criterion `kind` at `nscales` scales, AvgPool2d(2,2) between, weights 2^-i."""

    @staticmethod
    def forward(ctx, pred, gt, kind, nscales):
        pred, gt = pred.contiguous(), gt.contiguous()
        if pred.dtype != torch.float32 or gt.dtype != torch.float32 or pred.shape != gt.shape:
            raise RuntimeError("multiscale reconstruction loss: float32 NCHW tensors of equal shape expected")
        B, Cc, H, W = pred.shape
        loss = torch.empty((1,), dtype=torch.float32, device=pred.device)
        scratch = torch.empty((lib().uegan_msrec_scratch_floats(),), dtype=torch.float32, device=pred.device)
        _chk(pred, gt)
        L.check(lib().uegan_msrec_fwd(_p(pred), _p(gt), _p(loss), _p(scratch), B, Cc, H, W, kind, nscales, _stream()))
        ctx.save_for_backward(pred, gt)
        ctx.kind, ctx.nscales = kind, nscales
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        pred, gt = ctx.saved_tensors
        g = g.contiguous().float().reshape(1)
        B, Cc, H, W = pred.shape
        gp = torch.empty_like(pred)
        L.check(lib().uegan_msrec_bwd(_p(pred), _p(gt), _p(g), _p(gp), B, Cc, H, W, ctx.kind, ctx.nscales, _stream()))
        return gp, None, None, None


def multiscale_rec(pred, gt, kind="l1", nscales=3):
    return _MsRec.apply(pred, gt, REC_KINDS[kind], int(nscales))


def multiscale_l1(pred, gt):
    return multiscale_rec(pred, gt, "l1", 3)


class _Percep(torch.autograd.Function):
    """sum_t w_t * MSE(IN(x_t), IN(y_t)) over VGG taps (losses.py:30-34). Gradient flows to the x taps only."""

    @staticmethod
    def forward(ctx, weights_act, ntaps, *taps):
        weights, ctx.in_act = weights_act
        xs = [t.contiguous() for t in taps[:ntaps]]
        ys = [t.contiguous() for t in taps[ntaps:]]
        dev = xs[0].device
        loss = torch.zeros((1,), dtype=torch.float32, device=dev)
        tmps = []
        st = _stream()
        _chk(*xs, *ys)
        for w, x, y in zip(weights, xs, ys):
            B, H, W, Cc = x.shape
            tmp = torch.empty((3 * lib().uegan_reduce_workspace_floats(B, H * W, Cc),), dtype=torch.float32, device=dev)
            L.check(lib().uegan_percep_tap_fwd(_dt(x), _p(x), _p(y), float(w), _p(loss), _p(tmp), B, H * W, Cc, IN_EPS, st))
            tmps.append(tmp)
        ctx.weights, ctx.ntaps = weights, ntaps
        ctx.save_for_backward(*xs, *ys, *tmps)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        nt = ctx.ntaps
        sv = ctx.saved_tensors
        xs, ys, tmps = sv[:nt], sv[nt:2 * nt], sv[2 * nt:]
        g = g.contiguous().float().reshape(1)
        st = _stream()
        grads = []
        for i, (w, x, y, tmp) in enumerate(zip(ctx.weights, xs, ys, tmps)):
            if not ctx.needs_input_grad[2 + i]:
                grads.append(None)
                continue
            B, H, W, Cc = x.shape
            gx = torch.empty_like(x)
            L.check(lib().uegan_percep_tap_bwd_act(_dt(x), ctx.in_act, _p(x), _p(y), float(w), _p(g), _p(gx), _p(tmp), B, H * W, Cc, IN_EPS, st))
            grads.append(gx)
        return (None, None) + tuple(grads) + (None,) * nt


def perceptual_taps_loss(x_taps, y_taps, weights, in_act=ACT_NONE):
    """in_act: the x taps' producers deferred their activation gradient to their consumers (see ConvCfg); applied here."""
    return _Percep.apply((tuple(weights), in_act), len(x_taps), *x_taps, *y_taps)


def gather_scalars(terms):
    """[n] fp32 device vector of n <= 8 device scalars (uegan_gather_scalars): the step's logged losses in ONE buffer, so that reading them
    back is one copy and one host sync instead of five (trainer.py:98-119 calls .item() five times)"""
    ts = [t.detach().contiguous().float().reshape(-1)[:1] for t in terms]
    out = torch.empty((len(ts),), dtype=torch.float32, device=ts[0].device)
    _chk(*ts)
    L.check(lib().uegan_gather_scalars(len(ts), _ptr_table(ts), _p(out), _stream()))
    return out


def zero_(t):
    """t.zero_() as a stream memset (uegan_fill_zero)"""
    _chk(t)
    L.check(lib().uegan_fill_zero(_p(t), t.numel() * t.element_size(), _stream()))
    return t


class _LossSum(torch.autograd.Function):
    """total = sum_i w_i * t_i over device scalars, left to right (trainer.py:104-115); also hands back the scaled terms for logging"""

    @staticmethod
    def forward(ctx, weights, *terms):
        ts = [t.contiguous().float().reshape(1) for t in terms]
        n = len(ts)
        dev = ts[0].device
        total = torch.empty((1,), dtype=torch.float32, device=dev)
        scaled = torch.empty((n,), dtype=torch.float32, device=dev)
        _chk(*ts)
        L.check(lib().uegan_scalar_wsum(n, _ptr_table(ts), (C.c_float * n)(*weights), _p(total), _p(scaled), _stream()))
        ctx.weights, ctx.shapes = weights, [t.shape for t in terms]
        ctx.mark_non_differentiable(scaled)
        return total, scaled

    @staticmethod
    def backward(ctx, g, _gs):
        n = len(ctx.weights)
        g = g.contiguous().float().reshape(1)
        gout = torch.empty((n,), dtype=torch.float32, device=g.device)
        L.check(lib().uegan_scalar_wsum_bwd(n, (C.c_float * n)(*ctx.weights), _p(g), _p(gout), _stream()))
        return (None,) + tuple(gout[i].reshape(s) for i, s in enumerate(ctx.shapes))


def loss_sum(terms, weights):
    """(sum_i w_i * t_i as a [1] tensor, the n scaled terms as a detached [n] tensor) for device-scalar losses"""
    return _LossSum.apply(tuple(float(w) for w in weights), *terms)


# --------------------------------------------------------------------------------------------------------------------
# fused Adam
# --------------------------------------------------------------------------------------------------------------------
class FusedAdamL2:
    """torch.optim.Adam(lr, betas, eps, weight_decay) semantics (trainer.py:337-338) as ONE kernel launch over all
    tensors; gradients are read from a flat fp32 bucket (the RCCL all-reduce buffer) scaled by `grad_scale`.
    `state_dict()` / `load_state_dict()` speak torch.optim.Adam's format (the reference checkpoint's `g_optimizer` /
    `d_optimizer` entries, trainer.py:199-207, 409-410)."""

    def __init__(self, params, lr, betas=(0.5, 0.999), eps=1e-8, weight_decay=1e-4):
        self.params = [p for p in params if p.requires_grad]
        self.before_access = None        # set by the Trainer: applies an update it left pending (data-parallel overlap) before the
                                         # optimizer's state or learning rate is read or changed from outside
        self._in_step = False
        self.lr, self.betas, self.eps, self.weight_decay = lr, tuple(betas), eps, weight_decay
        self.initial_lr = None           # set by trainer.LambdaLR (torch writes `initial_lr` into the param group)
        self.step_count = 0
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.flat_grad = torch.zeros((total,), dtype=torch.float32, device=dev)
        self.m = torch.zeros_like(self.flat_grad)
        self.v = torch.zeros_like(self.flat_grad)
        descs = (L.AdamTensor * len(self.params))()
        off = 0
        self.max_n = 0
        self._offsets = []
        for i, p in enumerate(self.params):
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise TypeError("FusedAdamL2 expects contiguous float32 parameters")
            n = p.numel()
            p.grad = self.flat_grad[off:off + n].view_as(p)       # autograd accumulates in place into the bucket
            p._uegan_sink = GradSink(p.grad)                      # ... and the weight-gradient kernels write it directly
            descs[i].p = p.data_ptr()
            descs[i].g = self.flat_grad.data_ptr() + 4 * off
            descs[i].m = self.m.data_ptr() + 4 * off
            descs[i].v = self.v.data_ptr() + 4 * off
            descs[i].n = n
            self._offsets.append(off)
            off += n
            self.max_n = max(self.max_n, n)
        raw = bytes(descs)
        host = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
        self.desc_dev = host.to(dev)
        self._views = [p.grad for p in self.params]
        self._packs = PackTable()

    def _flush(self):
        if self.before_access is not None and not self._in_step:
            self._in_step = True
            try:
                self.before_access()
            finally:
                self._in_step = False

    @property
    def lr(self):
        return self._lr

    @lr.setter
    def lr(self, value):
        self._flush()                    # a pending update belongs to the OLD learning rate
        self._lr = value

    def zero_grad(self):
        zero_(self.flat_grad)
        for p, v in zip(self.params, self._views):
            p.grad = v
            p._uegan_sink.dirty = False

    def step(self, grad_scale=1.0):
        self.step_count += 1
        for p, v in zip(self.params, self._views):
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                raise RuntimeError("FusedAdamL2: a parameter's .grad no longer aliases the flat bucket")
        L.check(lib().uegan_adam_l2_step(_p(self.desc_dev), len(self.params), self.max_n, self.lr, self.betas[0], self.betas[1], self.eps,
                                         self.weight_decay, grad_scale, self.step_count, _stream()))
        invalidate_weight_caches(self.params)
        self._packs.repack(self.params)          # every packed copy of the updated weights, one launch

    # ---- torch.optim.Adam checkpoint format
    @property
    def param_groups(self):
        g = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False,
             "params": list(range(len(self.params)))}
        if self.initial_lr is not None:
            g["initial_lr"] = self.initial_lr
        return [g]

    def state_dict(self):
        """{'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [...]} exactly as torch.optim.Adam.state_dict():
        parameter index = position in the constructor's iterable, empty `state` before the first step.  `step` is a Python int
        (torch 1.4, the reference's version; current torch converts it on load)."""
        self._flush()
        state = {}
        if self.step_count > 0:
            for i, (p, off) in enumerate(zip(self.params, self._offsets)):
                n = p.numel()
                state[i] = {"step": self.step_count, "exp_avg": self.m[off:off + n].view_as(p).clone(),
                            "exp_avg_sq": self.v[off:off + n].view_as(p).clone()}
        return {"state": state, "param_groups": self.param_groups}

    def load_state_dict(self, sd):
        self._flush()
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.params):
            raise ValueError("loaded state dict has a different number of parameter groups / parameters")
        g = groups[0]
        if g.get("amsgrad", False):
            raise NotImplementedError("amsgrad checkpoints are not supported (the reference never sets it, trainer.py:337-338)")
        self.lr, self.betas, self.eps, self.weight_decay = float(g["lr"]), tuple(float(b) for b in g["betas"]), float(g["eps"]), float(g["weight_decay"])
        if "initial_lr" in g:
            self.initial_lr = float(g["initial_lr"])
        state = sd["state"]
        steps = set()
        self.m.zero_()
        self.v.zero_()
        for key, st in state.items():
            i = int(key)
            p, off = self.params[i], self._offsets[i]
            n = p.numel()
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError("optimizer state %d has shape %s, parameter has %s" % (i, tuple(st["exp_avg"].shape), tuple(p.shape)))
            self.m[off:off + n].view_as(p).copy_(st["exp_avg"])
            self.v[off:off + n].view_as(p).copy_(st["exp_avg_sq"])
            steps.add(int(float(st["step"])))
        if len(state) not in (0, len(self.params)) or len(steps) > 1:
            raise ValueError("FusedAdamL2 keeps ONE step counter: every parameter must carry the same `step` (got %s)" % sorted(steps))
        self.step_count = steps.pop() if steps else 0
