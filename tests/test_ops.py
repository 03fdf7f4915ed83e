"""Kernel-level parity: every C-ABI op (through the autograd wrappers in uegan_amd/ops.py) against plain PyTorch fp32 /
the CPU oracle on identical seeded inputs.  Each test runs on the CPU fiber emulator (`-m "not gpu"`) and on the
MI355X (`-m gpu`).  Tolerances: fp32 path 2e-5 relative-to-max (accumulation order only); bf16 storage 2e-2
against a reference evaluated on bf16-rounded operands."""
import pytest
import torch
import torch.nn.functional as F

from helpers import BACKENDS, bf16_round, half_round, nchw, nhwc, rel, set_tuning, use_backend
from oracle import uegan_oracle as O
from uegan_amd import _lib, ops

F32_TOL = 2e-5
BF16_TOL = 2e-2
F16_TOL = 3e-3           # fp16 storage: 11 significant bits (the fp16-format build of the library, csrc/common.h)


@pytest.fixture(autouse=True)
def _pin_direct_reflect_dgrad(monkeypatch):
    """The kernel-matrix tests in this file pin which kernel a case reaches, including the mirrored-image (MODE 2)
    dgrad kernels; the pad-grid + fold route small reflection-padded maps take by default is switched off here and has
    its own test (test_conv_dgrad_pad_grid_fold).  The model-level tests run the library defaults."""
    set_tuning("FOLD_MAX", 0)


def ref_conv(x, w, b, stride, pad_mode, act):
    p = (w.shape[-1] - 1) // 2
    if pad_mode == ops.PAD_REFLECT and p > 0:
        y = F.conv2d(F.pad(x, (p, p, p, p), mode="reflect"), w, b, stride=stride)
    else:
        y = F.conv2d(x, w, b, stride=stride, padding=p)
    return {0: lambda t: t, 1: lambda t: F.leaky_relu(t, 0.2), 2: F.relu, 3: torch.tanh}[act](y)


# (B, C1, C2, H, W, Cout, k, stride, pad_mode, act)
CONV_CASES = [
    (1, 8, 0, 12, 12, 8, 3, 1, 1, 1),     # reflect 3x3 + LeakyReLU (G ConvBlock)
    (2, 3, 0, 16, 16, 8, 7, 1, 1, 1),     # enc1 / d1-like: 3 input channels (scalar gather path)
    (1, 8, 0, 16, 16, 16, 3, 2, 1, 1),    # stride 2 (G encoder)
    (1, 16, 0, 8, 8, 1, 5, 1, 1, 3),      # D prediction head: Cout = 1, tanh, 5x5
    (1, 8, 0, 10, 10, 8, 3, 1, 0, 2),     # VGG: zero pad + ReLU
    (1, 32, 0, 8, 8, 32, 3, 1, 1, 0),     # K a multiple of the 32-wide K step
    (1, 8, 0, 3, 3, 8, 5, 2, 1, 1),       # 3x3 input with pad 2: every pixel has 3 reflected images (d5 at 96^2)
    (1, 4, 0, 8, 8, 4, 1, 1, 1, 0),       # 1x1
    (1, 8, 8, 12, 12, 8, 3, 1, 1, 1),     # two-source (virtual concat) decoder conv
    (3, 8, 0, 20, 12, 136, 3, 1, 1, 1),   # Cout > 128 (two N tiles, ragged), M not a tile multiple, B = 3
    (1, 8, 0, 9, 7, 8, 7, 2, 1, 1),       # odd sizes, 7x7 stride 2 (D trunk)
    (2, 32, 0, 4, 4, 1, 7, 1, 1, 3),      # 4x4 input with pad 3
    (1, 72, 0, 9, 20, 16, 3, 1, 1, 1),    # several 64-channel chunks per patch x reflected images, 2 tiles wide
    (1, 40, 0, 6, 6, 8, 5, 1, 0, 2),      # 5x5 zero pad
    (1, 40, 32, 17, 17, 24, 3, 1, 1, 1),  # two sources, chunk boundary inside source 2, 3 x 2 tiles
    (2, 136, 0, 6, 34, 8, 7, 1, 1, 0),    # 7x7, 3 chunks, 3 tiles wide
    # channel counts that are multiples of one K step: these take the patch-resident kernel (fwd and dgrad)
    (1, 64, 0, 9, 20, 64, 3, 1, 1, 1),    # reflect, border tiles with mirrored images, 2x2 tiles
    (1, 128, 0, 6, 18, 64, 3, 1, 0, 2),   # zero pad (VGG), two chunks
    (1, 64, 0, 7, 7, 64, 5, 1, 1, 0),     # 5x5
    (1, 64, 64, 10, 10, 64, 3, 1, 1, 1),  # two sources, dgrad writes two destinations in one launch
    (1, 64, 0, 8, 8, 64, 7, 1, 1, 1),     # 7x7
    (2, 64, 0, 3, 3, 128, 5, 1, 1, 1),    # 3x3 input, pad 2: all three images on both axes
    # stride-2 layers whose dgrad runs the patch kernel per parity class (dz channels = Cout = multiple of a K step)
    (1, 8, 0, 16, 16, 64, 3, 2, 1, 1),    # G encoder 3x3 s2
    (1, 8, 0, 17, 13, 64, 7, 2, 1, 1),    # D trunk 7x7 s2, odd sizes (parity classes of different extent)
    (1, 8, 0, 12, 36, 64, 5, 2, 1, 1),    # D trunk 5x5 s2, several tiles wide
    (2, 16, 0, 6, 6, 128, 5, 2, 1, 1),    # small map: every tile has mirrored images, two chunks
    # narrow heads (<= 4 real output channels, stride 1): VALU head kernels for forward, dgrad and wgrad
    (1, 32, 0, 12, 40, 3, 7, 1, 1, 3),    # G last layer shape: 32 -> 3, 7x7, tanh; two tiles wide
    (1, 8, 0, 9, 9, 2, 3, 1, 1, 0),       # 3x3
    (2, 72, 0, 10, 34, 1, 5, 1, 1, 3),    # D head: C -> 1, 5x5, three channel chunks, ragged tiles
    (1, 128, 0, 24, 80, 1, 7, 1, 1, 3),   # D head 128 -> 1, 7x7 on a map with an interior tile: the VALU data-gradient kernel's unrolled path + its border path
    (2, 64, 0, 9, 33, 1, 7, 1, 1, 3),     # ... every pixel of the map within 3 of a border (up to 2 x 2 images), ragged tiles
    # maps >= 16 rows with > 32 output channels: 256-pixel tiles, 8 waves, 3-deep weight ring
    (1, 64, 0, 20, 20, 64, 3, 1, 1, 1),   # reflect fwd + dgrad with images
    (1, 64, 0, 18, 17, 72, 3, 1, 0, 2),   # zero pad, ragged tile edges
    (1, 40, 0, 34, 36, 64, 5, 2, 1, 1),   # stride-2 dgrad: parity-class sub-grids of 17 x 18
    # 64-channel blocks on 256-pixel tiles keep ONE patch buffer (two blocks per CU): several chunks -> the chunk switch re-stages it in place
    (1, 128, 0, 20, 20, 64, 3, 1, 1, 1),  # two chunks, reflect
    (1, 192, 0, 17, 33, 40, 3, 1, 0, 2),  # three chunks, zero pad, ragged tiles, N = 40
    (1, 128, 0, 16, 32, 128, 1, 1, 1, 0), # 1x1, 128 -> 128: two 64-channel blocks per tile (grid y), two chunks, one patch buffer
    # maps of several tiles per side (the split into image-free interior tiles + border frame needs a chip-filling grid: LARGE_GRID_CASES)
    (1, 64, 0, 64, 96, 64, 3, 1, 1, 1),   # stride 1, 4 x 6 tiles of 16 x 16
    (1, 8, 0, 128, 160, 64, 3, 2, 1, 1),  # stride 2: per parity class 4 x 5 tiles
    # >= 256 output channels on a map >= 16 rows: 256-channel blocks (each wave 64 px x 128 channels)
    (1, 64, 0, 16, 20, 264, 3, 1, 0, 2),  # forward N = 264 (ragged second block); dgrad N = 64
    (1, 256, 0, 16, 16, 64, 3, 1, 1, 1),  # dgrad N = 256, four chunks forward
    # ... forwards on a grid that leaves most CUs empty (single-image inference: G.dec1 / dec2, enc4 / enc5): the K loop's chunks split over blocks (split-K)
    (1, 128, 64, 20, 20, 136, 3, 1, 1, 1),  # three chunks across two sources in two parts (2 + 1), ragged third block and tiles, reflect
    (1, 256, 0, 18, 34, 72, 3, 2, 1, 1),    # stride 2, four chunks in four parts, ragged tiles (9 x 17 outputs) and second block
    # stride-2 forwards with >= 64 input channels: input parity classes on the patch structure (conv_s2.hip); the dgrads are the class dgrads above
    (2, 64, 0, 32, 32, 128, 3, 2, 1, 1),  # enc3-like 3x3: classes of 2x2 / 2x1 / 1x2 / 1x1 taps, one tile per image
    (1, 128, 0, 34, 70, 72, 3, 2, 1, 1),  # two chunks, ragged tiles (17 x 35 outputs), N = 72
    (1, 64, 0, 33, 47, 128, 7, 2, 1, 1),  # d3-like 7x7: 4x4 / 4x3 / 3x4 / 3x3 taps, odd input sizes
    (2, 128, 0, 20, 36, 136, 5, 2, 1, 1), # d4-like 5x5, two N blocks (128 + ragged 8)
    (1, 64, 0, 40, 40, 64, 3, 2, 0, 2),   # zero padding, N = 64
    # ... and with 32 input channels (bf16): pixel-pair rows, the two row classes as phases (D.d2, G.enc2)
    (2, 32, 0, 32, 32, 64, 3, 2, 1, 1),   # enc2-like 3x3
    (1, 32, 0, 33, 47, 64, 7, 2, 1, 1),   # d2-like 7x7, odd input sizes: the last tap pair is half empty
    (1, 32, 0, 40, 72, 72, 5, 2, 0, 2),   # 5x5, zero padding, N = 72 (two 64-channel blocks, the second ragged)
    # 1x1 convs with >= 64 channels (attention fuse conv, decoder upsample convs): the patch kernel as a plain GEMM (the patch is the tile)
    (2, 64, 0, 9, 20, 128, 1, 1, 1, 0),   # C = 64 -> N = 128 forward, dgrad C = 128 -> N = 64; ragged tiles, reflect flag with pad 0
    (1, 128, 0, 16, 33, 64, 1, 1, 1, 0),  # two chunks forward, 16-row tiles, three tiles wide
    (1, 256, 0, 18, 16, 264, 1, 1, 1, 0), # 256-channel blocks + ragged second block
]

# Variants the launcher only picks when the grid covers the chip (>= 256 blocks): reached on emulator-sized maps by
# dropping the small-grid threshold (uegan_set_tuning(UEGAN_TUNE_SMALL_GRID)).
LARGE_GRID_CASES = [
    # 65..128 output channels on a map >= 32 rows: 32 x 16 tiles, one patch buffer reloaded per 64-channel chunk
    (1, 128, 0, 36, 20, 128, 3, 1, 0, 2),   # forward and (zero-pad) dgrad both N = 128, ragged tile rows and columns
    (1, 64, 32, 40, 16, 72, 3, 1, 1, 1),    # two inputs, N = 72 forward (tall tiles); reflect dgrad keeps 16-row tiles
    (2, 128, 0, 32, 32, 128, 4, 1, 0, 0),   # 4x4 taps (KB = 4 LDS budget), batch 2
    # stride-2 forwards: the 16 x 16 tile x 128-channel variant (the default on these map sizes is the 8 x 16 x 64 small-grid one)
    (2, 64, 0, 32, 32, 128, 3, 2, 1, 1),
    (1, 64, 0, 19, 35, 136, 7, 2, 1, 1),
    (1, 128, 0, 20, 36, 72, 5, 2, 0, 2),
    # reflect dgrads split into the image-free tile rectangle + the frame with the mirrored images (two launches; only when the rectangle alone fills the chip)
    (1, 64, 0, 64, 96, 64, 3, 1, 1, 1),     # stride 1, 4 x 6 tiles of 16 x 16: rectangle 2 x 4
    (1, 8, 0, 128, 160, 64, 3, 2, 1, 1),    # stride 2, per parity class 4 x 5 tiles: no far mirror (the forward stops short of the bottom / right padding): rectangle 3 x 4
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", LARGE_GRID_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_large_grid_variants(monkeypatch, backend, dtype, case):
    set_tuning("SMALL_GRID", 0)
    set_tuning("FOLD_MAX", 0)
    _conv_case(backend, dtype, case)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_fwd_dgrad_wgrad(backend, dtype, case):
    _conv_case(backend, dtype, case)


# The fp16-format build (libuegan_hip_f16.so / the emulator's libuegan_emu_f16.so: the same sources with -DUEGAN_HALF_FP16) through one case of
# every kernel family: generic gather-GEMM, patch (+ mirrored images), stride-2 forward, VALU heads, streaming, Toeplitz, one-wave-per-SIMD,
# transpose-read weight gradient -- forward, data gradient, weight gradient at fp16's tolerance (3e-3 where bf16 needs 2e-2)
F16_CASES = [
    (1, 8, 0, 12, 12, 8, 3, 1, 1, 1),            # generic kernel, reflect + LeakyReLU
    (1, 72, 0, 9, 20, 16, 3, 1, 1, 1),           # patch kernel, several chunks x reflected images
    (2, 64, 0, 32, 32, 128, 3, 2, 1, 1),         # stride-2 forward by parity classes, class data gradients
    (1, 16, 0, 8, 8, 1, 5, 1, 1, 3),             # prediction head on the vector ALU (v_dot2_f32_f16), tanh
    (1, 32, 32, 18, 34, 32, 3, 1, 1, 1),         # streaming kernel, two sources / two destinations
    (2, 32, 0, 40, 36, 3, 7, 1, 1, 3),           # Toeplitz kernel (G.dec5.1)
    (1, 64, 0, 17, 33, 128, 3, 1, 0, 2),         # conv_tall_kernel (needs its minimum grid lowered)
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", F16_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_fp16_storage_format(backend, case):
    set_tuning("TALL_MIN_GRID", 1)
    _conv_case(backend, torch.float16, case)


# ELEMENT-wise checks of the 16-bit-only kernels that carry the timed step: conv_tall_kernel, conv_stream_kernel,
# conv_flat_kernel, conv_s2fwd_kernel and wgrad_tr_kernel against F.conv2d on identically rounded operands.  No activation (act = 0): the
# result is then linear in the operands and every element can be bounded by output rounding + fp32 accumulation order alone.
ELEM_CASES = [
    ("conv_tall_kernel", (1, 64, 0, 17, 33, 128, 3, 1, 0, 0)),         # 128-channel blocks forward, 64-channel blocks in the data gradient; wgrad_tr 3x3
    ("conv_tall_kernel", (2, 128, 0, 32, 40, 128, 3, 1, 0, 0)),        # four chunks, 2 x 2 tiles
    ("conv_tall_kernel", (1, 64, 64, 20, 36, 64, 3, 1, 1, 0)),         # two sources, reflection padding: MODE 2 data gradient (mirrored images folded into the operand: one more rounding)
    ("conv_stream_kernel", (1, 32, 32, 18, 34, 32, 3, 1, 1, 0)),       # G.dec4 shape: streaming kernel, two sources / two destinations
    ("conv_stream_kernel", (2, 32, 0, 24, 40, 32, 3, 1, 1, 0)),        # G.dec5.0 shape
    ("conv_flat_kernel", (2, 128, 0, 20, 36, 64, 5, 2, 1, 0)),         # D.d4 shape: stride-2 forward by parity classes, flat data gradient + fold, wgrad_tr 5x5
    ("conv_flat_kernel", (1, 64, 0, 24, 40, 32, 7, 2, 1, 0)),          # D.d3 shape
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("kc", ELEM_CASES, ids=lambda kc: kc[0] + "-" + "x".join(map(str, kc[1])))
def test_16bit_kernels_elementwise(backend, dtype, kc):
    import ctypes
    kernel, case = kc
    set_tuning("TALL_MIN_GRID", 1)
    use_backend(backend)
    ops.set_compute_dtype(dtype)
    lib = _lib.load()
    _lib.check(lib.uegan_profile_begin(64))
    _conv_case(backend, dtype, case, elem=True)
    ents = (_lib.ProfileEntry * 16)()
    n = ctypes.c_int(0)
    _lib.check(lib.uegan_profile_end(ents, 16, ctypes.byref(n)))
    names = [ents[i].name.decode() for i in range(n.value)]
    assert any(nm.startswith(kernel) for nm in names), names
    assert any(nm.startswith("wgrad_tr_kernel") for nm in names), names


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", [(1, 16, 0, 8, 8, 1, 5, 1, 1, 3), (2, 72, 0, 10, 34, 1, 5, 1, 1, 3), (1, 40, 0, 9, 33, 1, 7, 1, 1, 3)],
                         ids=lambda c: "x".join(map(str, c)))
def test_head_fwd_one_thread_per_pixel(backend, dtype, case, monkeypatch):
    """Single-output heads on small maps default to 4 threads per pixel; this pins the 8 x 32-tile variant large maps use."""
    set_tuning("HEADS_NO_CG", 1)
    _conv_case(backend, dtype, case)


# reflection-padded dgrad through the padded grid + fold_reflect_kernel (the default for maps <= 128 x 128)
FOLD_CASES = [
    (1, 8, 0, 12, 12, 8, 3, 1, 1, 1),       # 3x3 pad 1
    (2, 16, 8, 9, 13, 24, 3, 1, 1, 1),      # two destinations (virtual concat), odd sizes
    (1, 8, 0, 16, 20, 16, 7, 1, 1, 0),      # 7x7 pad 3
    (1, 16, 0, 16, 16, 32, 3, 2, 1, 1),     # stride 2, even map
    (1, 8, 0, 17, 19, 16, 7, 2, 1, 1),      # stride 2, odd map: padded rows no tap reaches
    (1, 16, 0, 12, 12, 16, 5, 2, 1, 1),     # 5x5 stride 2 (D.d4/d5 shape)
    (1, 64, 0, 40, 36, 64, 3, 1, 1, 1),     # patch kernel over the padded grid
    (1, 8, 0, 4, 5, 8, 7, 1, 1, 0),         # pad 3 on a 4 x 5 map: a pixel mirrored across both borders
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", FOLD_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_dgrad_pad_grid_fold(backend, dtype, case, monkeypatch):
    set_tuning("FOLD_MAX", 16384)
    B, C1, C2, H, W, Co, k, s, pm, act = case
    _conv_case(backend, dtype, case)
    import ctypes
    x1 = torch.empty(B, H, W, ops.cpad(C1, dtype), dtype=dtype)
    x2 = torch.empty(B, H, W, ops.cpad(C2, dtype), dtype=dtype) if C2 else None
    d = ops._desc(x1, x2, torch.empty(Co, C1 + C2, k, k), ops.ConvCfg(s, pm, act))
    assert ops.lib().uegan_conv2d_dgrad_workspace_bytes(ctypes.byref(d)) > 0       # (the fold route was the one taken)


def _conv_case(backend, dtype, case, elem=False):
    dev = use_backend(backend)
    B, C1, C2, H, W, Co, k, s, pm, act = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(B, C1 + C2, H, W, generator=g)
    w = torch.randn(Co, C1 + C2, k, k, generator=g) * (1.0 / (k * (C1 + C2) ** 0.5))
    b = torch.randn(Co, generator=g)
    if dtype != torch.float32:
        x, w = half_round(x, dtype), half_round(w, dtype)
        ops.set_compute_dtype(dtype)              # (selects the bf16- or fp16-format build of the library; conftest.py resets it)
    x.requires_grad_(True), w.requires_grad_(True), b.requires_grad_(True)
    y = ref_conv(x, w, b, s, pm, act)
    r = torch.randn(y.shape, generator=g)
    if dtype != torch.float32:
        r = half_round(r, dtype)
    (y * r).sum().backward()

    def padc(t):        # NHWC tensors are carried with channels zero-padded to one 16-byte chunk
        cp = ops.cpad(t.shape[-1], dtype)
        return F.pad(t, (0, cp - t.shape[-1])).contiguous()

    xn = nhwc(x).to(dtype).to(dev)
    x1 = padc(xn[..., :C1]).requires_grad_(True)
    x2 = padc(xn[..., C1:]).requires_grad_(True) if C2 else None
    w2 = w.detach().clone().to(dev).requires_grad_(True)
    b2 = b.detach().clone().to(dev).requires_grad_(True)
    y2 = ops.conv2d(x1, x2, w2, b2, ops.ConvCfg(s, pm, act))
    assert y2.dtype == dtype and tuple(y2.shape) == (B, y.shape[2], y.shape[3], ops.cpad(Co, dtype))
    assert float(y2[..., Co:].abs().sum()) == 0.0                 # padding channels stay exactly zero
    y2.backward(padc(nhwc(r).to(dtype).to(dev)))
    gx = torch.cat([x1.grad[..., :C1].float()] + ([x2.grad[..., :C2].float()] if C2 else []), -1)
    y2 = y2[..., :Co]
    tol = F32_TOL if dtype == torch.float32 else (BF16_TOL if dtype == torch.bfloat16 else F16_TOL)
    if dtype == torch.float16 and act == 3:
        tol = 1e-2      # tanh' = 1 - y^2 is taken from the STORED (rounded) output: near saturation the rounding of y is a large part of 1 - y^2
    if elem:
        p_ = (k - 1) // 2
        # ELEMENT-wise (a max-norm ratio cannot see an error on a small-magnitude element).  Operands are identical 16-bit values
        # on both sides, so what may differ is the rounding of the stored result (half an ulp: 2^-11 relative in fp16, 2^-8 in bf16) and the
        # fp32 accumulation order (absolute, ~1e-6 of the tensor's scale per sqrt(K) terms); weight / bias gradients are fp32 sums: order only.
        ulp = {torch.float16: 2.0 ** -11, torch.bfloat16: 2.0 ** -8}[dtype]

        def elem_ok(what, got, ref, rel_t, abs_t, ring=0):
            got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
            bound = rel_t * ref.abs() + abs_t * ref.abs().max()
            if ring:
                # pixels within `ring` of a border of a REFLECTION-padded layer's data gradient are sums of up to 3 x 3 mirrored contributions that
                # meet as ROUNDED 16-bit values (operand fold of conv_tall's MODE 2, the padded-grid workspace + fold_reflect_kernel, the streaming
                # kernel's row fix-up): a few half-ulps of the LARGEST term, not of the (possibly cancelling) sum
                loose = torch.zeros_like(bound, dtype=torch.bool)
                loose[..., :ring, :] = True; loose[..., -ring:, :] = True; loose[..., :, :ring] = True; loose[..., :, -ring:] = True
                bound = torch.where(loose, bound + 3.0 * ulp * ref.abs().max(), bound)
            bad = (got - ref).abs() > bound
            assert not bool(bad.any()), (what, int(bad.sum()), float(((got - ref).abs() / bound).max()))
        assert act == 0
        elem_ok("y", nchw(y2), y, 1.01 * ulp, 2e-5)
        elem_ok("dx", nchw(gx), x.grad, 1.01 * ulp, 6e-5, ring=(p_ + 1) if (pm == ops.PAD_REFLECT and p_ > 0) else 0)
        elem_ok("dw", w2.grad, w.grad, 0.0, 3e-5)
        elem_ok("db", b2.grad, b.grad, 0.0, 3e-5)
        return
    assert rel(nchw(y2), y) < tol
    assert rel(nchw(gx), x.grad) < tol
    assert rel(w2.grad, w.grad) < tol
    assert rel(b2.grad, b.grad) < tol


# bf16 wide layers (>= 256 output channels): conv_wide.hip -- one wave per SIMD, 32x32x16 fragments.  (B, C1, C2, H, W, Cout, k, stride,
# pad_mode, act, launches of conv_wide_kernel expected in fwd + dgrad); UEGAN_TUNE_WIDE_MIN_GRID = 1 drops the minimum grid size
WIDE_CASES = [
    (1, 64, 0, 8, 32, 256, 3, 1, 0, 2, 1),       # one tile, one 64-channel chunk; the 64-channel data gradient is the patch kernel's
    (1, 256, 0, 9, 33, 256, 3, 1, 0, 2, 2),      # four chunks, ragged tile rows and columns; forward and zero-padded data gradient
    (2, 128, 0, 16, 64, 512, 3, 1, 1, 1, 1),     # reflection-padded forward (two channel blocks, 2 x 2 tiles, batch 2), LeakyReLU
    (1, 128, 0, 24, 40, 256, 3, 1, 0, 0, 1),     # no activation, tiles overhanging to the right
    # reflection-padded data gradient with N = 128 + 128 = 256 (two destinations) from 64 dz channels: interior (64 x 96 pixels) of the 96 x 128 map on
    # conv_wide_kernel (conv_interior_run), frame on the patch kernel; the 64-channel forward is conv_tall_kernel's
    (1, 128, 128, 96, 128, 64, 3, 1, 1, 1, 1),
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", WIDE_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_wide_kernel(backend, case, monkeypatch):
    import ctypes
    set_tuning("WIDE_MIN_GRID", 1)
    set_tuning("TALL_MIN_GRID", -1)      # (conv_tall_kernel has the first pick)
    use_backend(backend)
    lib = _lib.load()
    _lib.check(lib.uegan_profile_begin(64))
    _conv_case(backend, torch.bfloat16, case[:10])
    ents = (_lib.ProfileEntry * 16)()
    n = ctypes.c_int(0)
    _lib.check(lib.uegan_profile_end(ents, 16, ctypes.byref(n)))
    launches = sum(ents[i].launches for i in range(n.value) if ents[i].name.decode().startswith("conv_wide_kernel"))
    assert launches == case[10], [(ents[i].name.decode(), ents[i].launches) for i in range(n.value)]


# bf16 3x3 stride-1 layers with 64 / 128 output channels: conv_tall_kernel (conv_wide.hip) -- one wave per SIMD, the four waves of a block split a
# 16 x 32-pixel tile and share the weight slices; 32-channel patch chunks; two-source forwards.  (B, C1, C2, H, W, Cout, k, stride, pad_mode, act,
# launches of conv_tall_kernel expected in fwd + dgrad); UEGAN_TUNE_TALL_MIN_GRID = 1 drops the minimum grid size
TALL_CASES = [
    (1, 64, 0, 16, 32, 64, 3, 1, 0, 2, 2),       # one tile, two chunks, 64-channel blocks (VGG conv1_2); forward and zero-padded data gradient
    (1, 64, 0, 17, 33, 128, 3, 1, 0, 2, 2),      # 128-channel blocks forward (conv2_1), 64-channel blocks in the data gradient; ragged tiles
    (2, 128, 0, 32, 40, 128, 3, 1, 0, 0, 2),     # four chunks, 2 x 2 tiles, batch 2, no activation
    # two sources, reflection padding, LeakyReLU (G.dec3 forward); the reflection-padded data gradient into TWO destinations (virtual concat, N = 64 + 64)
    # is MODE 2 of the same kernel: every tile in one launch, the mirrored images folded into the pixel operand.  20 x 36: ragged tiles, rows 1 and
    # H-2 in different waves, columns 1 and W-2 in different tile columns
    (1, 64, 64, 20, 36, 64, 3, 1, 1, 1, 2),
    (1, 128, 128, 16, 64, 128, 3, 1, 1, 1, 2),   # G.dec2: two sources of 128, eight chunks; data gradient with two channel blocks (N = 256)
    (2, 64, 64, 96, 128, 64, 3, 1, 1, 1, 2),     # G.dec3-like map of 12 x 4 tiles, batch 2: interior tiles take the scalar branch around the folds
    (1, 128, 0, 16, 32, 128, 3, 1, 1, 0, 2),     # one tile column holding columns 1 and W-2 (both x mirrors in every fragment), corners in the first / last wave
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("rpw", [2, 4])
@pytest.mark.parametrize("case", TALL_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_tall_kernel(backend, case, rpw):
    """rpw: tile rows per wave -- 2 (8-row tiles, two blocks per CU: the default below 512 input channels) or 4 (16-row tiles)"""
    import ctypes
    set_tuning("TALL_RPW", rpw)
    set_tuning("TALL_MIN_GRID", 1)
    set_tuning("FOLD_MAX", 0)
    use_backend(backend)
    lib = _lib.load()
    _lib.check(lib.uegan_profile_begin(64))
    _conv_case(backend, torch.bfloat16, case[:10])
    ents = (_lib.ProfileEntry * 16)()
    n = ctypes.c_int(0)
    _lib.check(lib.uegan_profile_end(ents, 16, ctypes.byref(n)))
    launches = sum(ents[i].launches for i in range(n.value) if ents[i].name.decode().startswith("conv_tall_kernel"))
    assert launches == case[10], [(ents[i].name.decode(), ents[i].launches) for i in range(n.value)]


# Forward convolution + InstanceNorm moments from the streaming kernel's epilogue (uegan_conv2d_fwd_stats).  (B, C, H, W, Cout, k, pad_mode, act)
STATS_CASES = [
    (2, 32, 64, 64, 32, 1, 1, 0),        # G.ga1 shape: 1x1, 32 -> 32; blocks that span two images
    (1, 64, 32, 96, 64, 1, 1, 0),        # G.ga2 shape: 64 -> 64
    (3, 3, 48, 64, 64, 3, 0, 2),         # VGG conv1_1 shape: 3 -> 64, zero padding, bias + ReLU, batch 3
    (1, 32, 40, 72, 32, 3, 1, 1),        # 3x3 reflect + LeakyReLU, ragged tile rows (40 = 2.5 tiles) and columns (72 = 4.5 tiles)
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("case", STATS_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_fwd_stats(backend, dtype, case):
    import ctypes
    dev = use_backend(backend)
    ops.set_compute_dtype(dtype)
    B, Cc, H, W, Co, k, pm, act = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = half_round(torch.randn(B, Cc, H, W, generator=g) + 0.5, dtype)
    w = half_round(torch.randn(Co, Cc, k, k, generator=g) * (1.0 / (k * Cc ** 0.5)), dtype)
    b = torch.randn(Co, generator=g) if act else None
    y = ref_conv(x, w, b, 1, pm, act)
    mean_r, var_r = y.mean((2, 3)), y.var((2, 3), unbiased=False)
    cp = ops.cpad(Cc, dtype)
    xn = F.pad(nhwc(x).to(dtype), (0, cp - Cc)).contiguous().to(dev)
    holder = ops.StatsHolder()
    with torch.no_grad():
        y2 = ops.conv2d(xn, None, w.to(dev), None if b is None else b.to(dev), ops.ConvCfg(1, pm, act), stats=holder)
    assert holder.value is not None, "the streaming kernel should have taken this layer"
    st = holder.value.cpu()
    tol = BF16_TOL if dtype == torch.bfloat16 else F16_TOL
    assert rel(nchw(y2[..., :Co]), y) < tol
    std_r = var_r.sqrt()
    assert float(((st[0, :, :Co] - mean_r).abs() / (std_r + 1e-3)).max()) < 2e-3            # moments of the fp32 accumulators: no storage rounding in them
    assert float(((st[1, :, :Co] - 1.0 / (var_r + 1e-5).sqrt()).abs() * (var_r + 1e-5).sqrt()).max()) < 2e-3
    # ... and the normalising pass with the given moments = InstanceNorm of the stored tensor
    with torch.no_grad():
        z = ops.instnorm(y2, holder.value)
    zr = F.instance_norm(nchw(y2[..., :Co].float().cpu()))
    assert rel(nchw(z[..., :Co]), zr) < tol
    # the knob declines: plain forward, no moments
    set_tuning("FWD_STATS", 0)
    with torch.no_grad():
        ops.conv2d(xn, None, w.to(dev), None if b is None else b.to(dev), ops.ConvCfg(1, pm, act), stats=holder)
    assert holder.value is None


# Data gradients left on the PADDED grid of their reflection-padded conv (uegan_conv2d_dgrad_padded: head_dgrad_mfma_kernel for the one-channel
# prediction heads, conv_flat_kernel for the stride-2 trunk layers) + the activation backwards that add the mirror images while they read them
# (uegan_act_bwd_p, uegan_sn_act_bwd_p through the fused discriminator pass: tests/test_fused.py, tests/test_oracle_at_size.py).
# (B, C, H, W, Cout, k, stride)
PADDED_CASES = [
    (2, 64, 24, 40, 1, 7, 1),        # D.d2 head: 64 -> 1, 7x7; two tile columns (the second ragged), 4 tile rows
    (1, 128, 16, 16, 1, 7, 1),       # D.d3 head on a map narrower than the tile
    (3, 256, 9, 12, 1, 5, 1),        # D.d4 head 5x5, two 128-channel blocks, every pixel within 2 of a border
    (1, 512, 8, 8, 1, 5, 1),         # D.d5 head shape
    (2, 32, 20, 36, 1, 7, 1),        # D.d1 head: 32 channels (one channel fragment per block)
    (2, 128, 20, 36, 64, 5, 2),      # D.d4 trunk: stride 2 on conv_flat_kernel, consumer folds pad 2
    (1, 64, 24, 40, 32, 7, 2),       # D.d3 trunk: 7x7 stride 2
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("case", PADDED_CASES, ids=lambda c: "x".join(map(str, c)))
def test_dgrad_on_padded_grid_and_folding_activation_backward(backend, dtype, case):
    import ctypes
    dev = use_backend(backend)
    ops.set_compute_dtype(dtype)
    lib = _lib.load()
    B, Cc, H, W, Co, k, st = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    a = half_round(torch.randn(B, Cc, H, W, generator=g), dtype)                 # the activated trunk tensor (conv input); its sign drives LeakyReLU'
    w = half_round(torch.randn(Co, Cc, k, k, generator=g) * (1.0 / (k * Cc ** 0.5)), dtype)
    p = (k - 1) // 2
    xr = a.clone().requires_grad_(True)
    y = F.conv2d(F.pad(xr, (p, p, p, p), mode="reflect"), w, None, stride=st)
    dzr = half_round(torch.randn(y.shape, generator=g), dtype)
    (y * dzr).sum().backward()
    g2 = half_round(torch.randn(B, Cc, H, W, generator=g), dtype)                # the activation's other consumer
    ref = (xr.grad + g2) * torch.where(a > 0, torch.ones_like(a), torch.full_like(a, 0.2))

    def padc(t):
        cp = ops.cpad(t.shape[-1], dtype)
        return F.pad(t, (0, cp - t.shape[-1])).contiguous()

    an = nhwc(a).to(dtype).to(dev).contiguous()
    dzn = padc(nhwc(dzr).to(dtype)).to(dev)
    g2n = nhwc(g2).to(dtype).to(dev).contiguous()
    cfg = ops.ConvCfg(st, ops.PAD_REFLECT, ops.ACT_NONE)
    d = ops._desc(an, None, w, cfg)
    ohwi, ihwo = cfg.packed.get(w.to(dev), dtype, d.C1, d.Cout)
    nbytes = lib.uegan_conv2d_dgrad_padded_bytes(ctypes.byref(d))
    assert nbytes == B * (H + 2 * p) * (W + 2 * p) * Cc * 2
    ws = torch.full((B, H + 2 * p, W + 2 * p, Cc), float("nan"), dtype=dtype, device=dev)      # (every element of the padded grid must be written)
    pad = ctypes.c_int(-1)
    _lib.check(lib.uegan_conv2d_dgrad_padded(ctypes.byref(d), ops._p(dzn), ops._p(ihwo), ops._p(ohwi), None, ops._p(ws), nbytes, ctypes.byref(pad), None))
    assert pad.value == p, pad.value
    assert not bool(torch.isnan(ws.float()).any())
    out = torch.empty_like(an)
    _lib.check(lib.uegan_act_bwd_p(ops._dt(an), ops.ACT_LRELU, ops._p(ws), p, ops._p(g2n), 0, ops._p(an), ops._p(out), B, H, W, Cc, None))
    tol = BF16_TOL if dtype == torch.bfloat16 else F16_TOL
    assert rel(nchw(out), ref) < tol
    # the other argument order (plain first, padded second) and the knob that declines the heads
    _lib.check(lib.uegan_act_bwd_p(ops._dt(an), ops.ACT_LRELU, ops._p(g2n), 0, ops._p(ws), p, ops._p(an), ops._p(out), B, H, W, Cc, None))
    assert rel(nchw(out), ref) < tol
    # the spectral-norm form of the same backward (one group, sigma = 1, zero bias): its non-ring / ring block split must give the same dz
    B2 = B if (B % 2) else B // 2
    ng = B // B2
    snws = torch.empty(lib.uegan_sn_act_bwd_workspace_floats(ng, Cc), dtype=torch.float32, device=dev)
    out2 = torch.full_like(an, float("nan"))
    inv = torch.ones(ng, dtype=torch.float32, device=dev)
    bias = torch.zeros(Cc, dtype=torch.float32, device=dev)
    nbx = lib.uegan_sn_act_bwd_p(ops._dt(an), ops.ACT_LRELU, ops._p(ws), p, ops._p(g2n), 0, ops._p(an), ops._p(bias), Cc, ops._p(inv), ops._p(out2), ops._p(snws),
                                 B2 * H * W, H, W, Cc, ng, None)
    assert nbx > 0, nbx
    assert torch.equal(out2.float().cpu(), out.float().cpu())
    # ... and its bias-gradient partials add up to the sum of the raw gradient over the pixels
    db = snws[ng * 256: ng * 256 + ng * nbx * Cc].view(ng * nbx, Cc).sum(0).cpu()        # partials: [group * nbx + block][C] behind the 256 x groups coefficient slots
    assert rel(db, ref.sum((0, 2, 3))) < tol
    if Co == 1:
        set_tuning("HEADS_MFMA", 0)
        _lib.check(lib.uegan_conv2d_dgrad_padded(ctypes.byref(d), ops._p(dzn), ops._p(ihwo), ops._p(ohwi), None, ops._p(ws), nbytes, ctypes.byref(pad), None))
        assert pad.value == -1


# bf16 stride-2 data gradients with 64 / 128 k input channels: conv_flat_kernel (conv_flat.hip) -- all four parity classes of the PADDED grid in one
# launch over flattened positions, then fold_reflect_kernel.  (B, C1, C2, H, W, Cout, k, stride, pad_mode, act)
FLAT_CASES = [
    (2, 64, 0, 32, 32, 128, 7, 2, 1, 1),         # 64-channel blocks, four chunks; tiles across the image boundary
    (1, 128, 0, 20, 36, 64, 5, 2, 1, 1),         # D.d4 shape: 5x5, 128-channel blocks, two chunks, class grid 12 x 20
    (3, 256, 0, 12, 12, 96, 5, 2, 1, 0),         # D.d5 shape: two channel blocks per class, three chunks, 8 x 8 class grids, batch 3, last tile ragged
    (1, 64, 0, 24, 40, 32, 7, 2, 1, 1),          # D.d3 shape: 7x7, classes of 16 / 12 / 12 / 9 taps, one chunk
    (2, 128, 0, 16, 64, 160, 3, 2, 1, 1),        # G.enc4 shape: 3x3 on 128-channel blocks (classes of 4 / 2 / 2 / 1 taps, padded to 4 steps with slices of zeros), five chunks
    # 32 input channels (D.d2): a block takes both column classes of its row class as its two channel fragments
    (2, 32, 0, 32, 48, 64, 7, 2, 1, 1),          # flattened positions, two chunks
    (1, 32, 0, 16, 288, 64, 7, 2, 1, 1),         # a map too wide for the flattened patch: 8 x 32 tiles of the class grid (D.d2 at 512^2), ragged last tile column / row
    (3, 32, 0, 20, 36, 96, 5, 2, 1, 0),          # 5x5, three chunks, batch 3
    # 2-D tiles with one class per block (the 1024^2 configuration's d3 / d4)
    (1, 128, 0, 16, 160, 64, 5, 2, 1, 1),
    (1, 64, 0, 16, 272, 64, 7, 2, 1, 0),
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("case", FLAT_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_flat_kernel(backend, dtype, case):
    import ctypes
    use_backend(backend)
    ops.set_compute_dtype(dtype)
    lib = _lib.load()
    _lib.check(lib.uegan_profile_begin(64))
    _conv_case(backend, dtype, case)
    ents = (_lib.ProfileEntry * 16)()
    n = ctypes.c_int(0)
    _lib.check(lib.uegan_profile_end(ents, 16, ctypes.byref(n)))
    launches = sum(ents[i].launches for i in range(n.value) if ents[i].name.decode().startswith("conv_flat_kernel"))
    assert launches == 1, [(ents[i].name.decode(), ents[i].launches) for i in range(n.value)]
    # ... and the round-4 route (one launch per parity class) is still there behind the knob
    set_tuning("FLAT_S2", 0)
    _conv_case(backend, dtype, case)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("rpw", [2, 4])
@pytest.mark.parametrize("case", [(1, 64, 0, 16, 256, 256, 3, 1, 0, 2, 2), (2, 64, 0, 32, 64, 512, 3, 1, 1, 1, 1)], ids=lambda c: "x".join(map(str, c)))
def test_conv_tall_kernel_channel_blocks(backend, case, rpw):
    """256 / 512 output channels on conv_tall_kernel: N / 128 channel blocks per tile, XCD-aware (tile, channel block) order (grids that are
    multiples of 8).  Launches with >= 512 such blocks take this route by default (the VGG conv3_x / conv4_x layers); here conv_wide_kernel is
    switched off to reach it on small maps."""
    import ctypes
    set_tuning("TALL_RPW", rpw)
    set_tuning("WIDE_MIN_GRID", -1)
    set_tuning("TALL_MIN_GRID", 1)
    set_tuning("FOLD_MAX", 0)
    use_backend(backend)
    lib = _lib.load()
    _lib.check(lib.uegan_profile_begin(64))
    _conv_case(backend, torch.bfloat16, case[:10])
    ents = (_lib.ProfileEntry * 16)()
    n = ctypes.c_int(0)
    _lib.check(lib.uegan_profile_end(ents, 16, ctypes.byref(n)))
    launches = sum(ents[i].launches for i in range(n.value) if ents[i].name.decode().startswith("conv_tall_kernel"))
    assert launches == case[10], [(ents[i].name.decode(), ents[i].launches) for i in range(n.value)]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("rpw", [2, 4])
@pytest.mark.parametrize("case", [(1, 64, 32, 64, 64), (2, 64, 16, 40, 128)], ids=lambda c: "x".join(map(str, c)))
def test_conv_tall_kernel_pool_and_masked_dgrad(backend, case, rpw):
    """the POOL epilogue (conv + ReLU + 2x2 max-pool: VGG conv1_2 / conv2_2) and the MASK epilogue (deferred activation gradient of the
    producer in the data gradient) of conv_tall_kernel"""
    import ctypes
    set_tuning("TALL_RPW", rpw)
    set_tuning("TALL_MIN_GRID", 1)
    dev = use_backend(backend)
    lib = _lib.load()
    B, Cc, H, W, Co = case
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(5 + Co)
    x = torch.randn(B, H, W, Cc, generator=g).relu().to(dtype)
    w = torch.randn(Co, Cc, 3, 3, generator=g) * 0.05
    b = torch.randn(Co, generator=g)
    cfg = ops.ConvCfg(1, ops.PAD_ZERO, ops.ACT_RELU)
    _lib.check(lib.uegan_profile_begin(16))
    y, d, _, yp = ops.raw_conv_fwd(x.to(dev), None, w.to(dev), b.to(dev), cfg, pool=True)
    ents = (_lib.ProfileEntry * 16)()
    n = ctypes.c_int(0)
    _lib.check(lib.uegan_profile_end(ents, 16, ctypes.byref(n)))
    names = [ents[i].name.decode() for i in range(n.value)]
    assert names and all(nm.startswith("conv_tall_kernel") for nm in names), names
    ref = ops.maxpool2x2(y)
    assert yp.shape == ref.shape and torch.equal(yp, ref)                # pooled in the epilogue == pooling the stored tensor, bit for bit
    yt = F.relu(F.conv2d(nchw(x.float()), bf16_round(w), b, padding=1))
    assert rel(nchw(y[..., :Co]), yt) < BF16_TOL
    # uegan_conv2d_fwd_pool_part: only the first n_full images need the full-resolution tensor (the others feed nothing but the pool: the
    # fidelity loss's reference images) -- same y[:n_full], same pooled tensor for ALL images, and this kernel leaves y[n_full:] alone
    for n_full in range(B + 1):
        yq, _, _, ypq = ops.raw_conv_fwd(x.to(dev), None, w.to(dev), b.to(dev), cfg, pool=True, n_full=n_full)
        assert torch.equal(ypq, yp) and torch.equal(yq[:n_full], y[:n_full])
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
    d0 = ops._desc(xd, None, wd, cfg)
    ohwi, _ = cfg.packed.get(wd, dtype, Cc, Co, None, cfg.cin_used)
    marker = torch.full((B, H, W, Co), 7.0, dtype=dtype, device=dev)
    yp2 = torch.empty_like(yp)
    _lib.check(lib.uegan_conv2d_fwd_pool_part(ctypes.byref(d0), ops._p(xd), None, ops._p(ohwi), ops._p(bd), None, ops._p(marker), ops._p(yp2), 1, ops._stream()))
    assert torch.equal(yp2, yp) and torch.equal(marker[:1], y[:1]) and bool((marker[1:] == 7.0).all())
    # uegan_conv2d_fwd_pool_idx: the window positions of the maxima (first n_idx images) from the same epilogue, no full-resolution tensor at all;
    # uegan_maxpool2x2_bwd_idx routes the gradient with them and the pooled tensor exactly as uegan_maxpool2x2_bwd_act does with y
    yq, _, _, ypq, idx = ops.raw_conv_fwd(xd, None, wd, bd, cfg, pool=True, n_full=0, n_idx=B)
    assert torch.equal(ypq, yp) and idx.dtype == torch.uint8 and int(idx.max()) <= 3
    idx_ref = torch.empty_like(idx)
    yp3 = torch.empty_like(yp)
    _lib.check(lib.uegan_maxpool2x2_fwd_idx(1, ops._p(y), ops._p(yp3), ops._p(idx_ref), B, H, W, Co, ops._stream()))
    assert torch.equal(yp3, yp) and torch.equal(idx_ref, idx)
    gp = torch.randn(yp.shape, generator=g).to(dtype).to(dev)
    gx_ref, gx_idx = torch.empty_like(y), torch.empty_like(y)
    _lib.check(lib.uegan_maxpool2x2_bwd_act(1, ops.ACT_RELU, ops._p(y), ops._p(gp), ops._p(gx_ref), B, H, W, Co, ops._stream()))
    _lib.check(lib.uegan_maxpool2x2_bwd_idx(1, ops.ACT_RELU, ops._p(yp), ops._p(idx), ops._p(gp), ops._p(gx_idx), B, H, W, Co, ops._stream()))
    assert torch.equal(gx_idx, gx_ref)
    # masked data gradient: dx = dgrad(dz) * relu'(x)
    cfg2 = ops.ConvCfg(1, ops.PAD_ZERO, ops.ACT_NONE)
    cfg2.in_act = ops.ACT_RELU
    x1 = x.clone().to(dev).requires_grad_(True)
    w2 = w.clone().to(dev).requires_grad_(True)
    r = bf16_round(torch.randn(B, H, W, Co, generator=g))
    _lib.check(lib.uegan_profile_begin(16))
    y2 = ops.conv2d(x1, None, w2, None, cfg2)
    y2.backward(r.to(dtype).to(dev))
    _lib.check(lib.uegan_profile_end(ents, 16, ctypes.byref(n)))
    names = [ents[i].name.decode() for i in range(n.value)]
    assert sum(nm.startswith("conv_tall_kernel") for nm in names) == 2, names
    xt = nchw(x.detach().float()).clone().requires_grad_(True)
    (F.conv2d(xt, bf16_round(w), None, padding=1) * nchw(r)).sum().backward()
    assert rel(nchw(x1.grad.float()), xt.grad * (xt.detach() > 0)) < BF16_TOL


# Which kernel the DEFAULT thresholds pick at the benchmark's own shapes (16 x 512^2 step: VGG sees 32 images forward, 16 backward): the per-layer
# numbers in DESIGN.md / profiles/*_bench_conv.log are numbers of THESE kernels.  (layer, B, C1, C2, H, Cout, pad_mode, act, forward kernel, dgrad kernels)
DISPATCH_CASES = [
    ("VGG conv1_2", 4, 64, 0, 512, 64, 0, 2, "conv_tall_kernel<bf16,BN=64,KS=3,MODE=0>", ["conv_tall_kernel<bf16,BN=64,KS=3,MODE=1>"]),
    ("VGG conv3_2", 16, 256, 0, 128, 256, 0, 2, "conv_tall_kernel<bf16,BN=128,KS=3,MODE=0>", ["conv_tall_kernel<bf16,BN=128,KS=3,MODE=1>"]),
    ("VGG conv5_1", 16, 512, 0, 32, 512, 0, 2, "conv_tall_kernel<bf16,BN=128,KS=3,MODE=0>", ["conv_tall_kernel<bf16,BN=128,KS=3,MODE=1>"]),
    # reflection-padded, two sources: the whole data gradient (two destinations) in ONE launch of conv_tall_kernel, mirrored images folded into the pixel operand
    ("G.dec2", 16, 128, 128, 128, 128, 1, 1, "conv_tall_kernel<bf16,BN=128,KS=3,MODE=0>", ["conv_tall_kernel<bf16,BN=128,KS=3,MODE=2>"]),
    ("G.dec3", 16, 64, 64, 256, 64, 1, 1, "conv_tall_kernel<bf16,BN=64,KS=3,MODE=0>", ["conv_tall_kernel<bf16,BN=128,KS=3,MODE=2>"]),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", DISPATCH_CASES, ids=lambda c: c[0].replace(" ", "_"))
def test_default_dispatch_at_benchmark_shapes(case):
    import ctypes
    name, B, C1, C2, S, Co, pm, act, fwd_kernel, dgrad_kernels = case
    dev = use_backend("gpu")
    ops.set_compute_dtype(torch.bfloat16)
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    x1 = torch.randn(B, S, S, C1, generator=g).to(torch.bfloat16).to(dev).requires_grad_(True)
    x2 = torch.randn(B, S, S, C2, generator=g).to(torch.bfloat16).to(dev).requires_grad_(True) if C2 else None
    w = (torch.randn(Co, C1 + C2, 3, 3, generator=g) * 0.02).to(dev).requires_grad_(True)
    b = torch.zeros(Co, device=dev, requires_grad=True)

    def names_of(fn):
        _lib.check(lib.uegan_profile_begin(64))
        out = fn()
        ents = (_lib.ProfileEntry * 32)()
        n = ctypes.c_int(0)
        _lib.check(lib.uegan_profile_end(ents, 32, ctypes.byref(n)))
        return out, [ents[i].name.decode() for i in range(n.value)]

    y, fwd = names_of(lambda: ops.conv2d(x1, x2, w, b, ops.ConvCfg(1, pm, act)))
    assert fwd == [fwd_kernel], (name, fwd)
    _, bwd = names_of(lambda: y.backward(torch.ones_like(y)))
    convs = [k for k in bwd if k.startswith("conv_")]
    assert len(convs) == len(dgrad_kernels) and all(k.startswith(e) for k, e in zip(sorted(convs, reverse=True), sorted(dgrad_kernels, reverse=True))), (name, bwd)


# bf16 thin full-resolution layers: the persistent streaming kernel (conv_stream.h).  (B, C1, C2, H, W, Cout, k, pad_mode, act, launches of the kernel expected in fwd + dgrad[, stride])
STREAM_CASES = [
    (1, 32, 0, 20, 40, 32, 3, 1, 1, 2),      # dec5.0-like: reflect, fwd borders in-kernel, dgrad = zero-fill stream + mirrored-image fix-up
    (1, 3, 0, 24, 48, 32, 7, 1, 1, 2),       # enc1-like: 8-channel rows, one MFMA K step = 4 taps
    (1, 32, 32, 18, 34, 32, 3, 1, 1, 2),     # two sources (128-byte rows), dgrad into two destinations
    (2, 32, 0, 40, 36, 3, 7, 1, 3, 1),       # G head 32 -> 3, tanh: the forward is the Toeplitz kernel's (below); 8-channel dz rows in the data gradient
    (1, 16, 0, 16, 32, 16, 3, 0, 2, 2),      # zero padding, 16-channel rows
    (1, 32, 0, 33, 50, 32, 1, 1, 0, 2),      # 1x1, ragged sizes
    (1, 64, 0, 19, 35, 32, 3, 1, 1, 2),      # 64 -> 32 (dec4-like single source): data gradient with 64 output channels
    (1, 32, 0, 32, 64, 1, 7, 1, 3, 1),       # D head 32 -> 1 (forward: Toeplitz kernel)
    (2, 3, 0, 40, 72, 32, 7, 1, 1, 2, 2),    # d1-like: stride 2 forward, 8-channel rows; class dgrad on a map where every tile touches a border
    (1, 32, 0, 36, 66, 64, 3, 1, 1, 2, 2),   # enc2-like: stride 2 forward, 32 -> 64; class dgrad (64 dz channels -> 32) on the one-block 8-wave variant
    (1, 3, 0, 96, 160, 32, 7, 1, 1, 2, 2),   # d1-like at a size with interior tiles: stride-2 dgrad by parity classes (dz 32 ch)
    (1, 32, 0, 96, 128, 32, 3, 1, 1, 2, 2),  # stride-2 dgrad by parity classes with a 3x3 kernel (dz 32 channels)
    (1, 3, 0, 96, 160, 64, 3, 1, 1, 2, 2),   # class dgrad with 64 dz channels (two K steps per tap), 3 -> 8 padded outputs
    # widths that are whole 16-pixel tiles: the data gradient adds its x-mirrored images inside the streaming kernel (extra K steps on the
    # first / last tile column), the fix-up kernel only the y-mirrored rows
    (1, 32, 0, 20, 48, 32, 3, 1, 1, 2),      # 3x3, three tile columns
    (1, 32, 32, 18, 32, 32, 3, 1, 1, 2),     # two destinations, the first tile column is also next to the last
    (2, 32, 0, 40, 32, 3, 7, 1, 3, 1),       # 7x7 (pad 3): 8-channel dz rows, four taps per K step (forward: Toeplitz kernel)
    (1, 16, 0, 24, 64, 16, 5, 1, 0, 2),      # 5x5 (pad 2), 16-channel rows
    (1, 32, 0, 112, 192, 32, 5, 1, 0, 2, 2), # 5x5 stride 2 (pad 2): forward streams; the class dgrad needs the one-block class of the LDS: 8-wave variant
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", STREAM_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_stream_kernel(backend, case):
    import ctypes
    set_tuning("FLAT_S2", 0)          # (the 5x5 stride-2 case pins the streaming kernel's class data gradient -- G.enc2's route; with 32 input channels and a
                                      # 5x5 / 7x7 kernel conv_flat_kernel has the first pick since round 5: tests/test_ops.py::test_conv_flat_kernel)
    dev = use_backend(backend)
    lib = _lib.load()
    B, C1, C2, H, W, Co, k, pm, act, nlaunch = case[:10]
    stride = case[10] if len(case) > 10 else 1
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = bf16_round(torch.randn(B, C1 + C2, H, W, generator=g)).requires_grad_(True)
    w = bf16_round(torch.randn(Co, C1 + C2, k, k, generator=g) * (1.0 / (k * (C1 + C2) ** 0.5))).requires_grad_(True)
    b = torch.randn(Co, generator=g).requires_grad_(True)
    y = ref_conv(x, w, b, stride, pm, act)
    r = bf16_round(torch.randn(y.shape, generator=g))
    (y * r).sum().backward()

    def padc(t):
        cp = ops.cpad(t.shape[-1], dtype)
        return F.pad(t, (0, cp - t.shape[-1])).contiguous()

    xn = nhwc(x.detach()).to(dtype).to(dev)
    x1 = padc(xn[..., :C1]).requires_grad_(True)
    x2 = padc(xn[..., C1:]).requires_grad_(True) if C2 else None
    w2 = w.detach().clone().to(dev).requires_grad_(True)
    b2 = b.detach().clone().to(dev).requires_grad_(True)
    _lib.check(lib.uegan_profile_begin(64))
    y2 = ops.conv2d(x1, x2, w2, b2, ops.ConvCfg(stride, pm, act))
    y2.backward(padc(nhwc(r).to(dtype).to(dev)))
    ents = (_lib.ProfileEntry * 16)()
    n = ctypes.c_int(0)
    _lib.check(lib.uegan_profile_end(ents, 16, ctypes.byref(n)))
    names = [ents[i].name.decode() for i in range(n.value)]
    assert sum(nm.startswith("conv_stream_kernel") for nm in names) == nlaunch, names
    assert float(y2[..., Co:].abs().sum()) == 0.0
    gx = torch.cat([x1.grad[..., :C1].float()] + ([x2.grad[..., :C2].float()] if C2 else []), -1)
    assert rel(nchw(y2[..., :Co]), y) < BF16_TOL
    assert rel(nchw(gx), x.grad) < BF16_TOL
    assert rel(w2.grad, w.grad) < BF16_TOL


# bf16 stride-1 forwards with <= 4 output channels on 32 input channels: the Toeplitz kernel (conv_toep.hip).
# (B, C, H, W, Cout, k, pad_mode, act)
TOEP_CASES = [
    (2, 32, 40, 36, 3, 7, 1, 3),      # G.dec5.1: 7x7, 32 -> 3, tanh; ragged second tile column, three tile rows
    (1, 32, 32, 64, 1, 7, 1, 3),      # D head 32 -> 1
    (1, 32, 16, 32, 4, 3, 0, 0),      # 3x3, zero padding, all four channel slots, exactly one tile (wave 3 has no tap row)
    (2, 32, 33, 70, 2, 5, 1, 1),      # 5x5, LeakyReLU, ragged both ways
    (3, 32, 48, 96, 3, 7, 1, 3),      # more tiles than a block takes in one pass on the emulator's grid
    # round 5: more than 32 input channels, one 32-channel chunk at a time (the discriminator's prediction heads d2 - d5)
    (2, 64, 24, 40, 1, 7, 1, 3),      # D.d2 head: 64 -> 1, two chunks
    (3, 128, 16, 16, 1, 7, 1, 3),     # D.d3 head on a map narrower than the tile (overhanging tile columns)
    (2, 256, 9, 12, 1, 5, 1, 3),      # D.d4 head 5x5, eight chunks, map smaller than one tile both ways
    (1, 512, 8, 8, 2, 5, 1, 3),       # D.d5 head shape, sixteen chunks, two channel slots
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", TOEP_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_toeplitz_kernel(backend, case):
    import ctypes
    set_tuning("TOEP_HEADS", 2)       # (the default stops at 128 input channels)
    dev = use_backend(backend)
    lib = _lib.load()
    B, C, H, W, Co, k, pm, act = case
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = bf16_round(torch.randn(B, C, H, W, generator=g))
    w = bf16_round(torch.randn(Co, C, k, k, generator=g) * (1.0 / (k * C ** 0.5)))
    b = torch.randn(Co, generator=g)
    y = ref_conv(x, w, b, 1, pm, act)
    xn = nhwc(x).to(dtype).to(dev)
    _lib.check(lib.uegan_profile_begin(16))
    with torch.no_grad():
        y2 = ops.conv2d(xn, None, w.to(dev), b.to(dev), ops.ConvCfg(1, pm, act))
    ents = (_lib.ProfileEntry * 16)()
    n = ctypes.c_int(0)
    _lib.check(lib.uegan_profile_end(ents, 16, ctypes.byref(n)))
    names = [ents[i].name.decode() for i in range(n.value)]
    assert names == ["conv_toep_kernel<bf16,K=%d,MODE=0>" % k], names
    assert y2.shape[-1] == 8 and float(y2[..., Co:].abs().sum()) == 0.0       # padding channels stay exactly zero
    assert rel(nchw(y2[..., :Co]), y) < BF16_TOL


# bf16 weight gradients: the transpose-read kernel (wgrad_tr.h).  (B, C1, C2, H, W, Cout, k, stride, pad_mode)
WGRAD_TR_CASES = [
    (1, 64, 0, 12, 100, 32, 3, 1, 1),     # TW=32: interior + border tiles, 4 tiles wide, 64-ch rows x 32 dz channels (dec4-like)
    (2, 32, 0, 20, 40, 32, 3, 1, 1),      # 32-channel rows (64-byte LDS rows), k-step waves (WK=2, WS=2)
    (1, 32, 32, 9, 36, 32, 3, 1, 1),      # two sources inside one 64-channel chunk
    (1, 8, 0, 24, 70, 32, 7, 1, 1),       # 8-channel rows: a fragment = two taps (enc1-like)
    (1, 128, 0, 10, 33, 128, 3, 1, 1),    # two x chunks x two dz blocks
    (1, 64, 0, 10, 34, 8, 7, 1, 1),       # 7x7, 64 channels: one kernel row per slot range; narrow head (Cout 3 -> 8)
    (2, 32, 0, 40, 40, 1, 7, 1, 1),       # D head shape: 32 -> 1, 7x7
    (1, 256, 0, 6, 6, 1, 5, 1, 1),        # 5x5 head on a 6x6 map, 4 chunks
    (1, 64, 0, 8, 40, 64, 1, 1, 1),       # 1x1 (GAM fuse / up convs)
    (1, 32, 0, 16, 33, 32, 1, 1, 1),      # 1x1, 32 channels: all four waves split the k-steps
    (1, 64, 0, 12, 40, 64, 3, 1, 0),      # zero padding
    (1, 32, 0, 40, 72, 64, 3, 2, 1),      # stride 2, interior tiles (enc2-like)
    (2, 8, 0, 33, 70, 32, 7, 2, 1),       # stride 2, 7x7, 8-channel rows (d1-like), odd sizes
    (1, 64, 0, 20, 36, 128, 7, 2, 1),     # stride 2, 7x7, 64 channels (d3-like)
    (1, 128, 0, 9, 9, 256, 5, 2, 1),      # stride 2, 5x5 small map (d4/d5-like)
    (1, 16, 0, 8, 8, 16, 3, 1, 1),        # 16-channel rows
    # heads on maps >= 32 wide: im2col of dz over tx (MFMA rows = (tx, n) pairs)
    (1, 32, 0, 20, 70, 3, 7, 1, 1),       # G head 32 -> 3, 7x7: 21 rows (two row blocks), ragged width, 3 tiles wide
    (2, 64, 0, 33, 64, 1, 7, 1, 1),       # D head 64 -> 1, 7x7
    (1, 128, 0, 12, 40, 1, 5, 1, 1),      # 5x5, two channel chunks
    (1, 32, 0, 9, 33, 2, 3, 1, 1),        # 3x3, 2 outputs
    (1, 32, 0, 16, 32, 1, 7, 1, 0),       # zero padding
    (1, 32, 0, 12, 40, 4, 5, 1, 1),       # 4 outputs (20 (tx, n) rows)
    (2, 256, 0, 34, 30, 128, 5, 2, 1),    # stride 2, 5x5: every slot range is one kernel row -> only that row's input parity is staged; 17 x 15 outputs
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", WGRAD_TR_CASES, ids=lambda c: "x".join(map(str, c)))
def test_wgrad_transpose_read_kernel(backend, case):
    import ctypes
    dev = use_backend(backend)
    lib = _lib.load()
    B, C1, C2, H, W, Co, k, s, pm = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = bf16_round(torch.randn(B, C1 + C2, H, W, generator=g))
    w = bf16_round(torch.randn(Co, C1 + C2, k, k, generator=g) * 0.1).requires_grad_(True)
    b = torch.zeros(Co, requires_grad=True)
    y = ref_conv(x, w, b, s, pm, ops.ACT_NONE)
    r = bf16_round(torch.randn(y.shape, generator=g))
    (y * r).sum().backward()
    xn = nhwc(x).to(torch.bfloat16).to(dev)
    x1 = xn[..., :C1].contiguous()
    x2 = xn[..., C1:].contiguous() if C2 else None
    w2 = w.detach().clone().to(dev).requires_grad_(True)
    b2 = b.detach().clone().to(dev).requires_grad_(True)
    _lib.check(lib.uegan_profile_begin(64))
    y2 = ops.conv2d(x1, x2, w2, b2, ops.ConvCfg(s, pm, ops.ACT_NONE))
    cp = ops.cpad(Co, torch.bfloat16)
    y2.backward(F.pad(nhwc(r), (0, cp - Co)).to(torch.bfloat16).to(dev).contiguous())
    ents = (_lib.ProfileEntry * 16)()
    n = ctypes.c_int(0)
    _lib.check(lib.uegan_profile_end(ents, 16, ctypes.byref(n)))
    names = [ents[i].name.decode() for i in range(n.value)]
    assert any(nm.startswith("wgrad_tr_kernel") for nm in names), names
    assert rel(w2.grad, w.grad) < BF16_TOL
    assert rel(b2.grad, b.grad) < BF16_TOL


@pytest.mark.parametrize("backend", BACKENDS)
def test_conv_direct_kernels_agree(backend):
    """the scalar cross-check kernels (UEGAN_IMPL_DIRECT) give the same answers as the MFMA path"""
    dev = use_backend(backend)
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 7, 7, 8, generator=g).to(dev)
    w = (torch.randn(12, 8, 5, 5, generator=g) * 0.1).to(dev)
    b = torch.randn(12, generator=g).to(dev)
    outs = []
    for impl in (1, 3, 2):          # MFMA + direct-to-LDS staging, MFMA + register staging, scalar direct
        lib.uegan_set_conv_impl(impl)
        try:
            x1 = x.clone().requires_grad_(True)
            w1 = w.clone().requires_grad_(True)
            b1 = b.clone().requires_grad_(True)
            y = ops.conv2d(x1, None, w1, b1, ops.ConvCfg(2, ops.PAD_REFLECT, ops.ACT_LRELU))
            y.backward(torch.ones_like(y))
            outs.append((y.detach(), x1.grad, w1.grad, b1.grad))
        finally:
            lib.uegan_set_conv_impl(0)
    for other in outs[1:]:
        for a, c in zip(outs[0], other):
            assert rel(a, c) < F32_TOL


@pytest.mark.parametrize("backend", BACKENDS)
def test_conv_rejects_oversized_reflection_pad(backend):
    dev = use_backend(backend)
    x = torch.zeros(1, 2, 2, 8, device=dev)
    w = torch.zeros(1, 8, 5, 5, device=dev)
    with pytest.raises(RuntimeError, match="Padding size should be less"):       # same failure the reference D hits at 64x64
        ops.conv2d(x, None, w, None, ops.ConvCfg(1, ops.PAD_REFLECT, ops.ACT_TANH))


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_resample_pool_norm_elementwise(backend, dtype):
    dev = use_backend(backend)
    tol = F32_TOL if dtype == torch.float32 else BF16_TOL
    g = torch.Generator().manual_seed(3)

    def rnd(*s):
        t = torch.randn(*s, generator=g)
        return bf16_round(t) if dtype == torch.bfloat16 else t

    def chk(fn_ref, fn_ours, *shapes, name=""):
        xs = [rnd(*s).requires_grad_(True) for s in shapes]
        y = fn_ref(*xs)
        r = rnd(*y.shape)
        (y * r).sum().backward()
        xn = [nhwc(x).to(dtype).to(dev).requires_grad_(True) for x in xs]
        y2 = fn_ours(*xn)
        y2.backward(nhwc(r).to(dtype).to(dev))
        assert rel(nchw(y2), y) < tol, name
        for a, c in zip(xn, xs):
            assert rel(nchw(a.grad), c.grad) < tol, name

    chk(lambda x: F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True), ops.upsample2x, (2, 5, 6, 7), name="upsample")
    chk(lambda x: F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True), ops.upsample2x, (1, 16, 1, 3), name="upsample-1row")
    # (40 input rows x 2 column blocks)
    chk(lambda x: F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True), ops.upsample2x, (2, 64, 20, 40), name="upsample-rows")
    chk(lambda x: F.max_pool2d(x, 2, 2), ops.maxpool2x2, (2, 5, 6, 8), name="maxpool")
    chk(lambda a, b: a * b, ops.mul, (2, 3, 4, 5), (2, 3, 4, 5), name="mul")
    if dtype == torch.float32:      # statistics are fp32 either way; bf16 storage of y is covered by the model tests
        for shp in ((2, 5, 6, 8), (1, 70, 40, 40), (2, 64, 3, 3)):      # C not a power of two; HW > one split
            x = (torch.randn(*shp, generator=g) * 2 + 3).requires_grad_(True)
            y = F.instance_norm(x, eps=1e-5)
            r = torch.randn(y.shape, generator=g)
            (y * r).sum().backward()
            xn = nhwc(x).to(dev).requires_grad_(True)
            y2 = ops.instnorm(xn)
            y2.backward(nhwc(r).to(dev))
            assert rel(nchw(y2), y) < tol and rel(nchw(xn.grad), x.grad) < 5 * tol, shp


@pytest.mark.parametrize("backend", BACKENDS)
def test_boundary_layout_ops(backend):
    dev = use_backend(backend)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 3, 4, 6, generator=g).requires_grad_(True)
    a, b = [0.5, 2.0, 3.0], [0.1, -0.2, 0.3]
    y = x * torch.tensor(a).view(1, 3, 1, 1) + torch.tensor(b).view(1, 3, 1, 1)
    r = torch.randn(y.shape, generator=g)
    (y * r).sum().backward()
    xn = x.detach().clone().to(dev).requires_grad_(True)
    y2 = ops.to_nhwc(xn, torch.float32, a, b)
    assert tuple(y2.shape) == (2, 4, 6, 4) and float(y2[..., 3:].abs().sum()) == 0     # 3 -> 4 channels (fp32 chunk)
    y2.backward(F.pad(nhwc(r), (0, 1)).to(dev))
    assert rel(nchw(y2[..., :3]), y) < 1e-6 and rel(xn.grad, x.grad) < 1e-6
    xh = F.pad(nhwc(x), (0, 1)).contiguous().to(dev).requires_grad_(True)      # 3 real channels + 1 padding channel
    y3 = ops.to_nchw(xh, 3)
    y3.backward(r.to(dev))
    assert rel(y3, x) == 0 and rel(nchw(xh.grad[..., :3]), r) == 0 and float(xh.grad[..., 3:].abs().sum()) == 0
    # residual + clamp (models.py:72) incl. values exactly at the clamp bounds
    rs = torch.randn(2, 3, 8, 8, generator=g)
    xx = torch.randn(2, 3, 8, 8, generator=g)
    rs[0, 0, 0, 0], xx[0, 0, 0, 0] = 0.5, 0.5          # sum == 1.0: gradient passes (inclusive bound)
    rs.requires_grad_(True), xx.requires_grad_(True)
    y = torch.clamp(rs + xx, -1, 1)
    (y * 1.5).sum().backward()
    rn = F.pad(nhwc(rs), (0, 1)).contiguous().to(dev).requires_grad_(True)
    xn = xx.detach().clone().to(dev).requires_grad_(True)
    y2 = ops.residual_clamp(rn, xn)
    (y2 * 1.5).sum().backward()
    assert rel(y2, y) == 0 and rel(nchw(rn.grad[..., :3]), rs.grad) == 0 and rel(xn.grad, xx.grad) == 0
    assert float(rn.grad[..., 3:].abs().sum()) == 0


@pytest.mark.parametrize("backend", BACKENDS)
def test_losses_against_oracle(backend):
    dev = use_backend(backend)
    g = torch.Generator().manual_seed(6)
    sizes = (12, 6, 3)
    reals = [torch.tanh(torch.randn(2, 1, s, s, generator=g)).requires_grad_(True) for s in sizes]
    fakes = [torch.tanh(torch.randn(2, 1, s, s, generator=g) - 0.3).requires_grad_(True) for s in sizes]
    for ford in (True, False):
        l = O.rahinge_loss(reals, fakes, ford)
        gs = torch.autograd.grad((l * 0.7).sum(), reals + fakes)
        r2 = [t.detach().clone().to(dev).requires_grad_(True) for t in reals]
        f2 = [t.detach().clone().to(dev).requires_grad_(True) for t in fakes]
        l2 = ops.rahinge(r2, f2, ford)
        assert tuple(l2.shape) == (1,)
        (l2 * 0.7).sum().backward()
        assert abs(float(l2) - float(l)) < 1e-6 * abs(float(l))
        for t, gr in zip(r2 + f2, gs):
            assert rel(t.grad, gr) < F32_TOL
    a = (torch.rand(2, 3, 16, 24, generator=g) * 2 - 1).requires_grad_(True)
    b = torch.rand(2, 3, 16, 24, generator=g) * 2 - 1
    l = O.multiscale_l1(a, b)
    (l * 0.1).backward()
    a2 = a.detach().clone().to(dev).requires_grad_(True)
    l2 = ops.multiscale_l1(a2, b.to(dev))
    assert l2.dim() == 0
    (l2 * 0.1).backward()
    assert abs(float(l2) - float(l)) < 1e-6 * float(l) and rel(a2.grad, a.grad) < F32_TOL
    # sizes AvgPool2d(2, 2) floors (6 x 10 -> 3 x 5 -> 1 x 2), against the oracle's F.avg_pool2d (reference-generated cases: test_variants.py)
    a = (torch.rand(1, 3, 6, 10, generator=g) * 2 - 1).requires_grad_(True)
    b = torch.rand(1, 3, 6, 10, generator=g) * 2 - 1
    l = O.multiscale_l1(a, b)
    l.backward()
    a2 = a.detach().clone().to(dev).requires_grad_(True)
    l2 = ops.multiscale_l1(a2, b.to(dev))
    l2.backward()
    assert abs(float(l2) - float(l)) < 1e-6 * float(l) and rel(a2.grad, a.grad) < F32_TOL
    with pytest.raises(RuntimeError, match="too small"):                  # a map that would pool to nothing
        ops.multiscale_l1(torch.zeros(1, 3, 3, 8, device=dev), torch.zeros(1, 3, 3, 8, device=dev))
    # fidelity-loss taps (InstanceNorm + MSE, weights as losses.py:17)
    xs = [torch.randn(2, c, h, h, generator=g).abs().requires_grad_(True) for c, h in ((8, 16), (16, 8), (70, 40))]
    ys = [(x.detach() + 0.3 * torch.randn(x.shape, generator=g)).abs() for x in xs]
    ws = [1 / 64, 1 / 32, 1.0]
    l = sum(w * F.mse_loss(F.instance_norm(x, eps=1e-5), F.instance_norm(y, eps=1e-5)) for w, x, y in zip(ws, xs, ys))
    gs = torch.autograd.grad(l * 1.5, xs)
    xn = [nhwc(x).to(dev).requires_grad_(True) for x in xs]
    l2 = ops.perceptual_taps_loss(xn, [nhwc(y).to(dev) for y in ys], ws)
    (l2 * 1.5).backward()
    assert abs(float(l2) - float(l)) < 1e-5 * float(l)
    for t, gr in zip(xn, gs):
        assert rel(nchw(t.grad), gr) < 5 * F32_TOL


@pytest.mark.parametrize("backend", BACKENDS)
def test_spectral_norm_and_adam(backend):
    dev = use_backend(backend)
    g = torch.Generator().manual_seed(8)
    w = torch.randn(12, 3, 5, 5, generator=g)
    u = F.normalize(torch.randn(12, generator=g), dim=0)
    v = F.normalize(torch.randn(75, generator=g), dim=0)
    P = {"x.weight_orig": w.clone().requires_grad_(True), "x.weight_u": u.clone(), "x.weight_v": v.clone()}
    wn = O.spectral_norm_weight(P, "x", True)
    r = torch.randn(wn.shape, generator=g)
    (wn * r).sum().backward()
    wd, u2, v2 = w.to(dev), u.clone().to(dev), v.clone().to(dev)
    sn = ops.specnorm_sigma(wd, u2, v2, True)
    assert rel(u2, P["x.weight_u"]) < F32_TOL and rel(v2, P["x.weight_v"]) < F32_TOL
    sig = float(P["x.weight_u"] @ (w.view(12, -1) @ P["x.weight_v"]))
    assert abs(float(sn.sigma[0]) - sig) < 1e-6 * sig and abs(float(sn.sigma[1]) * sig - 1) < 1e-6
    gsc = (r / sig).contiguous().to(dev)
    dw = torch.empty_like(gsc)
    # (workspace contract: uegan_specnorm_grad_workspace_floats() floats, one partial of <G, W> per block; no zero-initialisation needed)
    tmp = torch.full((_lib.load().uegan_specnorm_grad_workspace_floats(),), float("nan"), device=dev)
    _lib.check(_lib.load().uegan_specnorm_grad(gsc.data_ptr(), wd.data_ptr(), u2.data_ptr(), v2.data_ptr(), sn.sigma.data_ptr(), dw.data_ptr(),
                                               12, 75, tmp.data_ptr(), None))
    assert rel(dw, P["x.weight_orig"].grad) < F32_TOL
    # a matrix large enough for several partials (rows * cols > 1024: more than one block of the dot kernel)
    wb = torch.randn(24, 8, 5, 5, generator=g)
    ub, vb = F.normalize(torch.randn(24, generator=g), dim=0), F.normalize(torch.randn(200, generator=g), dim=0)
    Pb = {"x.weight_orig": wb.clone().requires_grad_(True), "x.weight_u": ub.clone(), "x.weight_v": vb.clone()}
    rb = torch.randn(wb.shape, generator=g)
    (O.spectral_norm_weight(Pb, "x", True) * rb).sum().backward()
    wbd, ub2, vb2 = wb.to(dev), ub.clone().to(dev), vb.clone().to(dev)
    snb = ops.specnorm_sigma(wbd, ub2, vb2, True)
    gb = (rb / float(snb.sigma[0])).contiguous().to(dev)
    dwb = torch.empty_like(gb)
    tmp.fill_(float("nan"))
    _lib.check(_lib.load().uegan_specnorm_grad(gb.data_ptr(), wbd.data_ptr(), ub2.data_ptr(), vb2.data_ptr(), snb.sigma.data_ptr(), dwb.data_ptr(),
                                               24, 200, tmp.data_ptr(), None))
    assert rel(dwb, Pb["x.weight_orig"].grad) < F32_TOL
    sn2 = ops.specnorm_sigma(wd, u2, v2, False)                 # eval mode: no iteration, same sigma
    assert abs(float(sn2.sigma[0]) - sig) < 1e-6 * sig and rel(u2, P["x.weight_u"]) < F32_TOL
    # fused Adam == torch.optim.Adam(weight_decay) with the 1/world grad scale folded in
    ps = [torch.randn(5, 3, generator=g), torch.randn(3000, generator=g), torch.randn(7, generator=g)]
    params = [torch.nn.Parameter(p.clone().to(dev)) for p in ps]
    refp = [torch.nn.Parameter(p.clone()) for p in ps]
    opt = ops.FusedAdamL2(params, 1e-2, (0.5, 0.999), 1e-8, 1e-4)
    topt = torch.optim.Adam(refp, lr=1e-2, betas=(0.5, 0.999), weight_decay=1e-4)
    for _ in range(3):
        opt.zero_grad()
        topt.zero_grad()
        for p, q in zip(params, refp):
            gr = torch.randn(q.shape, generator=g)
            p.grad.add_((2 * gr).to(dev))
            q.grad = gr.clone()
        opt.step(grad_scale=0.5)
        topt.step()
    for p, q in zip(params, refp):
        assert rel(p, q) < 1e-6


@pytest.mark.gpu
def test_mfma_fragment_layouts_on_hardware():
    """the fragment layouts the kernels (and the emulator) assume, checked on the real matrix cores: A = I, asymmetric B"""
    dev = use_backend("gpu")
    scratch = torch.zeros(4096, device=dev)
    _lib.check(_lib.load().uegan_selftest_mfma(scratch.data_ptr(), None))
    s = scratch.cpu()
    for lane in range(64):
        for r in range(4):
            i, j = 4 * (lane >> 4) + r, lane & 15
            assert s[lane * 4 + r] == (100.0 * i + j if i < 4 else 0.0)
            assert s[256 + lane * 4 + r] == (i * 8 + j) * 0.5
        for r in range(16):       # 32x32x16 bf16 (conv_wide.hip): D[i][j] = i + 1 / j + 1 for i < 16
            i, j = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), lane & 31
            assert s[512 + lane * 16 + r] == (i + 1.0 if i < 16 else 0.0)
            assert s[1536 + lane * 16 + r] == (j + 1.0 if i < 16 else 0.0)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_deferred_activation_gradient_matches_plain_autograd(backend, dtype):
    """VGG19_relu(deferred_act_grad=True) (no ReLU-backward passes: relu'(y) applied by y's consumers -- the next conv's
    dgrad epilogue (uegan_conv2d_dgrad_act), the max-pool backward and the fidelity-loss gradient) against the same network
    with one act_bwd pass per layer.  The restructuring is exact: the masks are 0/1 factors."""
    from uegan_amd import losses
    dev = use_backend(backend)
    sd = losses.seeded_vgg19_weights(width_div=8)
    nets = [losses.VGG19_relu(sd, 8, deferred_act_grad=flag).to(dev) for flag in (False, True)]
    g = torch.Generator().manual_seed(11)
    x = torch.rand(2, 3, 32, 48, generator=g) * 2 - 1
    y = torch.rand(2, 3, 32, 48, generator=g) * 2 - 1
    ws = [1 / 64, 1 / 64, 1 / 32, 1 / 32, 1.0]
    grads, vals = [], []
    for net, act in zip(nets, (ops.ACT_NONE, ops.ACT_RELU)):
        xi = x.clone().to(dev).requires_grad_(True)
        tx = net(ops.to_nhwc(xi, dtype=dtype))
        with torch.no_grad():
            ty = net(ops.to_nhwc(y.to(dev), dtype=dtype))
        l = ops.perceptual_taps_loss(tx, ty, ws, in_act=act)
        l.backward()
        grads.append(xi.grad.clone())
        vals.append(float(l))
    assert vals[0] == vals[1]
    assert float(grads[0].abs().max()) > 0
    # fp32: identical up to the order in which autograd sums a tap's two gradients; bf16: the mask is applied before
    # instead of after a bf16 rounding of the same value -- identical products, so the same bound holds
    assert rel(grads[1], grads[0]) < (F32_TOL if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_optimizer_step_repacks_all_weights_in_one_launch(backend, dtype):
    """FusedAdamL2.step re-packs every packed copy of the weights it updated with ONE uegan_pack_weights_multi launch: the copies
    must equal what a fresh uegan_pack_weights_slice of the updated master gives (incl. a column-slice weight and padded channels)."""
    dev = use_backend(backend)
    torch.manual_seed(5)
    ws = [torch.nn.Parameter(torch.randn(16, 8, 3, 3, device=dev)), torch.nn.Parameter(torch.randn(3, 32, 7, 7, device=dev)),
          torch.nn.Parameter(torch.randn(8, 24, 1, 1, device=dev))]
    cfgs = [ops.ConvCfg(1, ops.PAD_REFLECT, 1), ops.ConvCfg(1, ops.PAD_REFLECT, 3), ops.ConvCfg(1, ops.PAD_REFLECT, 0, cin_used=16)]
    shapes = [(8, 16), (32, 8), (16, 8)]       # (cin_pad, cout_pad)
    opt = ops.FusedAdamL2(ws, 1e-2)
    for w, c, (ci, co) in zip(ws, cfgs, shapes):
        c.packed.get(w, dtype, ci, co, None, c.cin_used)
    old = ops.get_compute_dtype()
    ops.set_compute_dtype(dtype)
    try:
        opt.flat_grad.copy_(torch.randn_like(opt.flat_grad))
        opt.step()
        assert opt._packs.n == 3
        for w, c, (ci, co) in zip(ws, cfgs, shapes):
            key = c.packed.key
            a, b = c.packed.get(w, dtype, ci, co, None, c.cin_used)
            assert c.packed.key == key, "the copies made by the optimizer step must be cache hits"
            fresh = ops.PackedWeight()
            fa, fb = fresh.get(w, dtype, ci, co, None, c.cin_used)
            assert torch.equal(a, fa) and torch.equal(b, fb)
    finally:
        ops.set_compute_dtype(old)


# conv + ReLU + 2x2 max-pool with the pooled tensor written by the convolution's epilogue (uegan_conv2d_fwd_pool).  (B, C, H, W, Cout)
POOL_CASES = [
    (1, 64, 20, 36, 64),       # 64-channel blocks on 256-pixel tiles (VGG conv1_2's kernel), ragged tiles
    (2, 64, 32, 32, 128),      # 128 channels x 32-row tiles (conv2_2's kernel; needs UEGAN_TUNE_SMALL_GRID = 0 on these map sizes)
    (1, 128, 16, 32, 72),      # N = 72 on 128-channel blocks of 16 rows: no fused variant -> the pooling kernel runs behind the conv
    (1, 8, 12, 12, 8),         # generic kernel: fallback
]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", POOL_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_fwd_with_fused_maxpool(backend, dtype, case, monkeypatch):
    import ctypes
    set_tuning("SMALL_GRID", 0)
    # (the unfused launch this compares with bit for bit must sum its K loop in the same order: no split-K on these emulator-sized grids)
    monkeypatch.setattr(ops, "split_k", [False])
    dev = use_backend(backend)
    lib = _lib.load()
    B, Cc, H, W, Co = case
    g = torch.Generator().manual_seed(3 + Co)
    x = torch.randn(B, H, W, Cc, generator=g).to(dtype).to(dev)
    w = (torch.randn(Co, Cc, 3, 3, generator=g) * 0.05).to(dev)
    b = torch.randn(Co, generator=g).to(dev)
    cfg = ops.ConvCfg(1, ops.PAD_ZERO, ops.ACT_RELU)
    _lib.check(lib.uegan_profile_begin(16))
    y, d, _, yp = ops.raw_conv_fwd(x, None, w, b, cfg, pool=True)
    ents = (_lib.ProfileEntry * 16)()
    n = ctypes.c_int(0)
    _lib.check(lib.uegan_profile_end(ents, 16, ctypes.byref(n)))
    y0, _, _ = ops.raw_conv_fwd(x, None, w, b, cfg)
    assert torch.equal(y, y0)
    ref = ops.maxpool2x2(y0)
    assert yp.shape == ref.shape and torch.equal(yp, ref), float((yp.float() - ref.float()).abs().max())
    # with the window positions of the maxima (uegan_conv2d_fwd_pool_idx; these kernels' epilogues do not produce them: the pooling kernel does)
    _, _, _, yp_i, idx = ops.raw_conv_fwd(x, None, w, b, cfg, pool=True, n_full=0, n_idx=B)
    assert torch.equal(yp_i, yp)
    win = y0.reshape(B, H // 2, 2, W // 2, 2, y0.shape[-1]).permute(0, 1, 3, 5, 2, 4).reshape(B, H // 2, W // 2, y0.shape[-1], 4).float()
    first_max = (win == win.max(dim=-1, keepdim=True).values).float().argmax(dim=-1)      # (argmax of a 0/1 tensor: the FIRST maximum)
    assert torch.equal(idx.long(), first_max)
    # ... and against plain PyTorch
    yt = F.max_pool2d(F.relu(F.conv2d(nchw(x.float().cpu()), w.cpu(), b.cpu(), padding=1)), 2)
    assert rel(nchw(yp[..., :Co]), yt) < (BF16_TOL if dtype == torch.bfloat16 else F32_TOL)


@pytest.mark.parametrize("backend", BACKENDS)
def test_tuning_api(backend):
    """uegan_set_tuning: the only way to move a launch threshold (the library reads no environment variable); returns the previous value, refuses
    unknown knobs"""
    import ctypes
    use_backend(backend)
    lib = _lib.load()
    prev = ctypes.c_int(-123)
    _lib.check(lib.uegan_set_tuning(0, 7, ctypes.byref(prev)))
    assert prev.value == 256                       # UEGAN_TUNE_SMALL_GRID's default
    _lib.check(lib.uegan_set_tuning(0, 256, ctypes.byref(prev)))
    assert prev.value == 7
    assert lib.uegan_set_tuning(99, 1, None) != 0
    assert b"unknown tuning knob" in lib.uegan_last_error()


# (B, C1, C2, H, W, Cout, k, stride): forwards on grids that leave most of the chip empty -- the emulator-sized ones and, on the GPU, the four layers of a
# single 512 x 512 image the split exists for (models.py:17-18 enc4 / enc5, :58-62 dec1 / dec2)
SPLITK_CASES = [(1, 128, 64, 20, 20, 136, 3, 1), (1, 256, 0, 18, 34, 72, 3, 2), (2, 320, 0, 16, 16, 128, 3, 1), (1, 128, 0, 34, 70, 72, 5, 2)]
SPLITK_GPU_CASES = [(1, 256, 256, 64, 64, 256, 3, 1), (1, 128, 128, 128, 128, 128, 3, 1), (1, 256, 0, 64, 64, 512, 3, 2), (1, 128, 0, 128, 128, 256, 3, 2)]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_conv_split_k_forward(backend, dtype):
    """uegan_conv2d_fwd_splitk against uegan_conv2d_fwd on the same operands: the workspace IS used (its NaN fill is overwritten by the parts), the result
    differs from the unsplit launch only by the fp32 summation order (a last-place rounding of a few elements), and a second call is bit-identical."""
    import ctypes as C
    from uegan_amd import _lib as L
    dev = use_backend(backend)
    ops.set_compute_dtype(dtype)
    lib = ops.lib()
    for case in SPLITK_CASES + (SPLITK_GPU_CASES if backend == "gpu" else []):
        B, C1, C2, H, W, Co, k, stride = case
        g = torch.Generator().manual_seed(sum(case))
        x1 = nhwc(torch.randn(B, C1, H, W, generator=g)).to(dtype).to(dev).contiguous()
        x2 = nhwc(torch.randn(B, C2, H, W, generator=g)).to(dtype).to(dev).contiguous() if C2 else None
        w = (torch.randn(Co, C1 + C2, k, k, generator=g) / (k * (C1 + C2) ** 0.5)).to(dev)
        bias = torch.randn(Co, generator=g).to(dev)
        cfg = ops.ConvCfg(stride, ops.PAD_REFLECT, ops.ACT_LRELU)
        d = ops._desc(x1, x2, w, cfg)
        ohwi, _ = cfg.packed.get(w, dtype, d.C1 + d.C2, d.Cout)
        wsb = lib.uegan_conv2d_fwd_splitk_workspace_bytes(C.byref(d))
        assert wsb > 0, case
        elems = d.B * d.Ho * d.Wo * d.Cout
        assert wsb % (4 * elems) == 0 and wsb // (4 * elems) >= 2, (case, wsb)
        y0 = torch.empty((d.B, d.Ho, d.Wo, d.Cout), dtype=dtype, device=dev)
        L.check(lib.uegan_conv2d_fwd(C.byref(d), x1.data_ptr(), x2.data_ptr() if C2 else None, ohwi.data_ptr(), bias.data_ptr(), None, y0.data_ptr(), ops._stream()))
        outs = []
        for rep in range(2):
            ws = torch.full((wsb // 4,), float("nan"), dtype=torch.float32, device=dev)
            y = torch.empty_like(y0)
            L.check(lib.uegan_conv2d_fwd_splitk(C.byref(d), x1.data_ptr(), x2.data_ptr() if C2 else None, ohwi.data_ptr(), bias.data_ptr(), None, y.data_ptr(),
                                                ws.data_ptr(), wsb, ops._stream()))
            assert not bool(torch.isnan(ws[:2 * elems]).any().cpu()), case      # at least two parts were written, every element of each
            outs.append(y.float().cpu())
        assert torch.equal(outs[0], outs[1]), case
        a, b = outs[0], y0.float().cpu()
        ulp = 2.0 ** (-10 if dtype == torch.float16 else -7)
        assert float((a - b).abs().max()) <= ulp * float(b.abs().max()), (case, float((a - b).abs().max()))
        assert float((a != b).float().mean()) < 0.05, (case, float((a != b).float().mean()))
        # a short workspace: the plain launch, bit-identical to uegan_conv2d_fwd
        y = torch.empty_like(y0)
        L.check(lib.uegan_conv2d_fwd_splitk(C.byref(d), x1.data_ptr(), x2.data_ptr() if C2 else None, ohwi.data_ptr(), bias.data_ptr(), None, y.data_ptr(),
                                            ws.data_ptr(), 4 * elems, ops._stream()))
        assert torch.equal(y.float().cpu(), b), case


@pytest.mark.parametrize("backend", BACKENDS)
def test_conv_stream_kernel_scale_per_image_group(backend):
    """ConvArgs::scale / scale_group on the persistent streaming kernel (a block walks consecutive tiles across image boundaries and fetches 1 / sigma
    when the group of its tile's image changes): three groups of two images with different scales, forward and data gradient (stride 1, and the
    stride-2 parity-class kernel) against the plain result times the group's scale"""
    import ctypes as C
    dev = use_backend(backend)
    ops.set_compute_dtype(torch.bfloat16)
    lib = ops.lib()
    B, Cc, H, W, Co = 6, 32, 32, 64, 32
    g = torch.Generator().manual_seed(12)
    x = torch.randn(B, H, W, Cc, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(Co, Cc, 3, 3, generator=g) / (3 * Cc ** 0.5)).to(dev)
    bias = torch.randn(Co, generator=g).to(dev)
    scale = torch.tensor([0.5, 2.0, -1.25], device=dev)
    cfg = ops.ConvCfg(1, ops.PAD_REFLECT, ops.ACT_NONE)
    d = ops._desc(x, None, w, cfg)
    ohwi, ihwo = cfg.packed.get(w, x.dtype, d.C1, d.Cout)
    d0 = ops._desc(x, None, w, cfg)
    y0 = torch.empty((B, H, W, Co), dtype=x.dtype, device=dev)
    _lib.check(lib.uegan_conv2d_fwd(C.byref(d0), x.data_ptr(), None, ohwi.data_ptr(), None, None, y0.data_ptr(), ops._stream()))
    d.scale_group = 2
    _lib.check(lib.uegan_profile_begin(16))
    y = torch.empty_like(y0)
    _lib.check(lib.uegan_conv2d_fwd(C.byref(d), x.data_ptr(), None, ohwi.data_ptr(), bias.data_ptr(), scale.data_ptr(), y.data_ptr(), ops._stream()))
    ents = (_lib.ProfileEntry * 16)()
    n = C.c_int(0)
    _lib.check(lib.uegan_profile_end(ents, 16, C.byref(n)))
    assert any(ents[i].name.decode().startswith("conv_stream_kernel") for i in range(n.value)), [ents[i].name.decode() for i in range(n.value)]
    sc = scale.cpu().repeat_interleave(2).view(B, 1, 1, 1)
    ref = y0.float().cpu() * sc + bias.cpu().view(1, 1, 1, Co)
    # (y0 is rounded to bf16 before the scale is applied here, the kernel scales the fp32 sum: one more half-ulp)
    assert float((y.float().cpu() - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())
    # data gradients: the stride-1 layer's and a stride-2 layer's (four parity classes per tile: the group's scale in every class's epilogue)
    for stride, wshape in ((1, (Co, Cc, 3, 3)), (2, (64, Cc, 3, 3))):
        w2 = (torch.randn(*wshape, generator=g) / (3 * Cc ** 0.5)).to(dev)
        cfg2 = ops.ConvCfg(stride, ops.PAD_REFLECT, ops.ACT_NONE)
        dd = ops._desc(x, None, w2, cfg2)
        _, ihwo2 = cfg2.packed.get(w2, x.dtype, dd.C1, dd.Cout)
        dz = torch.randn(B, dd.Ho, dd.Wo, dd.Cout, generator=g).to(torch.bfloat16).to(dev)
        wsb = lib.uegan_conv2d_dgrad_workspace_bytes(C.byref(dd))
        ws = torch.empty((wsb + 3) // 4 + 1, dtype=torch.float32, device=dev)
        g0 = torch.empty_like(x)
        _lib.check(lib.uegan_conv2d_dgrad_ws(C.byref(dd), dz.data_ptr(), ihwo2.data_ptr(), None, g0.data_ptr(), None, ws.data_ptr(), wsb, ops._stream()))
        dd.scale_group = 2
        g1 = torch.empty_like(x)
        _lib.check(lib.uegan_conv2d_dgrad_ws(C.byref(dd), dz.data_ptr(), ihwo2.data_ptr(), scale.data_ptr(), g1.data_ptr(), None, ws.data_ptr(), wsb, ops._stream()))
        refg = g0.float().cpu() * sc
        assert float((g1.float().cpu() - refg).abs().max()) <= 2.0 ** -7 * float(refg.abs().max()), stride
