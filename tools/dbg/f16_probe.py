"""bf16 vs fp16 storage (libuegan_hip_f16.so) against the fp32 path: one full 16x3x512^2 step (losses, images, gradient buckets) for several
loss scales, inference PSNR at 1x3x512^2, and the step time of each mode."""
import os, sys, random, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from oracle import uegan_oracle as O
from uegan_amd import losses, models, ops, tester, trainer
dev = torch.device("cuda:0")
def images(B, S, seed):
    g = torch.Generator().manual_seed(seed)
    lo = torch.rand(B, 3, S // 32, S // 32, generator=g)
    x = torch.nn.functional.interpolate(lo, size=(S, S), mode="bicubic", align_corners=False) + 0.03 * torch.randn(B, 3, S, S, generator=g)
    return (x.clamp(0, 1) * 2 - 1).contiguous()
PG = O.init_params(O.generator_param_shapes(32), 41, "default")
PD = O.init_params(O.discriminator_param_shapes(32), 42, "default")
B, S = 16, 512
raw, exp = images(B, S, 1990).to(dev), images(B, S, 1991).to(dev)
def run(dt, scale=None, steps=1):
    ops.set_compute_dtype(dt)
    G = models.Generator(32, "none", "LeakyReLU", False); D = models.Discriminator(32, "none", "LeakyReLU", True, "rahinge")
    G.load_state_dict(PG); D.load_state_dict(PD)
    T = trainer.Trainer(G.to(dev), D.to(dev), losses.PerceptualLoss(vgg_weights="seeded").to(dev), pool_size=50, rng=random.Random(1990), loss_scale=scale)
    T.train_step(raw, exp)
    torch.cuda.synchronize()
    out = dict(losses=T.loss_items(), fake=T.fake_exp.float().cpu(), gG=T.g_optimizer.flat_grad.clone().cpu() / T.loss_scale,
               gD=T.d_optimizer.flat_grad.clone().cpu() / T.loss_scale)
    t0 = time.perf_counter()
    for _ in range(6): T.train_step(raw, exp)
    torch.cuda.synchronize()
    out["ms"] = (time.perf_counter() - t0) / 6 * 1e3
    out["losses_after7"] = T.loss_items()
    return out
ref = run(torch.float32)
print("fp32: %.1f ms/step" % ref["ms"], ref["losses"])
def cmp(tag, r):
    a = ref
    dev_l = {k: abs(r["losses"][k] - a["losses"][k]) / abs(a["losses"][k]) for k in a["losses"]}
    cosG = float((r["gG"] * a["gG"]).sum() / r["gG"].norm() / a["gG"].norm()); cosD = float((r["gD"] * a["gD"]).sum() / r["gD"].norm() / a["gD"].norm())
    print("%-18s %.2f ms/step | loss dev %s | fake max|err| %.4f rms %.5f | gG cos %.6f ratio %.4f nan %d | gD cos %.6f ratio %.4f nan %d | losses after 7 steps %s" % (
        tag, r["ms"], {k: "%.1e" % v for k, v in dev_l.items()}, float((r["fake"] - a["fake"]).abs().max()), float((r["fake"] - a["fake"]).pow(2).mean().sqrt()),
        cosG, float(r["gG"].norm() / a["gG"].norm()), int(torch.isnan(r["gG"]).sum() + torch.isinf(r["gG"]).sum()),
        cosD, float(r["gD"].norm() / a["gD"].norm()), int(torch.isnan(r["gD"]).sum() + torch.isinf(r["gD"]).sum()),
        {k: round(v, 5) for k, v in r["losses_after7"].items()}))
cmp("bf16", run(torch.bfloat16))
for sc in (1.0, 256.0, 4096.0, 16384.0, 65536.0, 1048576.0):
    cmp("fp16 scale %g" % sc, run(torch.float16, sc))
# inference PSNR
x = images(1, 512, 1990)
with torch.no_grad():
    r8 = O.to_uint8_image(O.generator_forward(PG, x))
for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16), ("fp16", torch.float16)):
    ops.set_compute_dtype(dt)
    G = models.Generator(32, "none", "LeakyReLU", False); G.load_state_dict(PG); G = G.to(dev)
    y = tester.enhance(G, x.to(dev))
    q = tester.to_uint8_image(y)
    GG = tester.GraphedGenerator(G, x.shape)
    for _ in range(3): GG(x.to(dev))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): GG(x.to(dev))
    torch.cuda.synchronize()
    print("inference %s: PSNR %.2f dB SSIM %.6f  max|err| %.5f  %.3f ms/img" % (name, tester.calculate_psnr(q, r8.to(dev))[0], tester.calculate_ssim(q, r8.to(dev))[0],
          float((y.cpu() - O.generator_forward(PG, x)).abs().max()), (time.perf_counter() - t0) / 30 * 1e3))
