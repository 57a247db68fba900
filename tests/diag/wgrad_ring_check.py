"""GPU box: the LDS-DMA ring weight-gradient kernel (cfg 4, csrc/wgrad_ring.h) against the round-1 bf16 kernel (cfg 1/2/3 classes,
automatic = best of them) on the stride-1 layers of the two networks: max difference of the fp32 dW (both sum the same bf16
products in fp32, in another order), a float64 torch check on small odd shapes, and the time of each.
usage: wgrad_ring_check.py [B] [name-filter]"""
import pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "gdn-pytorch_amd"))
import torch
import torch.nn.functional as F
from gdn_amd import ops

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
only = sys.argv[2] if len(sys.argv) > 2 else None
L = [(128, 416), (64, 208), (32, 104), (16, 52), (8, 26)]
SHAPES = [("res64 k9", 64, 64, 9, 4, False, *L[0]), ("res128 k7", 128, 128, 7, 3, False, *L[1]), ("res256 k5", 256, 256, 5, 2, False, *L[2]),
          ("res512 k3 l3", 512, 512, 3, 1, False, *L[3]), ("res512 k3 l4", 512, 512, 3, 1, False, *L[4]),
          ("R up3 k7 refl", 128, 64, 7, 3, True, *L[0]), ("R up2 k5 refl", 256, 128, 5, 2, True, *L[1]), ("R up1 k3 refl", 512, 256, 3, 1, True, *L[2]),
          ("R up0 k3 refl", 512, 512, 3, 1, True, *L[3])]
SMALL = [("k9 20x40", 64, 64, 9, 4, False, 1, 20, 40), ("k7 refl 12x64", 128, 64, 7, 3, True, 1, 12, 64), ("k5 10x40", 64, 128, 5, 2, False, 2, 10, 40),
         ("k3 8x26", 128, 192, 3, 1, False, 3, 8, 26), ("k9 33x250", 64, 64, 9, 4, False, 2, 33, 250), ("k5 refl 7x19", 64, 64, 5, 2, True, 2, 7, 19),
         ("k7 9x104", 64, 128, 7, 3, False, 2, 9, 104), ("k3 refl 5x9", 64, 64, 3, 1, True, 1, 5, 9), ("k9 pad2 24x48", 64, 64, 9, 2, False, 1, 24, 48)]


def timeit(fn, reps=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


g = torch.Generator(device=dev).manual_seed(0)
print("-- float64 check on small shapes (relative to max|dW|)")
for name, ci, co, k, p, refl, b, H, W in SMALL:
    if only and only not in name:
        continue
    op = ops.Conv(ci, co, k, 1, p, reflect=refl)
    x = torch.randn(b, H, W, ci, device=dev, generator=g).bfloat16()
    Ho, Wo = H + 2 * p - k + 1, W + 2 * p - k + 1
    gy = torch.randn(b, Ho, Wo, co, device=dev, generator=g).bfloat16()
    xc = x.double().permute(0, 3, 1, 2).cpu().requires_grad_(False)
    w = torch.zeros(co, ci, k, k, dtype=torch.float64, requires_grad=True)
    xin = F.pad(xc, (p, p, p, p), mode="reflect") if refl else xc
    y = F.conv2d(xin, w, None, 1, 0 if refl else p)
    y.backward(gy.double().permute(0, 3, 1, 2).cpu())
    ref = w.grad.permute(2, 3, 0, 1).reshape(k * k, co, ci)
    res = []
    for cfg in (0, 4):
        dw = torch.full((k * k, co, ci), float("nan"), device=dev)
        try:
            op.wgrad(x, gy, dw, cfg=cfg)
            torch.cuda.synchronize()
            res.append("cfg%d err %.2e" % (cfg, (dw.double().cpu() - ref).abs().max().item() / ref.abs().max().item()))
        except Exception as e:
            res.append("cfg%d %s" % (cfg, str(e)[:40]))
    print("%-16s %s" % (name, "  ".join(res)), flush=True)

a = torch.randn(4096, 4096, device=dev)
for _ in range(40):
    a @ a
torch.cuda.synchronize()
print("-- B = %d layers" % B)
for name, ci, co, k, p, refl, H, W in SHAPES:
    if only and only not in name:
        continue
    op = ops.Conv(ci, co, k, 1, p, reflect=refl)
    x = torch.randn(B, H, W, ci, device=dev, generator=g).bfloat16()
    gy = torch.randn(B, H, W, co, device=dev, generator=g).bfloat16()
    gf = 2.0 * B * H * W * k * k * ci * co / 1e9
    dw0 = torch.empty(k * k, co, ci, device=dev)
    dw1 = torch.full((k * k, co, ci), float("nan"), device=dev)
    old = 1 if False else (3 if (W <= 52 or (W <= 104 and k <= 3)) else 2)
    op.wgrad(x, gy, dw0, cfg=old)
    t0 = timeit(lambda: op.wgrad(x, gy, dw0, cfg=old), reps=3)
    try:
        op.wgrad(x, gy, dw1, cfg=4)
        torch.cuda.synchronize()
        d = (dw1 - dw0).abs().max().item() / dw0.abs().max().item()
        dw2 = torch.empty_like(dw1)
        op.wgrad(x, gy, dw2, cfg=4)
        same = bool((dw1 == dw2).all().item())
        t1 = timeit(lambda: op.wgrad(x, gy, dw1, cfg=4), reps=3)
        dwk = torch.empty_like(dw1)
        tk1 = timeit(lambda: op.wgrad(x, gy, dwk, cfg=4 | (1 << 12)), reps=3)      # knob: no DMA after the first stage
        tk2 = timeit(lambda: op.wgrad(x, gy, dwk, cfg=4 | (2 << 12)), reps=3)      # knob: no MFMA loop
        ab = []
        for _ in range(3):                                                         # interleaved A/B: default vs knob 4 (bare stage boundary)
            ab.append((timeit(lambda: op.wgrad(x, gy, dw1, cfg=4), reps=5), timeit(lambda: op.wgrad(x, gy, dwk, cfg=4 | (4 << 12)), reps=5)))
        print("%-16s %7.1f GF | old (cfg %d) %6.1f TF %.3f ms | ring %6.1f TF %.3f ms  (diff %.1e, bitwise repeat %s) | no-DMA %.3f ms, no-kloop %.3f ms | A/B default vs bare boundary: %s" % (
            name, gf, old, gf / t0, t0, gf / t1, t1, d, same, tk1, tk2, " ".join("%.3f/%.3f" % v for v in ab)), flush=True)
    except Exception as e:
        print("%-16s %7.1f GF | old %6.1f TF | ring: %s" % (name, gf, gf / t0, e), flush=True)
