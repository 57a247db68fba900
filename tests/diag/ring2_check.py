"""GPU box: conv_ring2_bf16 (cfg 12: 512 x 64 tiles, 32-channel slabs) against conv_ring_bf16 (cfg 10 / 11) and the row-patch kernel
(cfg 9): difference of outputs / BatchNorm partial sums / data gradient with residual, and interleaved best-of-N times.
usage: ring2_check.py [B] [rounds]"""
import pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "gdn-pytorch_amd"))
import torch
from gdn_amd import ops

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
L = [(128, 416), (64, 208), (32, 104), (16, 52)]
SHAPES = [("res64 k9", 64, 64, 9, 4, False, *L[0]), ("res128 k7", 128, 128, 7, 3, False, *L[1]), ("res256 k5", 256, 256, 5, 2, False, *L[2]),
          ("res512 k3 l3", 512, 512, 3, 1, False, *L[3]), ("R up3 k7 refl", 128, 64, 7, 3, True, *L[0]), ("R up2 k5 refl", 256, 128, 5, 2, True, *L[1])]
SMALL = [("k9 20x40", 64, 64, 9, 4, False, 1, 20, 40), ("k7 refl 12x64", 128, 64, 7, 3, True, 1, 12, 64), ("k5 10x60", 64, 128, 5, 2, False, 2, 10, 60),
         ("k3 9x52", 128, 192, 3, 1, False, 3, 9, 52), ("k9 33x250", 64, 64, 9, 4, False, 2, 33, 250), ("k5 refl 7x59", 64, 64, 5, 2, True, 2, 7, 59)]


def timeit(fn, reps=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def rel(a, b):
    return (a.float() - b.float()).abs().max().item() / b.float().abs().max().item()


g = torch.Generator(device=dev).manual_seed(0)
print("-- small shapes: cfg 12 vs the tap-major kernel (cfg 1)")
for name, ci, co, k, p, refl, b, H, W in SMALL:
    op = ops.Conv(ci, co, k, 1, p, reflect=refl)
    x = torch.randn(b, H, W, ci, device=dev, generator=g).bfloat16()
    w = (torch.randn(k * k, co, ci, device=dev, generator=g) * 0.02).bfloat16()
    wt = ops.transpose_taps(w)
    Ho, Wo = H + 2 * p - k + 1, W + 2 * p - k + 1
    gy = torch.randn(b, Ho, Wo, co, device=dev, generator=g).bfloat16()
    add = torch.randn(b, H, W, ci, device=dev, generator=g).bfloat16()
    try:
        y0, s0 = op.fwd(x, w, stats=True, tile_cfg=1 | 0x800)
        y1, s1 = op.fwd(x, w, stats=True, tile_cfg=12)
        msg = "dy %.1e ds %.1e" % (rel(y1, y0), rel(s1.double().sum(0), s0.double().sum(0)))
        if not refl:
            d0 = op.dgrad(gy, wt, (H, W), addsrc=add, tile_cfg=1 | 0x800)
            d1 = op.dgrad(gy, wt, (H, W), addsrc=add, tile_cfg=12)
            msg += " dd %.1e" % rel(d1, d0)
    except Exception as e:
        msg = str(e)[:80]
    print("%-16s %s" % (name, msg), flush=True)

a = torch.randn(4096, 4096, device=dev)
for _ in range(40):
    a @ a
torch.cuda.synchronize()
print("-- B = %d" % B)
for name, ci, co, k, p, refl, H, W in SHAPES:
    op = ops.Conv(ci, co, k, 1, p, reflect=refl)
    x = torch.randn(B, H, W, ci, device=dev, generator=g).bfloat16()
    w = (torch.randn(k * k, co, ci, device=dev, generator=g) * 0.02).bfloat16()
    wt = ops.transpose_taps(w)
    gy = torch.randn(B, H, W, co, device=dev, generator=g).bfloat16()
    add = torch.randn(B, H, W, ci, device=dev, generator=g).bfloat16()
    gf = 2.0 * B * H * W * k * k * ci * co / 1e9
    y0, s0 = op.fwd(x, w, stats=True, tile_cfg=0)
    d0 = op.dgrad(gy, wt, (H, W), addsrc=add, tile_cfg=0) if not refl else None
    y1, s1 = op.fwd(x, w, stats=True, tile_cfg=12)
    msg = "dy %.1e ds %.1e" % (rel(y1, y0), rel(s1.double().sum(0), s0.double().sum(0)))
    if d0 is not None:
        msg += " dd %.1e" % rel(op.dgrad(gy, wt, (H, W), addsrc=add, tile_cfg=12), d0)
    cfgs = [0, 12, 0x800 | 10, 0x800 | 12] + ([0x800 | 11] if co % 128 == 0 else [])
    tf = {c: 1e9 for c in cfgs}
    td = {c: 1e9 for c in cfgs}
    for _ in range(R):
        for c in cfgs:
            tf[c] = min(tf[c], timeit(lambda: op.fwd(x, w, stats=True, tile_cfg=c)))
            if not refl:
                td[c] = min(td[c], timeit(lambda: op.dgrad(gy, wt, (H, W), addsrc=add, tile_cfg=c)))
    print("%-14s %7.1f GF %s" % (name, gf, msg))
    for c in cfgs:
        print("      cfg %#6x  fwd %.4f ms %6.1f TF | dgrad %.4f ms %6.1f TF" % (c, tf[c], gf / tf[c], td[c], gf / td[c] if td[c] < 1e8 else 0), flush=True)
