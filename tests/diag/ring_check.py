"""GPU box: the LDS-DMA ring kernel (cfg 10 / 11) against the round-1 bf16 kernels (row-patch 9, tap-major 1) on the stride-1
layers of the two networks: max difference of the bf16 outputs (both sum the same bf16 products in fp32, in another order), the
BatchNorm partial sums, the data gradient with addsrc, and the time of each.  usage: ring_check.py [B]"""
import pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "gdn-pytorch_amd"))
import torch
from gdn_amd import ops

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
L = [(128, 416), (64, 208), (32, 104), (16, 52), (8, 26)]
SHAPES = [("res64 k9", 64, 64, 9, 4, False, *L[0]), ("res128 k7", 128, 128, 7, 3, False, *L[1]), ("res256 k5", 256, 256, 5, 2, False, *L[2]),
          ("res512 k3 l3", 512, 512, 3, 1, False, *L[3]), ("res512 k3 l4", 512, 512, 3, 1, False, *L[4]),
          ("R up3 k7 refl", 128, 64, 7, 3, True, *L[0]), ("R up2 k5 refl", 256, 128, 5, 2, True, *L[1]), ("R up1 k3 refl", 512, 256, 3, 1, True, *L[2])]


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


a = torch.randn(4096, 4096, device=dev)
for _ in range(40):
    a @ a
torch.cuda.synchronize()
g = torch.Generator(device=dev).manual_seed(0)
for name, ci, co, k, p, refl, H, W in SHAPES:
    op = ops.Conv(ci, co, k, 1, p, reflect=refl)
    x = torch.randn(B, H, W, ci, device=dev, generator=g).bfloat16()
    w = (torch.randn(k * k, co, ci, device=dev, generator=g) * 0.02).bfloat16()
    wt = ops.transpose_taps(w)
    gy = torch.randn(B, H, W, co, device=dev, generator=g).bfloat16()
    add = torch.randn(B, H, W, ci, device=dev, generator=g).bfloat16()
    gf = 2.0 * B * H * W * k * k * ci * co / 1e9
    ref_cfg = 9 if co <= 128 else 1
    y0, s0 = op.fwd(x, w, stats=True, tile_cfg=ref_cfg | 0x800)
    d0 = op.dgrad(gy, wt, (H, W), addsrc=add, tile_cfg=ref_cfg | 0x800) if not refl else None
    line = "%-14s %7.1f GF | ref cfg%d fwd %6.1f TF" % (name, gf, ref_cfg, gf / timeit(lambda: op.fwd(x, w, stats=True, tile_cfg=ref_cfg | 0x800)))
    for cfg in (10, 11):
        if co % (64 if cfg == 10 else 128):
            continue
        y1, s1 = op.fwd(x, w, stats=True, tile_cfg=cfg)
        torch.cuda.synchronize()
        dy = (y1.float() - y0.float()).abs().max().item() / y0.float().abs().max().item()
        ds = (s1.double().sum(0) - s0.double().sum(0)).abs().max().item() / s0.double().sum(0).abs().max().item()
        msg = " | cfg%d fwd %6.1f TF (dy %.1e ds %.1e)" % (cfg, gf / timeit(lambda: op.fwd(x, w, stats=True, tile_cfg=cfg)), dy, ds)
        if d0 is not None:
            d1 = op.dgrad(gy, wt, (H, W), addsrc=add, tile_cfg=cfg)
            dd = (d1.float() - d0.float()).abs().max().item() / d0.float().abs().max().item()
            msg += " dgrad %6.1f TF (dd %.1e)" % (gf / timeit(lambda: op.dgrad(gy, wt, (H, W), addsrc=add, tile_cfg=cfg)), dd)
        line += msg
    print(line, flush=True)
