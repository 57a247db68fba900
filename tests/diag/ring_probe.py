"""GPU box: where the ring kernel's step time goes -- timing-only variants of the 9x9 / 64-channel layer at B = 20:
no weight traffic, no activation traffic, neither, prefetch distance 2 / 4 / 6."""
import pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "gdn-pytorch_amd"))
import torch
from gdn_amd import ops
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def timeit(fn, reps=8):
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
for name, ci, co, k, p, H, W in (("res64 k9", 64, 64, 9, 4, 128, 416), ("res128 k7", 128, 128, 7, 3, 64, 208)):
    op = ops.Conv(ci, co, k, 1, p)
    x = torch.randn(B, H, W, ci, device=dev).bfloat16()
    w = (torch.randn(k * k, co, ci, device=dev) * 0.02).bfloat16()
    gf = 2.0 * B * H * W * k * k * ci * co / 1e9
    tiles = (B * H * W + 255) // 256 * (co // 64)
    for label, knob in (("default (DP 4)", 0), ("no weight traffic", 2), ("no activation traffic", 4), ("neither", 6), 
                        ("no loop (prologue + epilogue only)", 1), ("a third of the stages", 8), ("a third of the stages, no traffic", 14), ("no loop, no traffic", 7)):
        for rep in range(2):
            ms = timeit(lambda: op.fwd(x, w, stats=True, tile_cfg=10 | (knob << 12)))
        print("%-10s %-24s %7.3f ms  %7.1f TF   %.2f us per tile-round" % (name, label, ms, gf / ms, ms * 1e3 / -(-tiles // 256)), flush=True)
    for cfg in (9, 11):
        if cfg == 11 and co % 128:
            continue
        ms = timeit(lambda: op.fwd(x, w, stats=True, tile_cfg=cfg))
        print("%-10s cfg %-20d %7.3f ms  %7.1f TF" % (name, cfg, ms, gf / ms), flush=True)
