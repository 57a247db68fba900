"""GPU box: where the time of the 3x3 / 512-channel and 5x5 / 256-channel bf16 layers goes on conv_ring_bf16 (tile ids 10 / 11): timing-only
variants (zero-record descriptors, skipped loops) at B = 20, every unit whole (0x800)."""
import pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "gdn-pytorch_amd"))
import torch
from gdn_amd import ops
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 20


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


a = torch.randn(4096, 4096, device=dev)
for _ in range(40):
    a @ a
for name, ci, co, k, p, H, W in (("res512 k3 l3", 512, 512, 3, 1, 16, 52), ("res512 k3 l4", 512, 512, 3, 1, 8, 26), ("res256 k5", 256, 256, 5, 2, 32, 104)):
    op = ops.Conv(ci, co, k, 1, p)
    x = torch.randn(B, H, W, ci, device=dev).bfloat16()
    w = (torch.randn(k * k, co, ci, device=dev) * 0.02).bfloat16()
    gf = 2.0 * B * H * W * k * k * ci * co / 1e9
    for cfg in (10, 11):
        bn = 64 if cfg == 10 else 128
        tiles = (B * H * W + 255) // 256 * (co // bn)
        for label, knob in (("default", 0), ("no weight traffic", 2), ("no activation traffic", 4), ("neither", 6), ("no loop (prologue + epilogue only)", 1),
                            ("a third of the stages", 8), ("a third of the stages, no traffic", 14)):
            ms = min(timeit(lambda: op.fwd(x, w, stats=True, tile_cfg=cfg | 0x800 | (knob << 12))) for _ in range(2))
            print("%-13s cfg %d %-36s %7.4f ms  %7.1f TF   %d units, %.2f us per round" % (name, cfg, label, ms, gf / ms, tiles, ms * 1e3 / -(-tiles // 256)), flush=True)
