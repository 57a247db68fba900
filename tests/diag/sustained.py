"""GPU box: one bf16 convolution launched 400 times back to back: time per launch in windows of 20 -- does the rate fall when
the matrix pipes stay busy (power management), as the in-step times of the same kernel suggest?"""
import sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "gdn-pytorch_amd")]
import torch
from gdn_amd import ops
dev = torch.device("cuda:0")
B = 20
for (name, c, k, H, W, dt) in (("res128 k7 bf16", 128, 7, 64, 208, torch.bfloat16), ("res64 k9 bf16", 64, 9, 128, 416, torch.bfloat16)):
    op = ops.Conv(c, c, k, 1, k // 2)
    x = torch.randn(B, H, W, c, device=dev).to(dt)
    w = (torch.randn(k * k, c, c, device=dev) * 0.02).to(dt)
    y, st = op.fwd(x, w, stats=True)
    gf = 2.0 * B * H * W * k * k * c * c / 1e9
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    evs[0].record()
    for win in range(20):
        for _ in range(20):
            op.fwd(x, w, stats=True, out=y, stats_out=st)
        evs[win + 1].record()
    torch.cuda.synchronize()
    ts = [evs[i].elapsed_time(evs[i + 1]) / 20 for i in range(20)]
    print("%-16s ms per launch, windows of 20: %s  -> %.0f TF first window, %.0f TF last" % (
        name, " ".join("%.3f" % t for t in ts), gf / ts[0], gf / ts[-1]), flush=True)
