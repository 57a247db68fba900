"""GPU box: the bf16 large-window layers back to back on the same buffers (what tools/tune_conv.py times: inputs and outputs
stay in the 256 MB Infinity Cache) against the same launches with 1 GB of unrelated traffic in between (what a training
step is)."""
import sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "gdn-pytorch_amd")]
import torch
from gdn_amd import ops
dev = torch.device("cuda:0")
B = 20
junk = torch.empty(1 << 28, dtype=torch.float32, device=dev)      # 1 GB
for (name, c, k, H, W) in (("res64 k9", 64, 9, 128, 416), ("res128 k7", 128, 7, 64, 208), ("res256 k5", 256, 5, 32, 104), ("res512 k3", 512, 3, 16, 52)):
    op = ops.Conv(c, c, k, 1, k // 2)
    x = torch.randn(B, H, W, c, device=dev).bfloat16()
    w = (torch.randn(k * k, c, c, device=dev) * 0.02).bfloat16()
    gf = 2.0 * B * H * W * k * k * c * c / 1e9
    res = []
    for cold in (False, True):
        ts = []
        for it in range(8):
            if cold:
                junk.mul_(1.0001)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            op.fwd(x, w, stats=True)
            e1.record()
            torch.cuda.synchronize()
            if it >= 2:
                ts.append(e0.elapsed_time(e1))
        res.append(sum(ts) / len(ts))
    print("%-10s %6.1f GFLOP: back to back %.3f ms (%.0f TF), after 1 GB of other traffic %.3f ms (%.0f TF)" % (
        name, gf, res[0], gf / res[0], res[1], gf / res[1]), flush=True)
