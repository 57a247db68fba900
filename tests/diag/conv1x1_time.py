"""GPU box: the 1x1 convolutions with a fused channel concat (AutoEncoder_2 / AutoEncoder conv1x1_c: cat((up, skip), 1) -> c
channels) on the direct kernel: time against the bytes they move."""
import sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "gdn-pytorch_amd")]
import torch
from gdn_amd import ops
dev = torch.device("cuda:0")


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for dt in (torch.float32, torch.bfloat16):
    for (c, H, W) in ((64, 128, 416), (128, 64, 208), (256, 32, 104), (512, 16, 52)):
        x1 = torch.randn(B, H, W, c, device=dev).to(dt)
        x2 = torch.randn(B, H, W, c, device=dev).to(dt)
        xc = torch.cat((x1, x2), 3).contiguous()
        w = (torch.randn(1, c, 2 * c, device=dev) * 0.05).to(dt)
        op = ops.Conv(2 * c, c, 1, 1, 0)
        t_cat = timeit(lambda: op.fwd(x1, w, x2=x2, stats=True))
        t_one = timeit(lambda: op.fwd(xc, w, stats=True))
        gb = B * H * W * c * 3 * x1.element_size() / 1e9
        gf = 2.0 * B * H * W * c * 2 * c / 1e9
        print("%s 1x1 %4d+%-4d -> %-4d at %3dx%-3d: fused concat %.3f ms, one tensor %.3f ms  (%.2f GB: %.0f GB/s; %.1f GFLOP: %.1f TF)" % (
            str(dt).split(".")[-1], c, c, c, H, W, t_cat, t_one, gb, gb / t_cat * 1e3, gf, gf / t_cat))
