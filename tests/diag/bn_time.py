"""GPU box: the BatchNorm / pointwise passes at the sizes of the B = 20 step, achieved bandwidth (algorithmic bytes / time).
usage: bn_time.py"""
import pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "gdn-pytorch_amd"))
import torch
from gdn_amd import ops
dev = torch.device("cuda:0")
B = 20
LV = [(128, 416, 64), (64, 208, 128), (32, 104, 256), (16, 52, 512), (8, 26, 512)]


def timeit(fn, reps=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for dt in (torch.bfloat16, torch.float32):
    es = 2 if dt == torch.bfloat16 else 4
    for H, W, C in LV:
        y = torch.randn(B, H, W, C, device=dev).to(dt)
        res = torch.randn(B, H, W, C, device=dev).to(dt)
        do = torch.randn(B, H, W, C, device=dev).to(dt)
        sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        gamma = torch.ones(C, device=dev)
        co = torch.stack([sc, sh, torch.zeros(C, device=dev), torch.ones(C, device=dev)])
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        n = y.numel() * es / 1e6
        out = torch.empty_like(y)
        t1 = timeit(lambda: ops.bn_apply(y, sc, sh, True, None, out=out))
        t2 = timeit(lambda: ops.bn_apply(y, sc, sh, False, res, out=out))
        t3 = timeit(lambda: ops.bn_bwd(do, y, gamma, co, True, dg, db, out_dtype=dt))
        print("%s %3dx%3dx%3d  %6.1f MB | bn_apply %6.1f us %5.2f TB/s | +residual %6.1f us %5.2f TB/s | bn_bwd (reduce+finalize+apply: 5 tensor passes) %6.1f us %5.2f TB/s"
              % ("bf16" if es == 2 else "fp32", H, W, C, n, t1, 2 * n / t1, t2, 3 * n / t2, t3, 5 * n / t3), flush=True)
