"""GPU box: conv_ring_bf16 forward (+stats) and data gradient (+residual) of the stride-1 layers at B = 20, best of N interleaved
rounds -- run once per library (GDN_HIP_LIB) and compare.  usage: ring_ab.py [rounds]"""
import pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "gdn-pytorch_amd"))
import torch
from gdn_amd import ops
dev = torch.device("cuda:0")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4
L = [(128, 416), (64, 208), (32, 104), (16, 52), (8, 26)]
SHAPES = [("res64 k9", 64, 64, 9, 4, False, *L[0]), ("res128 k7", 128, 128, 7, 3, False, *L[1]), ("res256 k5", 256, 256, 5, 2, False, *L[2]),
          ("res512 k3 l3", 512, 512, 3, 1, False, *L[3]), ("res512 k3 l4", 512, 512, 3, 1, False, *L[4]), ("R up3 k7 refl", 128, 64, 7, 3, True, *L[0])]


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
for _ in range(60):
    a @ a
g = torch.Generator(device=dev).manual_seed(0)
for name, ci, co, k, p, refl, H, W in SHAPES:
    op = ops.Conv(ci, co, k, 1, p, reflect=refl)
    x = torch.randn(20, H, W, ci, device=dev, generator=g).bfloat16()
    w = (torch.randn(k * k, co, ci, device=dev, generator=g) * 0.02).bfloat16()
    wt = ops.transpose_taps(w)
    gy = torch.randn(20, H, W, co, device=dev, generator=g).bfloat16()
    add = torch.randn(20, H, W, ci, device=dev, generator=g).bfloat16()
    gf = 2.0 * 20 * H * W * k * k * ci * co / 1e9
    tf = td = 1e9
    for _ in range(R):
        tf = min(tf, timeit(lambda: op.fwd(x, w, stats=True)))
        if not refl:
            td = min(td, timeit(lambda: op.dgrad(gy, wt, (H, W), addsrc=add)))
    print("%-14s fwd %.4f ms %6.1f TF | dgrad %.4f ms %6.1f TF" % (name, tf, gf / tf, td, gf / td if td < 1e8 else 0), flush=True)
