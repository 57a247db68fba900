import sys, pathlib
sys.path[:0] = ["/root/repo", "/root/repo/gdn-pytorch_amd"]
import torch
from gdn_amd import ops
dev = torch.device("cuda:0")
def timed(fn, reps=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for name, bins, M, N, K in (("9x9/64 NP40 real-embedded", 840, 1040, 128, 128), ("7x7/128 NP32", 544, 480, 256, 256), ("5x5/256 NP16", 144, 540, 512, 512)):
    A = torch.randn(bins, M, K, device=dev); B = torch.randn(bins, N, K, device=dev) * 0.05
    Bp = ops.gemm_x3_pack(B); C = torch.empty(bins, M, N, device=dev)
    for v in ("0", "4", "64"):
        import os; os.environ["GDN_X3_NT"] = v
        ms = timed(lambda: ops.gemm_x3_nt(A, Bp, N, out=C))
        fl = 2.0 * bins * M * N * K
        print("%-28s bins %4d M %5d N %4d K %4d variant %2s: %.3f ms = %.1f TF fp32-equiv (4-mult embedding); Gauss-equivalent would be ~%.3f ms" % (name, bins, M, N, K, v, ms, fl / ms / 1e9, 0.75 * ms), flush=True)
