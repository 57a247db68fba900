import sys, os
sys.path[:0] = ["/root/repo", "/root/repo/gdn-pytorch_amd"]
import torch
from gdn_amd import ops
dev = torch.device("cuda:0")
filler = torch.empty(256 << 20, dtype=torch.uint8, device=dev); filler2 = torch.empty_like(filler)
def timed_duty(fn, reps=20):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for _ in range(3):
        filler2.copy_(filler); fn()
    for e0, e1 in ev:
        filler2.copy_(filler); filler.copy_(filler2)
        e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
    return ts[len(ts) // 2]
for name, bins, M, N, K in (("F4 l3", 36, 1040, 512, 512), ("F4 l4", 36, 280, 512, 512), ("F2 l3", 16, 4160, 512, 512)):
    A = torch.randn(bins, M, K, device=dev); B = torch.randn(bins, N, K, device=dev) * 0.05
    Bp = ops.gemm_x3_pack(B); C = torch.empty(bins, M, N, device=dev); Ap = ops.gemm_x3_pack(A)
    for v in ("0", "4", "8", "64"):
        os.environ["GDN_X3_NT"] = v
        ms = timed_duty(lambda: ops.gemm_x3_nt(A, Bp, N, out=C))
        print("%-6s nt variant %2s: %.3f ms" % (name, v, ms), flush=True)
    os.environ.pop("GDN_X3_NT")
    D = torch.randn(bins, M, N, device=dev)
    for ns in (1, 2, 3):
        if M // ns >= 256 or ns == 1:
            print("%-6s tn splits %d: %.3f ms" % (name, ns, timed_duty(lambda: ops.gemm_x3_tn(D, A, ns))), flush=True)
