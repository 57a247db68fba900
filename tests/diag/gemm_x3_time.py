"""GPU box: the bf16 x 3 GEMM core against the fp32 MFMA per-bin GEMM on the shapes of the network (B = 20)."""
import sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "gdn-pytorch_amd")]
import torch
from gdn_amd import ops
dev = torch.device("cuda:0")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


shapes = [("wino l3 512->512", 16, 4160, 512, 512), ("wino l4 512->512", 16, 1040, 512, 512),
          ("wino2 A 256->512", 16, 20 * 6 * 18, 512, 1024), ("wino2 B 512->256", 16, 20 * 3 * 9, 1024, 512),
          ("R upconv1 512->256", 16, 20 * 17 * 53, 256, 512), ("big", 16, 8192, 1024, 1024)]
for name, bins, M, N, K in shapes:
    A = torch.randn(bins, M, K, device=dev)
    B = torch.randn(bins, N, K, device=dev) * 0.05
    Bp = ops.gemm_x3_pack(B)
    C = torch.empty(bins, M, N, device=dev)
    ms = timed(lambda: ops.gemm_x3_nt(A, Bp, N, out=C))
    mp = timed(lambda: ops.gemm_x3_pack(B))
    fl = 2.0 * bins * M * N * K
    ref = torch.bmm(A[:1].double(), B[:1].double().transpose(1, 2))
    err = float((C[:1].double() - ref).abs().max() / ref.abs().max())
    line = "%-22s M %5d N %4d K %4d: x3 %.3f ms = %6.1f TF fp32-equiv (%.0f TF bf16), pack B %.3f ms, err %.1e" % (
        name, M, N, K, ms, fl / ms / 1e9, 6 * fl / ms / 1e9, mp, err)
    print(line, flush=True)
# the fp32 MFMA kernel on the level-3 / level-4 shapes (through the Winograd measurement hook)
for lvl, (H, W) in ((3, (16, 52)), (4, (8, 26))):
    op = ops.Conv(512, 512, 3, 1, 1)
    tiles = 20 * (H // 2) * (W // 2)
    V = torch.randn(16, tiles, 512, device=dev); U = torch.randn(16, 512, 512, device=dev) * 0.05
    Mo = torch.empty(16, tiles, 512, device=dev)
    ms = timed(lambda: op.wino_gemm_only(V, U, Mo, 20, H, W))
    print("fp32 MFMA wino_gemm level %d (M %d): %.3f ms = %.1f TF" % (lvl, tiles, ms, 2.0 * 16 * tiles * 512 * 512 / ms / 1e9))
# the reduction (weight-gradient) GEMM: level 3 / level 4 of the 512-channel layers
for name, T in (("wino tn l3", 4160), ("wino tn l4", 1040)):
    A = torch.randn(16, T, 512, device=dev); B = torch.randn(16, T, 512, device=dev)
    for ns in (1, 2, 4):
        ms = timed(lambda: ops.gemm_x3_tn(A, B, ns))
        print("%s T %d splits %d: x3 %.3f ms = %.1f TF fp32-equiv" % (name, T, ns, ms, 2.0 * 16 * T * 512 * 512 / ms / 1e9), flush=True)
