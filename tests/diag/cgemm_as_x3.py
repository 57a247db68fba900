"""GPU box: what the per-bin complex GEMMs of the frequency-domain layers would cost as REAL GEMMs [M x 2K] x [2K x 2N] (the real
embedding of a complex product: four real products, no Gauss cancellation) on the bf16 x 3 core (gdn_gemm_x3_nt), against
cgemm_bins_kernel (fp32 MFMA, Gauss's three products) on the same plan.  Shapes: the training plans at B = 20."""
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


for name, C, k, H, W in (("9x9 / 64 ch", 64, 9, 128, 416), ("7x7 / 128 ch", 128, 7, 64, 208), ("5x5 / 256 ch", 256, 5, 32, 104)):
    op = ops.Conv(C, C, k, 1, k // 2)
    ws, bins, M, npnt = op.fft_cgemm_only(20, H, W, 0, train=True)
    t_c = min(timed(lambda: op.fft_cgemm_only(20, H, W, 0, ws=ws, train=True)) for _ in range(3))
    A = torch.randn(bins, M, 2 * C, device=dev)
    Bm = torch.randn(bins, 2 * C, 2 * C, device=dev) * 0.05
    Bp = ops.gemm_x3_pack(Bm)
    Cc = torch.empty(bins, M, 2 * C, device=dev)
    t_x = min(timed(lambda: ops.gemm_x3_nt(A, Bp, 2 * C, out=Cc)) for _ in range(3))
    t_p = timed(lambda: ops.gemm_x3_pack(Bm))
    print("%-14s %d-point plan: %4d bins x [%d x %d] . [%d x %d] complex | cgemm_bins (fp32 MFMA, Gauss) %.3f ms | real embedding on bf16 x 3 %.3f ms (+ pack %.3f ms)"
          % (name, npnt, bins, M, C, C, C, t_c, t_x, t_p), flush=True)
