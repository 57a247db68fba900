"""GPU box: launches the bf16 x 3 GEMMs of the level-3 512-channel Winograd layer (B = 20) a few times, for rocprofv3 passes:
the F(4x4,3x3) plan's shapes (36 bins x [1040 x 512] x [512 x 512], round 4) and the F(2x2,3x3) ones (16 bins x [4160 ...])."""
import sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "gdn-pytorch_amd")]
import torch
from gdn_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for bins, M, ns in ((36, 1040, 2), (16, 4160, 2)):
    N = K = 512
    A = torch.randn(bins, M, K, device=dev, generator=g)
    B = torch.randn(bins, N, K, device=dev, generator=g) * 0.05
    D = torch.randn(bins, M, N, device=dev, generator=g)
    Bp = ops.gemm_x3_pack(B)
    C = torch.empty(bins, M, N, device=dev)
    for _ in range(10):
        ops.gemm_x3_nt(A, Bp, N, out=C)
        ops.gemm_x3_tn(D, A, ns)
torch.cuda.synchronize()
print("done")
