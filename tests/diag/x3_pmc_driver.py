"""GPU box: launches the bf16 x 3 GEMMs of the level-3 512-channel Winograd layer (B = 20) a few times, for rocprofv3 passes."""
import sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "gdn-pytorch_amd")]
import torch
from gdn_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
M, N, K = 4160, 512, 512
A = torch.randn(16, M, K, device=dev, generator=g)
B = torch.randn(16, N, K, device=dev, generator=g) * 0.05
D = torch.randn(16, M, N, device=dev, generator=g)
Bp = ops.gemm_x3_pack(B)
C = torch.empty(16, M, N, device=dev)
for _ in range(10):
    ops.gemm_x3_nt(A, Bp, N, out=C)
    ops.gemm_x3_tn(D, A, 2)
torch.cuda.synchronize()
print("done")
