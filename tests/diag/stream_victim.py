"""GPU box: the neighbour experiment of tests/diag/torch_victim.py inside ONE process: torch.fft.rfft2 on one stream while a
second stream of the same process runs the barrier-paced bf16 matrix loop (tests/diag/mfma_neighbour.hip, variant 5) or
gemm_x3_nt.  Does the interference need two processes?"""
import ctypes, sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "gdn-pytorch_amd")]
import torch
from gdn_amd import ops
dev = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "mfma5"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
g = torch.Generator(device=dev).manual_seed(3)
img = torch.randn(8, 64, 64, 64, device=dev, generator=g)
x = torch.randn(64, 1 << 16, device=dev, generator=g)
side = torch.cuda.Stream()
mf = ctypes.CDLL(str(ROOT / "tests/diag/_build/libmfma_neighbour.so"))
mf.mfma_loop.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
sink = torch.zeros(4, device=dev)
A = torch.randn(16, 2080, 512, device=dev, generator=g)
B = torch.randn(16, 512, 512, device=dev, generator=g) * 0.05
Bp = ops.gemm_x3_pack(B)
C = torch.empty(16, 2080, 512, device=dev)
torch.cuda.synchronize()
ref, bad = {}, {"fft": 0, "elementwise": 0}
for it in range(n):
    if kind != "none":
        with torch.cuda.stream(side):
            for _ in range(6):
                if kind.startswith("mfma"):
                    mf.mfma_loop(int(kind[4:]), 1024, 200, sink.data_ptr(), side.cuda_stream)
                else:
                    ops.gemm_x3_nt(A, Bp, 512, out=C)
    outs = {"fft": torch.view_as_real(torch.fft.rfft2(img)), "elementwise": torch.tanh(x * 1.0001 + 0.5) * x}
    torch.cuda.synchronize()
    for name, t in outs.items():
        if name not in ref:
            ref[name] = t.clone()
        elif not torch.equal(t, ref[name]):
            bad[name] += 1
            if bad[name] <= 2:
                d = (t - ref[name]).abs()
                print("  iter %d %s: %d elements differ, max %.3e" % (it, name, int((d > 0).sum()), float(d.max())), flush=True)
print("one process, second stream running %s: repeats that differ of %d: %s" % (kind, n, bad), flush=True)
