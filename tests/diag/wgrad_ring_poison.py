"""GPU box: wgrad_ring_bf16 (cfg 4) with its split-K slab workspace pre-filled with NaN bit patterns (every slab element the reduce
pass reads must have been written by the kernel) and with the whole LDS of every CU filled with NaNs before the launch
(tests/diag/lds_fill.hip: the kernel must not read LDS it never wrote).  usage: wgrad_ring_poison.py"""
import ctypes
import pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "gdn-pytorch_amd")); sys.path.insert(0, str(ROOT / "tests"))
import torch
from gdn_amd import ops
from test_hip_bf16 import RING_WGRAD_CASES
dev = torch.device("cuda:0")
L = [(128, 416), (64, 208), (32, 104)]
BIG = [("res64 k9 B20", 64, 64, 9, 4, False, 20, *L[0]), ("res128 k7 B20", 128, 128, 7, 3, False, 20, *L[1]), ("res256 k5 B20", 256, 256, 5, 2, False, 20, *L[2]),
       ("up3 k7 refl B20", 128, 64, 7, 3, True, 20, *L[0]), ("res64 k9 B2", 64, 64, 9, 4, False, 2, *L[0]), ("res64 k9 B1 64x96", 64, 64, 9, 4, False, 1, 64, 96)]
g = torch.Generator(device=dev).manual_seed(0)
lf = ctypes.CDLL(str(ROOT / "tests/diag/_build/liblds_fill.so"))
lf.lds_fill.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
sink = torch.zeros(4, dtype=torch.int32, device=dev)
for name, ci, co, k, p, refl, B, H, W in list(RING_WGRAD_CASES) + BIG:
    op = ops.Conv(ci, co, k, 1, p, reflect=refl)
    Ho, Wo = H + 2 * p - k + 1, W + 2 * p - k + 1
    x = torch.randn(B, H, W, ci, device=dev, generator=g).bfloat16()
    gy = torch.randn(B, Ho, Wo, co, device=dev, generator=g).bfloat16()
    dw0 = torch.zeros(k * k, co, ci, device=dev)
    op.wgrad(x, gy, dw0, cfg=4)
    torch.cuda.synchronize()
    for b in ops._ws_cache.values():
        b.fill_(0xFF)
    dw1 = torch.zeros_like(dw0)
    op.wgrad(x, gy, dw1, cfg=4)
    torch.cuda.synchronize()
    bad = int((~torch.isfinite(dw1)).sum())
    for pat in (0xFFFFFFFF, 0x7F007F00):
        assert lf.lds_fill(pat, sink.data_ptr(), ops.stream()) == 0
        dw2 = torch.zeros_like(dw0)
        op.wgrad(x, gy, dw2, cfg=4)
        torch.cuda.synchronize()
        if not torch.equal(dw0, dw2):
            nb = int((~torch.isfinite(dw2)).sum())
            idx = (dw0 != dw2).nonzero()
            print("      LDS pattern %#x: %d non-finite, %d differing; taps %s co %d..%d ci %d..%d" % (pat, nb, idx.shape[0], sorted(set(idx[:, 0].tolist()))[:12],
                  int(idx[:, 1].min()), int(idx[:, 1].max()), int(idx[:, 2].min()), int(idx[:, 2].max())), flush=True)
    print("%-24s non-finite after poisoning: %d of %d   (first run finite: %s, equal: %s)" % (name, bad, dw1.numel(), bool(torch.isfinite(dw0).all()),
          bool(torch.equal(dw0, dw1))), flush=True)
    if bad:
        idx = (~torch.isfinite(dw1)).nonzero()
        print("      taps with NaN:", sorted(set(idx[:, 0].tolist()))[:20], " co range", int(idx[:, 1].min()), int(idx[:, 1].max()), " ci range", int(idx[:, 2].min()), int(idx[:, 2].max()))
