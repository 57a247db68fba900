"""GPU box: conv_ring2_bf16 (tile id 12) forward / data gradient launched N times on the B = 20 layers with other kernels in between:
every result must equal the first bit for bit (a race on the two-stage LDS images would show as a sporadic difference).
usage: ring2_determinism.py [N]"""
import pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "gdn-pytorch_amd"))
import torch
from gdn_amd import ops
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = torch.Generator(device=dev).manual_seed(0)
noise = torch.randn(4096, 4096, device=dev)
for name, ci, co, k, refl, H, W in (("res64 k9", 64, 64, 9, False, 128, 416), ("R up3 k7 refl", 128, 64, 7, True, 128, 416), ("res128 k7 (N = 128 on 64-col tiles)", 128, 128, 7, False, 64, 208)):
    op = ops.Conv(ci, co, k, 1, k // 2, reflect=refl)
    x = torch.randn(20, H, W, ci, device=dev, generator=g).bfloat16()
    w = (torch.randn(k * k, co, ci, device=dev, generator=g) * 0.02).bfloat16()
    wt = ops.transpose_taps(w)
    gy = torch.randn(20, H, W, co, device=dev, generator=g).bfloat16()
    y0, s0 = op.fwd(x, w, stats=True, tile_cfg=12)
    d0 = op.dgrad(gy, wt, (H, W), addsrc=x, tile_cfg=12) if not refl else None
    bad = 0
    for i in range(N):
        if i % 3 == 0:
            noise @ noise                                   # (another kernel's LDS contents and clock state in between)
        y, s = op.fwd(x, w, stats=True, tile_cfg=12)
        bad += int(not (torch.equal(y, y0) and torch.equal(s, s0)))
        if d0 is not None:
            bad += int(not torch.equal(op.dgrad(gy, wt, (H, W), addsrc=x, tile_cfg=12), d0))
    torch.cuda.synchronize()
    print("%-40s %d launches each: %d differing results, finite: %s" % (name, N, bad, bool(torch.isfinite(y0.float()).all())), flush=True)
