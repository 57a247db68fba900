"""Winograd F(3x3,2x2) vs the direct MFMA kernels on G's eight 4x4 stride-2 layers at B=20: ms per forward / backward."""
import sys, pathlib, torch
R = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(R / "gdn-pytorch_amd"))
from gdn_amd import ops

dev = torch.device("cuda:0")
B = 20
LAYERS = [("downconv1", 64, 128, 128, 416, False), ("downconv2", 128, 256, 64, 208, False), ("downconv3", 256, 512, 32, 104, False),
          ("downconv4", 512, 512, 16, 52, False), ("upconv0", 512, 512, 8, 26, True), ("upconv1", 512, 256, 16, 52, True),
          ("upconv2", 256, 128, 32, 104, True), ("upconv3", 128, 64, 64, 208, True)]


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print("%-10s %6s %6s | %8s %8s | %8s %8s   (ms; direct bwd = dgrad + wgrad)" % ("layer", "Cin", "Cout", "fwd dir", "fwd wino", "bwd dir", "bwd wino"))
tot = [0.0] * 4
for name, ci, co, H, W, tr in LAYERS:
    op = ops.Conv(ci, co, 4, 2, 1, reflect=not tr, transposed=tr)
    Ho, Wo = (2 * H, 2 * W) if tr else (H // 2, W // 2)
    x = torch.randn(B, H, W, ci, device=dev)
    w = torch.randn(16, co, ci, device=dev) * 0.02
    g = torch.randn(B, Ho, Wo, co, device=dev)
    dw = torch.empty_like(w)
    wt = ops.transpose_taps(w)
    y, st, sv = op.wino2_fwd(x, w, stats=True, state=True)
    t_fd = timed(lambda: op.fwd(x, w, stats=True))
    t_fw = timed(lambda: op.wino2_fwd(x, w, stats=True, state=True))
    t_bd = timed(lambda: (op.dgrad(g, wt, (H, W)), op.wgrad(x, g, dw)))
    t_bw = timed(lambda: op.wino2_bwd(g, w, (H, W), state=sv, dw_tap=dw))
    for i, v in enumerate((t_fd, t_fw, t_bd, t_bw)):
        tot[i] += v
    print("%-10s %6d %6d | %8.3f %8.3f | %8.3f %8.3f" % (name, ci, co, t_fd, t_fw, t_bd, t_bw))
print("%-24s | %8.3f %8.3f | %8.3f %8.3f" % ("sum", *tot))
