"""conv_c1 kernels vs the direct kernels on identical inputs (max abs difference, location)."""
import sys, pathlib, torch
R = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(R)); sys.path.insert(0, str(R / "gdn-pytorch_amd"))
from gdn_amd import ops
gpu = torch.device("cuda:0")
torch.manual_seed(0)
for (B, H, W) in ((1, 32, 64), (2, 128, 416)):
    dpre = torch.randn(B, H, W, 1, device=gpu)
    x = torch.randn(B, H, W, 64, device=gpu)
    w = torch.randn(81, 1, 64, device=gpu) * 0.05          # ConvT head [tap][Cout=1][Cin=64]
    op = ops.Conv(64, 1, 9, 1, 4, transposed=True)
    wt = ops.transpose_taps(w)
    d0 = op.dgrad(dpre, wt, (H, W))
    d1 = ops.conv_c1_fwd(dpre, w, flip=False)
    e = (d0 - d1).abs()
    i = int(e.argmax())
    print("head(ConvT) dgrad %dx%dx%d: max|diff| %.3e at %s, max|ref| %.3e" % (B, H, W, float(e.max()), list(torch.unravel_index(torch.tensor(i), e.shape)), float(d0.abs().max())))
    dw0 = torch.empty_like(w); dw1 = torch.empty_like(w)
    op.wgrad(x, dpre, dw0)
    ops.conv_c1_wgrad(dpre, x, dw1, flip=False)
    print("   wgrad: max|diff| %.3e, max|ref| %.3e" % (float((dw0 - dw1).abs().max()), float(dw0.abs().max())))
    # first conv (reflect)
    opf = ops.Conv(1, 64, 9, 1, 4, reflect=True)
    wf = torch.randn(81, 64, 1, device=gpu) * 0.1
    y0, s0 = opf.fwd(dpre, wf, stats=True)
    y1, s1 = ops.conv_c1_fwd(dpre, wf, reflect=True, stats=True)
    print("   first conv fwd: max|diff| %.3e; stats sum diff %.3e / %.3e" % (float((y0 - y1).abs().max()),
          float((s0.double().sum(0) - s1.double().sum(0)).abs().max()), float(s0.double().sum(0).abs().max())))
    g64 = torch.randn(B, H, W, 64, device=gpu)
    dwa = torch.empty_like(wf); dwb = torch.empty_like(wf)
    opf.wgrad(dpre, g64, dwa)
    ops.conv_c1_wgrad(dpre, g64, dwb, reflect=True)
    print("   first conv wgrad: max|diff| %.3e, max|ref| %.3e" % (float((dwa - dwb).abs().max()), float(dwa.abs().max())))


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


B, H, W = 20, 128, 416
dpre = torch.randn(B, H, W, 1, device=gpu)
x = torch.randn(B, H, W, 64, device=gpu)
w = torch.randn(81, 1, 64, device=gpu) * 0.05
dw = torch.empty_like(w)
op = ops.Conv(64, 1, 9, 1, 4, transposed=True)
wt = ops.transpose_taps(w)
opf = ops.Conv(1, 64, 9, 1, 4, reflect=True)
wf = torch.randn(81, 64, 1, device=gpu) * 0.1
print("B=20 128x416 (us):  c1 fwd+stats %.0f (direct %.0f) | head dgrad c1 %.0f (direct %.0f) | wgrad c1 %.0f (direct thin %.0f)" % (
    timed(lambda: ops.conv_c1_fwd(dpre, wf, reflect=True, stats=True)), timed(lambda: opf.fwd(dpre, wf, stats=True)),
    timed(lambda: ops.conv_c1_fwd(dpre, w, flip=False)), timed(lambda: op.dgrad(dpre, wt, (H, W))),
    timed(lambda: ops.conv_c1_wgrad(dpre, x, dw)), timed(lambda: op.wgrad(x, dpre, dw))))
