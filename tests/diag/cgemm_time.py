"""GPU box: the per-bin complex GEMMs of the three frequency-domain layer shapes alone (gdn_fftconv_cgemm hook): forward, data
gradient, weight-gradient reduction -- ms, fp32 TFLOP/s (three real products per complex product), spectrum GB/s.  usage: cgemm_time.py [B]"""
import pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "gdn-pytorch_amd"))
import torch
from gdn_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


a = torch.randn(4096, 4096, device="cuda")
for _ in range(30):
    a @ a
for (C, k, H, W, train) in [(64, 9, 128, 416, True), (128, 7, 64, 208, True), (256, 5, 32, 104, True), (64, 9, 128, 416, False)]:
    op = ops.Conv(C, C, k, 1, k // 2)
    ws, bins, M, npnt = op.fft_cgemm_only(B, H, W, 0, train=train)
    flop = 3 * 2.0 * bins * M * C * C
    by = 2 * M * bins * C * 8
    best = [1e9] * 3
    for _ in range(3):
        for w in range(3):
            best[w] = min(best[w], timeit(lambda w=w: op.fft_cgemm_only(B, H, W, w, ws=ws, train=train)))
    print("C=%3d k=%d np=%d bins=%d M=%d | %s" % (C, k, npnt, bins, M, " | ".join(
        "%s %.3f ms %5.1f TF %4.2f TB/s" % (n, t, flop / t / 1e9, by / t / 1e9) for n, t in zip(("fwd", "dgrad", "tn"), best))), flush=True)
