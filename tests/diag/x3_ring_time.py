"""gemm_x3 NT: the ring kernel (both operands packed, csrc/gemm_x3_ring.h) against gemm_x3_nt (fp32 A split while staged) on the
Winograd and frequency-domain shapes: bit-equality where no unit is split along K, time, TF (bf16-rate = 6 products)."""
import os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__)); root = os.path.dirname(os.path.dirname(here))
sys.path[:0] = [root, os.path.join(root, "gdn-pytorch_amd")]
from gdn_amd import ops
dev = torch.device("cuda:0")

def timed(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

_filler = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
_filler2 = torch.empty_like(_filler)

def timed_duty(fn, reps=20):
    """the GEMM between memory-bound kernels (a 256 MB copy, ~0.15 ms, before each call), as inside a training step: the chip is
    not held at its power limit by back-to-back matrix kernels; events bracket the GEMM only"""
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for _ in range(3):
        _filler2.copy_(_filler); fn()
    for e0, e1 in ev:
        _filler2.copy_(_filler); _filler.copy_(_filler2)
        e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
    return ts[len(ts) // 2]

shapes = [("wino l3 3x3/512 B20", 16, 4160, 512, 512), ("wino l4 3x3/512 B20", 16, 1040, 512, 512),
          ("fft 9x9/64 NP40 embedded", 840, 1040, 128, 128), ("fft 7x7/128 NP32 embedded", 544, 480, 256, 256),
          ("fft 5x5/256 NP16 embedded", 144, 540, 512, 512), ("odd", 5, 300, 384, 96),
          ("deepk 2048", 16, 4096, 512, 2048), ("deepk 512", 16, 4096, 512, 512), ("deepk 128", 16, 4096, 512, 128),
          ("resident 1 round K2048", 8, 1024, 1024, 2048), ("streaming 1 round K2048", 16, 1024, 512, 2048), ("resident 1 round K512", 8, 1024, 1024, 512)]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if sys.argv[1] in s[0]]
for name, bins, M, N, K in shapes:
    g = torch.Generator(device="cpu").manual_seed(1)
    A = torch.randn(bins, M, K, generator=g).to(dev); B = (torch.randn(bins, N, K, generator=g) * 0.05).to(dev)
    Ap, Bp = ops.gemm_x3_pack(A), ops.gemm_x3_pack(B)
    C0 = ops.gemm_x3_nt(A, Bp, N)
    C1 = torch.full((bins, M, N), float("nan"), device=dev)
    ops.gemm_x3_nt_packed(Ap, Bp, bins, M, N, K, out=C1)
    torch.cuda.synchronize()
    nbad = int((C0 != C1).sum()); nan = int(torch.isnan(C1).sum())
    err = float((C0 - C1).abs().max() / C0.abs().max())
    fl = 2.0 * bins * M * N * K
    t0 = timed(lambda: ops.gemm_x3_nt(A, Bp, N, out=C0))
    t1 = timed(lambda: ops.gemm_x3_nt_packed(Ap, Bp, bins, M, N, K, out=C1))
    d0 = timed_duty(lambda: ops.gemm_x3_nt(A, Bp, N, out=C0))
    d1 = timed_duty(lambda: ops.gemm_x3_nt_packed(Ap, Bp, bins, M, N, K, out=C1))
    print("%-28s between copies: nt %.3f ms, ring %.3f ms" % (name, d0, d1), flush=True)
    line = "%-28s bins %4d M %5d N %4d K %4d  nt %.3f ms (%.0f TF)  ring %.3f ms (%.0f TF bf16-rate, %.1f TF fp32-eq)  differing %d (rel %.1e) nan %d" % (
        name, bins, M, N, K, t0, 6 * fl / t0 / 1e9, t1, 6 * fl / t1 / 1e9, fl / t1 / 1e9, nbad, err, nan)
    for kn in os.environ.get("CLOCKS", "").split(","):
        if kn:
            # knob bit 4: workgroup 0 writes (shader cycles, 100 MHz ticks) behind the slabs
            os.environ["GDN_X3_RING_KNOBS"] = str(int(kn) | 16)
            nb = int(ops.lib.gdn_gemm_x3_ring_workspace_bytes())
            for mode in ("back to back", "between copies"):
                mhz = []
                for _ in range(12):
                    if mode != "back to back":
                        _filler2.copy_(_filler); _filler.copy_(_filler2)
                    ops.gemm_x3_nt_packed(Ap, Bp, bins, M, N, K, out=C1)
                    torch.cuda.synchronize() if mode != "back to back" else None
                    ws = ops.workspace(nb, dev, "x3ring")
                    v = ws[nb - 64:nb - 48].view(torch.int64).cpu()
                    mhz.append(100.0 * float(v[0]) / max(float(v[1]), 1.0))
                torch.cuda.synchronize()
                ms = timed(lambda: ops.gemm_x3_nt_packed(Ap, Bp, bins, M, N, K, out=C1)) if mode == "back to back" else \
                    timed_duty(lambda: ops.gemm_x3_nt_packed(Ap, Bp, bins, M, N, K, out=C1))
                print("    knobs %2s %-14s: %.3f ms, shader clock of workgroup 0: median %.0f MHz (min %.0f max %.0f)" % (
                    kn, mode, ms, sorted(mhz)[len(mhz) // 2], min(mhz), max(mhz)), flush=True)
            os.environ.pop("GDN_X3_RING_KNOBS")
    for kn in os.environ.get("KNOBS", "").split(","):
        if kn:
            os.environ["GDN_X3_RING_KNOBS"] = kn
            line += "  knobs %s: %.3f ms" % (kn, timed(lambda: ops.gemm_x3_nt_packed(Ap, Bp, bins, M, N, K, out=C1)))
            os.environ.pop("GDN_X3_RING_KNOBS")
    print(line, flush=True)
