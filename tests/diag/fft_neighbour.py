"""GPU box: the frequency-domain layers of the 32 x 64 test model, repeated on fixed inputs while a NEIGHBOUR process loops
over one kind of kernel on the same GPU (tests/diag/dp_solo.py aggressor).  Every output (y, BatchNorm partials, saved
spectrum, dx, dw) is compared with the first repeat: which kernel of the chain is the one that is disturbed?"""
import os, subprocess, sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[2]


def victim(n):
    sys.path[:0] = [str(ROOT), str(ROOT / "gdn-pytorch_amd")]
    import torch
    from gdn_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    layers = []
    for (c, k, B, H, W) in ((256, 5, 2, 8, 16), (64, 9, 2, 32, 64), (128, 7, 2, 16, 32)):
        cv = ops.Conv(c, c, k, stride=1, pad=k // 2)
        if not cv.fft_ok(B, H, W, backward=True, train=True):
            print("layer %d k%d: no frequency-domain plan" % (c, k))
            continue
        x = torch.randn(B, H, W, c, device=dev, generator=g)
        w = torch.randn(k * k, c, c, device=dev, generator=g) * 0.02
        dy = torch.randn(B, H, W, c, device=dev, generator=g)
        layers.append((cv, x, w, dy, (H, W)))
    ref, bad = {}, {}
    for it in range(n):
        for li, (cv, x, w, dy, hw) in enumerate(layers):
            y, st, xf = cv.fft_fwd(x, w, stats=True, spectrum=True, train=True)
            dw = torch.empty_like(w)
            dx = cv.fft_bwd(dy, w, hw, xf=xf, dw_tap=dw, need_dx=True, train=True)
            for name, t in (("y", y), ("stats", st), ("spectrum", xf), ("dx", dx), ("dw", dw)):
                key = (li, name)
                if key not in ref:
                    ref[key] = t.clone()
                    bad[key] = 0
                elif not torch.equal(t, ref[key]):
                    bad[key] += 1
                    if bad[key] <= 2:
                        a, b = (t, ref[key]) if t.dtype != torch.uint8 else (t.view(torch.float32), ref[key].view(torch.float32))
                        d = (a.double() - b.double()).abs()
                        nz = (d > 0).nonzero()
                        print("  iter %d layer %d %s %s: %d elements differ, max %.3e; index ranges %s" % (
                            it, li, name, tuple(a.shape), len(nz), float(d.max()),
                            [(int(nz[:, q].min()), int(nz[:, q].max()), len(nz[:, q].unique())) for q in range(nz.shape[1])]), flush=True)
    print("victim: repeats that differ of %d: %s" % (n, {"L%d.%s" % k: v for k, v in bad.items() if v}), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "victim":
        victim(int(sys.argv[2]))
        sys.exit(0)
    kind, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 400
    ag = None
    if kind != "none":
        env = dict(os.environ)
        if os.environ.get("GDN_NEIGHBOUR_LIB"):            # the neighbour alone loads another build of the library
            env["GDN_HIP_LIB"] = os.environ["GDN_NEIGHBOUR_LIB"]
        ag = subprocess.Popen([sys.executable, str(ROOT / "tests/diag/dp_solo.py"), "aggressor", kind], stdout=subprocess.PIPE,
                              stderr=subprocess.DEVNULL, text=True, env=env)
        ag.stdout.readline()
    try:
        subprocess.run([sys.executable, __file__, "victim", str(n)], stderr=subprocess.DEVNULL, timeout=900)
    finally:
        if ag:
            print("neighbour %s" % ("still looping" if ag.poll() is None else "HAD EXITED (rc %s): the victim ran alone" % ag.returncode), flush=True)
            ag.kill()
            ag.wait()
