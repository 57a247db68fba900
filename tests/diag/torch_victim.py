"""GPU box: is a process that runs only vendor / torch kernels disturbed by a neighbour looping over gemm_x3_nt?"""
import os, subprocess, sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[2]


def victim(n):
    import torch
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(64, 1 << 16, device=dev, generator=g)
    w = torch.randn(256, 256, device=dev, generator=g)
    a = torch.randn(2048, 256, device=dev, generator=g)
    img = torch.randn(8, 64, 64, 64, device=dev, generator=g)
    k = torch.randn(64, 64, 5, 5, device=dev, generator=g) * 0.05
    ref, bad = {}, {}
    for it in range(n):
        outs = {"elementwise": torch.tanh(x * 1.0001 + 0.5) * x, "softmax": torch.softmax(x, dim=1), "sum": x.sum(dim=1),
                "mm_fp32": a @ w, "conv_fp32": torch.nn.functional.conv2d(img, k, padding=2),
                "fft": torch.view_as_real(torch.fft.rfft2(img))}
        for name, t in outs.items():
            if name not in ref:
                ref[name], bad[name] = t.clone(), 0
            elif not torch.equal(t, ref[name]):
                bad[name] += 1
                if bad[name] <= 2:
                    d = (t - ref[name]).abs()
                    print("  iter %d %s: %d elements differ, max %.3e" % (it, name, int((d > 0).sum()), float(d.max())), flush=True)
    print("torch-only victim: repeats that differ of %d: %s" % (n, {k: v for k, v in bad.items() if v}), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "victim":
        victim(int(sys.argv[2]))
        sys.exit(0)
    kind, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 300
    ag = None
    if kind != "none":
        env = dict(os.environ)
        if os.environ.get("GDN_NEIGHBOUR_LIB"):
            env["GDN_HIP_LIB"] = os.environ["GDN_NEIGHBOUR_LIB"]
        ag = subprocess.Popen([sys.executable, str(ROOT / "tests/diag/dp_solo.py"), "aggressor", kind], stdout=subprocess.PIPE,
                              stderr=subprocess.DEVNULL, text=True, env=env)
        ag.stdout.readline()
    try:
        subprocess.run([sys.executable, __file__, "victim", str(n)], stderr=subprocess.DEVNULL, timeout=900)
    finally:
        if ag:
            print("neighbour %s" % ("still looping" if ag.poll() is None else "HAD EXITED"), flush=True)
            ag.kill()
            ag.wait()
