"""GPU box: repeat pack -> bf16 x 3 NT GEMM (and the TN GEMM) on the tiny shapes of the 32 x 64 test model many times and count
results that differ from the first.  `pair` runs two of these processes at the same time on different data -- the
condition under which two independent trainers on one GPU were not reproducible (tests/diag/dp_solo.py)."""
import os, subprocess, sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[2]


def child(seed, n):
    sys.path[:0] = [str(ROOT), str(ROOT / "gdn-pytorch_amd")]
    import torch
    from gdn_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(seed)
    cases = []
    for (bins, M, N, K) in ((16, 4, 512, 512), (16, 16, 512, 512), (16, 64, 256, 256), (9, 16, 512, 256)):
        A = torch.randn(bins, M, K, device=dev, generator=g)
        B = torch.randn(bins, N, K, device=dev, generator=g) * 0.05
        D = torch.randn(bins, M, N, device=dev, generator=g)
        cases.append((A, B, D, N))
    ref = {}
    bad = {"nt": 0, "tn": 0, "pack": 0}
    for it in range(n):
        for ci, (A, B, D, N) in enumerate(cases):
            Bp = ops.gemm_x3_pack(B)
            C = ops.gemm_x3_nt(A, Bp, N)
            P = ops.gemm_x3_tn(D, A, 1)
            for name, t in (("nt", C), ("tn", P), ("pack", Bp.view(torch.int32))):
                key = (ci, name)
                if key not in ref:
                    ref[key] = t.clone()
                elif not torch.equal(t, ref[key]):
                    bad[name] += 1
                    if bad[name] <= 2:
                        d = (t.float() - ref[key].float()).abs()
                        print("  seed %d iter %d case %d %s: %d elements differ, max %.3e" % (seed, it, ci, name, int((d > 0).sum()), float(d.max())), flush=True)
    print("seed %d (GDN_X3_NT=%s): repeats that differ: %s of %d x %d" % (seed, os.environ.get("GDN_X3_NT", "auto"), bad, n, len(cases)), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "child":
        child(int(sys.argv[2]), int(sys.argv[3]))
    else:
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
        ps = [subprocess.Popen([sys.executable, __file__, "child", str(s), str(n)], stderr=subprocess.DEVNULL) for s in (1, 2)]
        for p in ps:
            p.wait(timeout=900)
