"""GPU box: does the NT GEMM ever see OLD contents of a buffer the previous kernel of the stream rewrote?  Per iteration the
weights change a little (as under Adam), are packed into the SAME buffer, and multiplied; all products are kept and
recomputed at the end, one at a time with a device synchronisation around every launch.  `pair` runs two such processes
at once (the condition under which tests/diag/dp_solo.py is not reproducible)."""
import os, subprocess, sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[2]


def child(seed, n):
    sys.path[:0] = [str(ROOT), str(ROOT / "gdn-pytorch_amd")]
    import torch
    from gdn_amd import ops
    from gdn_amd._lib import lib
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(seed)
    bins, M, N, K = 16, 16, 512, 512
    A = torch.randn(bins, M, K, device=dev, generator=g)
    W = torch.randn(bins, N, K, device=dev, generator=g) * 0.05
    delta = torch.randn(bins, N, K, device=dev, generator=g) * 2e-4
    nb = int(lib.gdn_gemm_x3_packed_bytes(bins, N, K))
    Bp = torch.empty(nb, dtype=torch.uint8, device=dev)
    st = ops.stream()
    outs = torch.empty((n, bins, M, N), device=dev)
    filler = torch.randn(64, 1 << 18, device=dev, generator=g)           # 64 MB of other traffic between the iterations
    w = W.clone()
    for it in range(n):
        w.add_(delta)
        lib.gdn_gemm_x3_pack(w.data_ptr(), Bp.data_ptr(), bins, N, K, st)
        ops.gemm_x3_nt(A, Bp, N, out=outs[it])
        filler[it % 64].mul_(1.0001)
    torch.cuda.synchronize()
    w = W.clone()
    bad = 0
    C = torch.empty((bins, M, N), device=dev)
    for it in range(n):
        w.add_(delta)
        torch.cuda.synchronize()
        fresh = torch.empty(nb, dtype=torch.uint8, device=dev)
        lib.gdn_gemm_x3_pack(w.data_ptr(), fresh.data_ptr(), bins, N, K, st)
        torch.cuda.synchronize()
        ops.gemm_x3_nt(A, fresh, N, out=C)
        torch.cuda.synchronize()
        if not torch.equal(C, outs[it]):
            bad += 1
            if bad <= 3:
                d = (C - outs[it]).abs()
                print("  seed %d iter %d: %d elements differ, max %.3e (|C| max %.3e)" % (seed, it, int((d > 0).sum()), float(d.max()), float(C.abs().max())), flush=True)
    print("seed %d (GDN_X3_NT=%s): %d of %d products differ from the synchronised recomputation" % (seed, os.environ.get("GDN_X3_NT", "auto"), bad, n), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "child":
        child(int(sys.argv[2]), int(sys.argv[3]))
    else:
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 600
        ps = [subprocess.Popen([sys.executable, __file__, "child", str(s), str(n)], stderr=subprocess.DEVNULL) for s in (1, 2)]
        for p in ps:
            p.wait(timeout=900)
