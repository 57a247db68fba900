"""GPU box: are two INDEPENDENT training processes on one GPU bitwise reproducible when they run at the same time?
Each process trains the tiny DtoD model alone (no process group) for a few steps on its own data; the pair is run
concurrently several times and once one after the other; final weights and losses are compared.  Separates 'two
processes share the GPU' from 'gloo moves the gradients' when test_data_parallel_two_ranks is flaky."""
import hashlib, os, subprocess, sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[2]


def child(r, steps):
    if os.environ.get("GDN_X3_R%d" % r) is not None:          # per-process switch: who needs the bf16 x 3 kernels, victim or neighbour?
        os.environ["GDN_X3"] = os.environ["GDN_X3_R%d" % r]
    sys.path[:0] = [str(ROOT), str(ROOT / "gdn-pytorch_amd")]
    import contextlib, io
    import torch
    from oracle import gdn_oracle as O
    import gdn_amd.AE_model_unet as M
    from gdn_amd import utils as U
    from gdn_amd.optim import Adam
    dev = torch.device("cuda:0")
    names, sums, kept = [], [], {}
    keep_fft = os.environ.get("GDN_SOLO_KEEP") == "fft"
    keep_idx = set() if keep_fft else {int(v) for v in os.environ.get("GDN_SOLO_KEEP", "").split(",") if v}
    if os.environ.get("GDN_SOLO_TRACE"):
        # device-side checksums of every convolution call's results, no synchronisation: read back after the last step
        from gdn_amd import ops

        def chk(label, t):
            if t is None or not torch.is_tensor(t) or not t.is_cuda or t.numel() == 0:
                return
            if t.dtype not in (torch.uint8, torch.float32, torch.int32):
                return
            v = t if t.dtype == torch.uint8 else t.view(torch.int32)
            if len(names) in keep_idx or (keep_fft and label.startswith("fft_") and (label.endswith(" out0") or label.endswith(" dw_tap"))):
                kept[len(names)] = (label, t.detach().clone())
            names.append(label)
            sums.append(v.sum(dtype=torch.int64))

        def record(tag, a, k, r):
            for i, t in enumerate(a):
                chk(tag + " arg%d" % i, t)
            for kk, t in sorted(k.items(), key=lambda e: (e[0] == 'dw_tap', e[0])):     # results last
                if isinstance(t, tuple):
                    for j, tt in enumerate(t):
                        chk(tag + " %s.%d" % (kk, j), tt)
                else:
                    chk(tag + " " + kk, t)
            for i, t in enumerate(r if isinstance(r, tuple) else (r,)):
                chk(tag + " out%d" % i, t)

        def wrap_method(name):
            fn = getattr(ops.Conv, name)

            def inner(self, *a, **k):
                r = fn(self, *a, **k)
                record("%s[%d>%d k%d s%d] #%d" % (name, self.cin, self.cout, self.k, self.stride, len(names)), a, k, r)
                return r
            setattr(ops.Conv, name, inner)

        def wrap_fn(name):
            fn = getattr(ops, name)

            def inner(*a, **k):
                r = fn(*a, **k)
                record("%s #%d" % (name, len(names)), a, k, r)
                return r
            setattr(ops, name, inner)
        for nm in ("fwd", "dgrad", "wgrad", "fft_fwd", "fft_bwd", "wino_fwd", "wino_bwd", "wino2_fwd", "wino2_bwd"):
            wrap_method(nm)
        for nm in ("conv_c1_fwd", "conv_c1_wgrad", "bn_finalize_train", "bn_apply", "bn_bwd", "bn_bwd_coeffs", "bn_eval_bwd",
                   "upsample2x", "upsample2x_bwd", "nchw_to_nhwc", "nhwc_to_nchw", "add", "add_pitched", "copy_rows", "scale_dev",
                   "tanh_bwd", "berhu_masked", "sobel_l1", "transpose_taps", "adam_step", "adam_step_dev"):
            if hasattr(ops, nm):
                wrap_fn(nm)
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        m = M.AutoEncoder_DtoD(input_dim=1, height=32, width=64).to(dev).train()
    m(O.synthetic_batch(2, 32, 64, seed=100 + r)[0].to(dev), istrain=False)
    opt = Adam(m.parameters(), 2e-4, [0.9, 0.999], eps=1e-08, weight_decay=5e-4)
    losses = []
    for s in range(steps):
        depth, _, sparse = [t.to(dev) for t in O.synthetic_batch(2, 32, 64, seed=10 * s + r)]
        out = m(depth, istrain=False)
        loss, _, _ = U.dtod_loss(out, depth, sparse)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append("%.9f" % float(loss.detach()))
    w = m._gdn_param_arena.data
    if kept:
        torch.save({i: (n, t.cpu()) for i, (n, t) in kept.items()}, os.environ["GDN_SOLO_KEEP_FILE"] + ".%d" % r)
    if sums:
        vals = torch.stack(sums).cpu().tolist()
        for i, (n, v) in enumerate(zip(names, vals)):
            print("T%d %d %s = %d" % (r, i, n, v))
    print("R%d %s %s" % (r, hashlib.sha256(w.cpu().numpy().tobytes()).hexdigest()[:12], " ".join(losses)), flush=True)


def aggressor(kind):
    """A neighbour process that only loops over one kind of kernel until it is killed."""
    sys.path[:0] = [str(ROOT), str(ROOT / "gdn-pytorch_amd")]
    import torch
    from gdn_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(7)
    bins, M, N, K = 16, 16, 512, 512
    if kind.startswith("big"):
        bins, M, N, K = 16, 2080, 512, 512
    A = torch.randn(bins, M, K, device=dev, generator=g)
    B = torch.randn(bins, N, K, device=dev, generator=g) * 0.05
    D = torch.randn(bins, M, N, device=dev, generator=g)
    Bp = ops.gemm_x3_pack(B)
    C = torch.empty(bins, M, N, device=dev)
    Ab, Bb = A.bfloat16(), B.bfloat16()
    if kind.startswith("huge"):          # dense vendor GEMMs: the matrix pipes busy most of the time
        Ab = torch.randn(8192, 8192, device=dev, generator=g).to(torch.bfloat16 if kind.endswith("bf16") else torch.float32)
        Bb = torch.randn(8192, 8192, device=dev, generator=g).to(Ab.dtype)
    mf = None
    if kind.startswith("mfma"):          # tests/diag/mfma_neighbour.hip: a loop of matrix-core instructions and nothing else
        import ctypes
        mf = ctypes.CDLL(str(ROOT / "tests/diag/_build/libmfma_neighbour.so"))
        mf.mfma_loop.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        sink = torch.zeros(4, device=dev)
    print("aggressor %s running" % kind, flush=True)
    while True:
        for _ in range(200):
            if mf is not None:
                mf.mfma_loop(int(kind[4:]), 1024, 200, sink.data_ptr(), ops.stream())
            elif kind.startswith("huge"):
                torch.mm(Ab, Bb)
            elif kind.endswith("nt"):
                ops.gemm_x3_nt(A, Bp, N, out=C)
            elif kind.endswith("tn"):
                ops.gemm_x3_tn(D, A, 1)
            elif kind.endswith("pack"):
                ops.gemm_x3_pack(B)
            elif kind.endswith("bf16"):
                torch.bmm(Ab, Bb.transpose(1, 2))
            elif kind.endswith("fp32"):
                torch.bmm(A, B.transpose(1, 2))
            else:
                C.mul_(1.0001)
        torch.cuda.synchronize()


last_traces = None


def pair(concurrent, steps):
    cmd = lambda r: [sys.executable, __file__, "child", str(r), str(steps)]
    os.environ["GDN_SOLO_KEEP_FILE"] = "/tmp/solo_keep_%s" % ("c" if concurrent else "s")
    outs = []
    if concurrent and os.environ.get("GDN_SOLO_AGGRESSOR"):
        env = dict(os.environ)
        if os.environ.get("GDN_SOLO_AGGRESSOR_LIB"):          # the neighbour alone loads another build of the library
            env["GDN_HIP_LIB"] = os.environ["GDN_SOLO_AGGRESSOR_LIB"]
        ag = subprocess.Popen([sys.executable, __file__, "aggressor", os.environ["GDN_SOLO_AGGRESSOR"]], stdout=subprocess.PIPE,
                              stderr=subprocess.DEVNULL, text=True, env=env)
        ag.stdout.readline()                       # it is looping
        try:
            o = subprocess.run(cmd(0), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=600).stdout
        finally:
            ag.kill()
            ag.wait()
        outs = [o, "R1 (aggressor)"]
    elif concurrent:
        ps = [subprocess.Popen(cmd(r), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for r in (0, 1)]
        outs = [p.communicate(timeout=600)[0] for p in ps]
    else:
        outs = [subprocess.run(cmd(r), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=600).stdout for r in (0, 1)]
    global last_traces
    last_traces = [[l for l in o.splitlines() if l.startswith("T")] for o in outs]
    return [[l for l in o.splitlines() if l.startswith("R")][-1] for o in outs]


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "aggressor":
        aggressor(sys.argv[2])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(int(sys.argv[2]), int(sys.argv[3]))
        sys.exit(0)
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    ref = pair(False, steps)
    ref_traces = last_traces
    print("sequential:", ref, flush=True)
    bad = 0
    for i in range(reps):
        got = pair(True, steps)
        same = got == ref or (os.environ.get("GDN_SOLO_AGGRESSOR") and got[0] == ref[0])
        bad += not same
        if not same and os.environ.get("GDN_SOLO_KEEP"):
            import torch
            for r in (0, 1):
                a, b = torch.load("/tmp/solo_keep_s.%d" % r), torch.load("/tmp/solo_keep_c.%d" % r)
                for idx in sorted(a):
                    (n, ta), (_, tb) = a[idx], b[idx]
                    if not torch.equal(ta, tb):
                        d = (ta.double() - tb.double()).abs()
                        nz = (d > 0).nonzero()
                        print("   process %d record %d %s shape %s: %d of %d elements differ, max |diff| %.3e (max |value| %.3e)" % (
                            r, idx, n, tuple(ta.shape), len(nz), ta.numel(), float(d.max()), float(ta.abs().max())))
                        for dim in range(ta.dim()):
                            u = nz[:, dim].unique()
                            print("      dim %d: %d distinct indices, %s%s" % (dim, len(u), u[:24].tolist(), " ..." if len(u) > 24 else ""))
                        break
        if not same and ref_traces[0]:
            for r in (0, 1):
                for a, b in zip(ref_traces[r], last_traces[r]):
                    if a != b:
                        print("   first differing result of process %d (of %d recorded):\n      alone:      %s\n      concurrent: %s" % (r, len(ref_traces[r]), a, b))
                        break
        print("concurrent %d: %s" % (i, "same" if same else "DIFFERS " + repr(got)), flush=True)
    print("x3=%s: %d of %d concurrent pairs differ from the sequential pair" % (os.environ.get("GDN_X3", "1"), bad, reps))
