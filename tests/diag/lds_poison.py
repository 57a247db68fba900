"""GPU box: does any kernel of the library read LDS it never wrote?  Before every library call the whole LDS of every CU
is filled with a pattern (tests/diag/lds_fill.hip; 0xFFFFFFFF = NaN, 0x7F000000 = 1.7e38); a few tiny training steps must
give the bits of the clean run.  Every call's tensors are checksummed on the device, so the first call whose RESULT
changes is named.  Build the helper first:
    hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tests/diag/_build/liblds_fill.so tests/diag/lds_fill.hip"""
import ctypes, hashlib, os, subprocess, sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[2]


def child(pattern, steps, H, W, B):
    sys.path[:0] = [str(ROOT), str(ROOT / "gdn-pytorch_amd")]
    import contextlib, io
    import torch
    from oracle import gdn_oracle as O
    import gdn_amd.AE_model_unet as M
    from gdn_amd import utils as U
    from gdn_amd import ops
    from gdn_amd.optim import Adam
    dev = torch.device("cuda:0")
    fill = None
    if pattern >= 0:
        lf = ctypes.CDLL(str(ROOT / "tests/diag/_build/liblds_fill.so"))
        lf.lds_fill.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
        sink = torch.zeros(4, dtype=torch.int32, device=dev)

        def fill():
            rc = lf.lds_fill(pattern, sink.data_ptr(), ops.stream())
            assert rc == 0, rc
    names, sums = [], []

    def chk(label, t):
        if t is None or not torch.is_tensor(t) or not t.is_cuda or t.numel() == 0 or t.dtype not in (torch.uint8, torch.float32, torch.int32):
            return
        names.append(label)
        sums.append((t if t.dtype == torch.uint8 else t.view(torch.int32)).sum(dtype=torch.int64))

    def record(tag, a, k, r):
        for i, t in enumerate(a):
            chk(tag + " arg%d" % i, t)
        for kk, t in sorted(k.items(), key=lambda e: (e[0] == "dw_tap", e[0])):
            for j, tt in enumerate(t if isinstance(t, tuple) else (t,)):
                chk(tag + " %s.%d" % (kk, j), tt)
        for i, t in enumerate(r if isinstance(r, tuple) else (r,)):
            chk(tag + " out%d" % i, t)

    def wrap(owner, name, label):
        fn = getattr(owner, name)

        def inner(*a, **k):
            if fill:
                fill()
            r = fn(*a, **k)
            s = a[0] if owner is ops.Conv else None
            tag = ("%s[%d>%d k%d s%d]" % (name, s.cin, s.cout, s.k, s.stride)) if s is not None else name
            record("%s #%d" % (tag, len(names)), a[1:] if s is not None else a, k, r)
            return r
        setattr(owner, name, inner)
    for nm in ("fwd", "dgrad", "wgrad", "fft_fwd", "fft_bwd", "wino_fwd", "wino_bwd", "wino2_fwd", "wino2_bwd"):
        wrap(ops.Conv, nm, nm)
    for nm in ("conv_c1_fwd", "conv_c1_wgrad", "bn_finalize_train", "bn_apply", "bn_bwd", "bn_bwd_coeffs", "bn_eval_bwd",
               "upsample2x", "upsample2x_bwd", "nchw_to_nhwc", "nhwc_to_nchw", "add", "add_pitched", "copy_rows", "scale_dev",
               "tanh_bwd", "berhu_masked", "sobel_l1", "transpose_taps", "adam_step", "adam_step_dev"):
        if hasattr(ops, nm):
            wrap(ops, nm, nm)
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        m = M.AutoEncoder_DtoD(input_dim=1, height=H, width=W).to(dev).train()
    m(O.synthetic_batch(B, H, W, seed=100)[0].to(dev), istrain=False)
    opt = Adam(m.parameters(), 2e-4, [0.9, 0.999], eps=1e-08, weight_decay=5e-4)
    losses = []
    for s in range(steps):
        depth, _, sparse = [t.to(dev) for t in O.synthetic_batch(B, H, W, seed=10 * s)]
        out = m(depth, istrain=False)
        loss, _, _ = U.dtod_loss(out, depth, sparse)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append("%.9f" % float(loss.detach()))
    vals = torch.stack(sums).cpu().tolist()
    for i, (n, v) in enumerate(zip(names, vals)):
        print("T %d %s = %d" % (i, n, v))
    w = m._gdn_param_arena.data
    print("R %s %s" % (hashlib.sha256(w.cpu().numpy().tobytes()).hexdigest()[:12], " ".join(losses)), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(int(sys.argv[2]), *[int(v) for v in sys.argv[3:7]])
        sys.exit(0)
    H, W, B = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (32, 64, 2)
    steps = 2
    runs = {}
    for pattern in (-1, 0xFFFFFFFF, 0x7F000000, 0):
        o = subprocess.run([sys.executable, __file__, "child", str(pattern), str(steps), str(H), str(W), str(B)],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1200)
        res = [l for l in o.stdout.splitlines() if l.startswith("R ")]
        runs[pattern] = ([l for l in o.stdout.splitlines() if l.startswith("T ")], res[-1] if res else "FAILED " + o.stderr[-600:])
        print("LDS fill %10s: %s" % ("none" if pattern < 0 else hex(pattern), runs[pattern][1]), flush=True)
        if pattern >= 0:
            shown = 0
            for a, b in zip(runs[-1][0], runs[pattern][0]):
                if a != b:
                    print("   differs (of %d records):\n      clean:    %s\n      poisoned: %s" % (len(runs[-1][0]), a, b))
                    shown += 1
                    if shown >= 3:
                        break
    print("x3=%s %dx%d B=%d: %s" % (os.environ.get("GDN_X3", "1"), H, W, B, "all runs agree" if len({r[1] for r in runs.values()}) == 1
                                     else "RESULTS DEPEND ON LDS CONTENTS LEFT BY OTHER WORKGROUPS"))
