import os, sys, pathlib, tempfile
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "gdn-pytorch_amd"), str(ROOT / "tests")]
import torch
import torch.multiprocessing as mp
from oracle import gdn_oracle as O


def main():
    import gdn_amd.AE_model_unet as M
    from gdn_amd import utils as U
    from gdn_amd.optim import Adam
    import dp_worker
    gpu = torch.device("cuda:0")
    steps = 3
    tmp = tempfile.mkdtemp()
    if os.environ.get("GDN_DP_TRACE"):
        os.environ["GDN_DP_DEBUG"] = "1"
    import hashlib
    hsh = lambda t: hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:12]
    overlap = False
    mp.spawn(dp_worker.run, args=(2, 29655, steps, tmp, overlap, False), nprocs=2, join=True)
    r0, r1 = torch.load(tmp + "/rank0.pt"), torch.load(tmp + "/rank1.pt")
    torch.manual_seed(0)
    A = M.AutoEncoder_DtoD(input_dim=1, height=32, width=64).to(gpu).train()
    torch.manual_seed(123)
    Bm = M.AutoEncoder_DtoD(input_dim=1, height=32, width=64).to(gpu).train()
    for r, m in ((0, A), (1, Bm)):
        m(O.synthetic_batch(2, 32, 64, seed=100 + r)[0].to(gpu), istrain=False)
    with torch.no_grad():
        Bm._gdn_param_arena.data.copy_(A._gdn_param_arena.data)
        for bb, ba in zip(Bm.buffers(), A.buffers()):
            bb.copy_(ba)
    opt = Adam(A.parameters(), 2e-4, [0.9, 0.999], eps=1e-08, weight_decay=5e-4)
    opt.grad_scale = 0.5
    for s in range(steps):
        batches = [[t.to(gpu) for t in O.synthetic_batch(2, 32, 64, seed=10 * s + r)] for r in (0, 1)]
        outs = [m(b[0], istrain=False) for m, b in zip((A, Bm), batches)]
        for r, (m, b, out) in enumerate(zip((A, Bm), batches, outs)):
            depth, _, sparse = b
            loss, _, _ = U.dtod_loss(out, depth, sparse)
            m.zero_grad()
            loss.backward()
            ref = (r0 if r == 0 else r1)["losses"][s]
            print("step %d rank %d: emulation loss %.9f worker %.9f %s" % (s, r, float(loss), ref, "" if float(loss) == ref else "  <-- differs"))
        gl = [hsh(A._gdn_param_arena.grad), hsh(Bm._gdn_param_arena.grad)]
        A._gdn_param_arena.grad.add_(Bm._gdn_param_arena.grad)
        gr = hsh(A._gdn_param_arena.grad)
        opt.step()
        wh = hsh(A._gdn_param_arena.data)
        for r, rr in ((0, r0), (1, r1)):
            if not rr["trace"]:
                continue
            t = {k: v for k, ss, v in rr["trace"] if ss == s}
            print("   step %d rank %d: local grad %s (emu %s) | reduced %s (emu %s) | weights %s (emu %s)" % (
                s, r, t["grad_local"], gl[r], t["grad_reduced"], gr, t["weights"], wh))
        with torch.no_grad():
            Bm._gdn_param_arena.data.copy_(A._gdn_param_arena.data)
    nbad = 0
    for k, v in A.state_dict().items():
        if not torch.equal(v.cpu(), r0["sd"][k]):
            nbad += 1
            if nbad <= 8:
                d = (v.cpu().double() - r0["sd"][k].double()).abs().max()
                print("  rank-0 state differs at %-40s max diff %.3e" % (k, float(d)))
    print("x3=%s: %d rank-0 state tensors differ" % (os.environ.get("GDN_X3", "1"), nbad))


if __name__ == "__main__":
    main()
