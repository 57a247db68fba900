#!/usr/bin/env python3
"""GPU box: how far the bf16 path drifts from the fp32 HIP path on the full networks (random init, train-mode BN).
Prints loss, depth-map error statistics, correlation and per-parameter gradient errors."""
import copy
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "gdn-pytorch_amd"))
import numpy as np
import torch
import gdn_amd.AE_model_unet as M
from gdn_amd import utils as U
from oracle import gdn_oracle as O

gpu = torch.device("cuda:0")
for (B, H, W) in ((2, 64, 96), (2, 128, 416)):
    depth, rgb, sparse = [t.to(gpu) for t in O.synthetic_batch(B, H, W, seed=4)]
    torch.manual_seed(3)
    ref = M.AutoEncoder_DtoD(input_dim=1, height=H, width=W)
    blk = copy.deepcopy(ref)
    ref, blk = ref.to(gpu).train(), blk.to(gpu).train().compute_dtype("bf16")
    res = []
    for m in (ref, blk):
        outs = m(depth, istrain=True)
        out = outs[7]
        loss, _, _ = U.dtod_loss(out, depth, sparse)
        loss.backward()
        res.append(([o.detach().float() for o in outs], loss.item()))
    print("== %dx%dx%d  loss fp32 %.5f bf16 %.5f" % (B, H, W, res[0][1], res[1][1]))
    for i, (a, b) in enumerate(zip(res[0][0], res[1][0])):
        d = (a - b).abs()
        cc = torch.corrcoef(torch.stack((a.flatten(), b.flatten())))[0, 1].item()
        print("  feature %d %-22s max|ref| %.3f  mean|err| %.4f  max|err| %.3f  relL2 %.4f  corr %.5f" % (
            i, tuple(a.shape), a.abs().max().item(), d.mean().item(), d.max().item(), ((a - b).norm() / a.norm()).item(), cc))
    errs = []
    for (k, p), (_, q) in zip(blk.named_parameters(), ref.named_parameters()):
        e = float((p.grad.double() - q.grad.double()).norm() / (q.grad.double().norm() + 1e-30))
        errs.append((e, k, float(q.grad.norm()), float(p.grad.norm())))
    errs.sort(reverse=True)
    for e, k, n0, n1 in errs[:8]:
        print("  grad relL2 %.3f  %-40s |fp32| %.4e |bf16| %.4e" % (e, k, n0, n1))
    print("  median grad relL2 %.4f" % np.median([e[0] for e in errs]))

# ---- HIP bf16 vs the oracle's bf16 emulation (same rounding points, fp32 arithmetic on the CPU)
print("\n== HIP bf16 vs emulated-bf16 oracle")
for (B, H, W) in ((1, 32, 64), (2, 64, 96)):
    depth, rgb, sparse = O.synthetic_batch(B, H, W, seed=4)
    sd = O.init_state_dict("AutoEncoder_DtoD", seed=3)
    with O.bf16_emulation():
        ref = O.train_step("DtoD", {k: v.clone() for k, v in sd.items()}, (depth, rgb, sparse), {})
        with torch.no_grad():
            feats = O.forward_dtod({k: v.clone() for k, v in sd.items()}, depth, istrain=True, training=True)
    model = M.AutoEncoder_DtoD(input_dim=1, height=H, width=W)
    model.load_state_dict(sd)
    model = model.to(gpu).train().compute_dtype("bf16")
    outs = model(depth.to(gpu), istrain=True)
    loss, _, _ = U.dtod_loss(outs[7], depth.to(gpu), sparse.to(gpu))
    loss.backward()
    print("  %dx%dx%d loss hip %.6f emu %.6f" % (B, H, W, loss.item(), ref["loss"]))
    for i, (a, b) in enumerate(zip(feats, outs)):
        a, b = a.float(), b.detach().float().cpu()
        print("   feature %d relL2 %.5f max|err| %.4f (max|ref| %.3f)" % (i, ((a - b).norm() / a.norm()).item(), (a - b).abs().max().item(), a.abs().max().item()))
    errs = []
    for k, p in model.named_parameters():
        q = ref["grads"][k].double()
        errs.append((float((p.grad.detach().cpu().double() - q).norm() / (q.norm() + 1e-30)), k, float(q.norm())))
    errs.sort(reverse=True)
    for e, k, n0 in errs[:6]:
        print("   grad relL2 %.4f %-36s |ref| %.3e" % (e, k, n0))
    print("   median grad relL2 %.5f" % np.median([e[0] for e in errs]))
