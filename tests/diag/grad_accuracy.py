#!/usr/bin/env python3
"""GPU box: parameter-gradient accuracy of the HIP path against the fp64 truth, next to the fp32 CPU oracle's own distance
from it -- with the bf16 x 3 GEMMs (default) and with GDN_X3=0 (fp32 MFMA GEMMs).  Every backward starts from the SAME
dL/dout (the fp32 oracle's), so the sign functions of the L1-type losses play no part.
    rel(k) = |g_k - g64_k| / (|g64_k| + 1e-3 * typical)        typical = median parameter-gradient norm"""
import os, pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "gdn-pytorch_amd"))
import numpy as np
import torch
from oracle import gdn_oracle as O
import gdn_amd.AE_model_unet as M
from gdn_amd import ops
dev = torch.device("cuda:0")
H, W = 128, 416
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))


def oracle_grads(sd, depth, dout, dtype):
    keys = O.trainable_keys(sd)
    cast = (lambda v: v.to(dtype) if v.is_floating_point() else v.clone())
    work = {k: cast(v) for k, v in sd.items()}
    leaves = {k: work[k].detach().requires_grad_(True) for k in keys}
    work.update(leaves)
    out = O.forward_dtod(work, depth.to(dtype), istrain=False, training=True)
    out.backward(dout.to(dtype))
    return out.detach(), {k: leaves[k].grad.double() for k in keys}


for seed in range(2):
    depth, rgb, sparse = O.synthetic_batch(B, H, W, seed=200 + seed)
    sd = O.init_state_dict("AutoEncoder_DtoD", seed=seed)
    dout = O.train_step("DtoD", {k: v.clone() for k, v in sd.items()}, (depth, rgb, sparse), {})["dout"]   # one fixed dL/dout
    out64, g64 = oracle_grads(sd, depth, dout, torch.float64)
    out32, g32 = oracle_grads(sd, depth, dout, torch.float32)
    typical = float(np.median([v.norm().item() for v in g64.values()]))
    rows = {}

    def score(name, out, grads):
        rel = {k: float((grads[k] - g64[k]).norm() / (g64[k].norm() + 1e-3 * typical)) for k in g64}
        worst = max(rel, key=rel.get)
        e = out.double() - out64
        rows[name] = rel
        print("seed %d %-14s depth map max %.2e rms %.2e | grad rel err: median %.2e  p90 %.2e  worst %.2e (%s)" % (
            seed, name, float(e.abs().max()), float(e.pow(2).mean().sqrt()), float(np.median(list(rel.values()))),
            float(np.percentile(list(rel.values()), 90)), rel[worst], worst), flush=True)

    score("oracle fp32", out32, g32)
    for x3 in ("1", "0"):
        ops.set_x3(x3 == "1")
        model = M.AutoEncoder_DtoD(input_dim=1, height=H, width=W)
        model.load_state_dict(sd)
        model = model.to(dev).train()
        out = model(depth.to(dev), istrain=False)
        out.backward(dout.float().to(dev))
        score("hip x3=%s" % x3, out.detach().cpu(), {k: p.grad.detach().cpu().double() for k, p in model.named_parameters()})
    k = "res64_down1.main.4.bias"
    print("   %s: oracle32 %.2e  x3 %.2e  fp32-mfma %.2e   |g64| %.2e typical %.2e" % (
        k, rows["oracle fp32"][k], rows["hip x3=1"][k], rows["hip x3=0"][k], float(g64[k].norm()), typical))
