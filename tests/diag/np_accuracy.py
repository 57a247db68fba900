#!/usr/bin/env python3
"""GPU box: depth-map error of a train-mode DtoD forward against the CPU oracle (max and rms over all pixels) for several
seeds -- run once per tiling (GDN_FFT_NP=32 / default) to tell a real accuracy difference from the spread of a maximum."""
import pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "gdn-pytorch_amd"))
import torch
from oracle import gdn_oracle as O
import gdn_amd.AE_model_unet as M
dev = torch.device("cuda:0")
H, W, B = 128, 416, 4
for seed in range(4):
    depth, rgb, sparse = O.synthetic_batch(B, H, W, seed=100 + seed)
    sd = O.init_state_dict("AutoEncoder_DtoD", seed=seed)
    with torch.no_grad():
        ref = O.forward_dtod({k: v.clone() for k, v in sd.items()}, depth, istrain=False, training=True)
    ref64 = O.forward_dtod({k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}, depth.double(),
                           istrain=False, training=True).detach()
    model = M.AutoEncoder_DtoD(input_dim=1, height=H, width=W)
    model.load_state_dict(sd)
    model = model.to(dev).train()
    x = depth.to(dev).requires_grad_(True)          # (records a tape: the trained-layer plan)
    out = model(x, istrain=False).detach().cpu().double()
    e = out - ref.double()
    e64, o64 = out - ref64, ref.double() - ref64
    print("seed %d: hip-oracle32 max %.2e rms %.2e | hip-fp64 max %.2e rms %.2e | oracle32-fp64 max %.2e rms %.2e" % (
        seed, float(e.abs().max()), float(e.pow(2).mean().sqrt()), float(e64.abs().max()), float(e64.pow(2).mean().sqrt()),
        float(o64.abs().max()), float(o64.pow(2).mean().sqrt())), flush=True)
