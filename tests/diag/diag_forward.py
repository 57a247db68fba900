#!/usr/bin/env python3
"""Diagnostic (GPU box): per-feature error of the HIP forward vs the CPU oracle in fp32 and fp64."""
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "gdn-pytorch_amd"))
import torch
from oracle import gdn_oracle as O
import gdn_amd.AE_model_unet as M

torch.set_num_threads(16)
dev = torch.device("cuda:0")
depth, rgb, sparse = O.synthetic_batch(2, 128, 416, seed=0)
for name in ("AutoEncoder_DtoD", "AutoEncoder_2"):
    x = depth if name == "AutoEncoder_DtoD" else rgb
    sd = O.init_state_dict(name, seed=0)
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    with torch.no_grad():
        f32 = O.FORWARD[name]({k: v.clone() for k, v in sd.items()}, x, istrain=True, training=True)
        f64 = O.FORWARD[name](sd64, x.double(), istrain=True, training=True)
    m = getattr(M, name)()
    m.load_state_dict(sd)
    m = m.to(dev).train()
    with torch.no_grad():
        fh = m(x.to(dev), istrain=True)
    print(name)
    for i in range(8):
        h = fh[i].cpu().double(); a = f32[i].double(); b = f64[i]
        sc = float(b.abs().max())
        print("  f%d scale %.3e  |hip-cpu32| %.3e  |hip-f64| %.3e  |cpu32-f64| %.3e   rms(hip-f64) %.3e rms(cpu32-f64) %.3e" % (
            i, sc, float((h - a).abs().max()), float((h - b).abs().max()), float((a - b).abs().max()),
            float((h - b).pow(2).mean().sqrt()), float((a - b).pow(2).mean().sqrt())))
