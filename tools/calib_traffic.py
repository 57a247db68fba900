#!/usr/bin/env python3
"""GPU box: known-byte-count streams for calibrating FETCH_SIZE / WRITE_SIZE (MI355X_MICROARCH.md, HBM section)."""
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "gdn-pytorch_amd"))
import torch
from gdn_amd import ops

dev = torch.device("cuda:0")
n = 256 * 1024 * 1024          # 1 GiB per array: beyond the 256 MiB Infinity Cache
a = torch.randn(n, device=dev)
b = torch.randn(n, device=dev)
torch.cuda.synchronize()
for _ in range(3):
    ops.add(a, b)              # gdn add_kernel: reads 2 GiB, writes 1 GiB, 16 B per lane
torch.cuda.synchronize()
print("done")
