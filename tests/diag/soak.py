#!/usr/bin/env python3
"""GPU box: 300 training steps per mode on one synthetic batch -- ms/step, loss every 50 steps, allocator high-water marks
before and after (a leak or a growing workspace cache would show as growth)."""
import pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "gdn-pytorch_amd"))
import torch
import bench
from gdn_amd.synthetic import synthetic_batch
dev = torch.device("cuda:0")
depth, rgb, sparse = synthetic_batch(20, 128, 416, seed=0, device=dev)
for mode, dtype in (("DtoD", "fp32"), ("RtoD", "fp32"), ("RtoD", "bf16")):
    torch.manual_seed(0)
    step, _ = bench.make_train_step(mode, dtype, dev, (depth, rgb, sparse))
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    m0 = torch.cuda.memory_allocated(), torch.cuda.memory_reserved()
    t0 = time.perf_counter()
    losses = []
    for i in range(300):
        l = step()
        if i % 50 == 49:
            losses.append(float(l))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    m1 = torch.cuda.memory_allocated(), torch.cuda.memory_reserved()
    print(mode, dtype, "300 steps %.1f ms/step" % (dt / 300 * 1e3), "losses", ["%.4f" % v for v in losses],
          "allocated MB %d -> %d" % (m0[0] >> 20, m1[0] >> 20), "reserved MB %d -> %d" % (m0[1] >> 20, m1[1] >> 20), flush=True)
    del step
    torch.cuda.empty_cache()
