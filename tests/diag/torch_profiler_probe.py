import sys, pathlib, time
R = pathlib.Path("/root/repo") if pathlib.Path("/root/repo/bench.py").exists() else pathlib.Path(".")
sys.path.insert(0, str(R)); sys.path.insert(0, str(R / "gdn-pytorch_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
dev = torch.device("cuda:0")
from oracle import gdn_oracle as O
depth, rgb, sparse = [t.to(dev) for t in O.synthetic_batch(4, 128, 416, seed=0)]
step, _ = bench.make_train_step("DtoD", "fp32", dev, (depth, rgb, sparse))
for _ in range(3): step()
torch.cuda.synchronize()
t0=time.time()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(2): step()
    torch.cuda.synchronize()
print("profiled in", time.time()-t0)
ev = prof.key_averages()
rows = sorted(((e.key, e.device_time_total if hasattr(e,'device_time_total') else e.cuda_time_total, e.count) for e in ev), key=lambda r: -r[1])
for r in rows[:12]: print(r)
