"""GPU box: the direct bf16 convolutions of an RtoD bf16 training step, timed call by call inside the step (events, no
synchronisation) and again in isolation on the SAME captured tensors -- why does conv_rowpatch_bf16 take 0.86 ms in the
step and 0.47 ms in tools/tune_conv.py?"""
import os, sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "gdn-pytorch_amd")]
import torch
import bench
from gdn_amd import ops
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
from gdn_amd.synthetic import synthetic_batch
depth, rgb, sparse = synthetic_batch(20, 128, 416, seed=0, device=dev)
step, _ = bench.make_train_step("RtoD", "bf16", dev, (depth, rgb, sparse))
for _ in range(3):
    step()
torch.cuda.synchronize()
recs, keep = [], {}
orig_fwd, orig_dgrad = ops.Conv.fwd, ops.Conv.dgrad


def wrap(fn, kind):
    def inner(self, *a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(self, *a, **k)
        e1.record()
        x = a[0]
        key = (kind, self.cin, self.cout, self.k, self.stride, tuple(x.shape), x.dtype, x.is_contiguous(), x.stride(2), sorted(k.keys()).__repr__())
        recs.append((key, e0, e1))
        if self.k >= 5 and self.stride == 1 and key not in keep and x.dtype == torch.bfloat16:
            keep[key] = (self, [t.clone() if torch.is_tensor(t) else t for t in a],
                         {kk: (vv.clone() if torch.is_tensor(vv) else vv) for kk, vv in k.items() if kk not in ("out", "stats_out")})
        return r
    return inner


ops.Conv.fwd, ops.Conv.dgrad = wrap(orig_fwd, "fwd"), wrap(orig_dgrad, "dgrad")
step()
torch.cuda.synchronize()
ops.Conv.fwd, ops.Conv.dgrad = orig_fwd, orig_dgrad
import collections
agg = collections.defaultdict(list)
for key, e0, e1 in recs:
    agg[key].append(e0.elapsed_time(e1))
print("in-step, large-window stride-1 bf16 layers:")
for key, ts in sorted(agg.items(), key=lambda e: -sum(e[1])):
    if key[3] >= 5 and key[4] == 1 and key[6] == torch.bfloat16:
        print("  %-5s %3d->%-3d k%d in %-22s contiguous=%s pitch=%d kwargs=%s: x%d, %.3f ms each" % (
            key[0], key[1], key[2], key[3], str(key[5]), key[7], key[8], key[9], len(ts), sum(ts) / len(ts)))
print("the same calls replayed alone on the captured tensors (10 back-to-back launches):")
for key, (self, a, k) in keep.items():
    fn = orig_fwd if key[0] == "fwd" else orig_dgrad
    fn(self, *a, **k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn(self, *a, **k)
    e1.record()
    torch.cuda.synchronize()
    print("  %-5s %3d->%-3d k%d in %-22s kwargs=%s: %.3f ms" % (key[0], key[1], key[2], key[3], str(key[5]), key[9], e0.elapsed_time(e1) / 10))
