"""GPU box: one tiny DtoD training step; prints hashes of the depth map, the loss and the gradient arena.  Run it in several
fresh processes: differing hashes = a kernel reads memory nobody wrote (same-process repeats see the same stale bytes)."""
import hashlib, os, sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "gdn-pytorch_amd")]
import torch
from oracle import gdn_oracle as O
import gdn_amd.AE_model_unet as M
from gdn_amd import utils as U
dev = torch.device("cuda:0")
H, W, B = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (32, 64, 2)
# dirty the allocator with process-dependent garbage first
junk = torch.empty(256 * 1024 * 1024, dtype=torch.uint8, device=dev).random_(generator=torch.Generator(device=dev).manual_seed(os.getpid()))
del junk
torch.manual_seed(0)
import contextlib, io
with contextlib.redirect_stdout(io.StringIO()):
    m = M.AutoEncoder_DtoD(input_dim=1, height=H, width=W).to(dev).train()
depth, rgb, sparse = [t.to(dev) for t in O.synthetic_batch(B, H, W, seed=3)]
h = lambda t: hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:12]
for step in range(2):
    out = m(depth, istrain=False)
    loss, _, _ = U.dtod_loss(out, depth, sparse)
    m.zero_grad()
    loss.backward()
    print("step", step, "out", h(out), "loss %.9f" % float(loss), "grad", h(m._gdn_param_arena.grad),
          "x3=%s" % os.environ.get("GDN_X3", "1"), flush=True)
