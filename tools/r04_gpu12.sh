cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_bf16.py -x -q -k "head or mini" 2>&1 | tail -8 > gpurun_out/r04_t12.log
python - >> gpurun_out/r04_t12.log 2>&1 <<'PY'
import sys; sys.path.insert(0, "."); sys.path.insert(0, "gdn-pytorch_amd")
import torch
from gdn_amd import ops
dev = torch.device("cuda:0")
for B in (20, 40):
    op = ops.Conv(64, 1, 9, 1, 4)
    x = torch.randn(B, 128, 416, 64, device=dev).bfloat16(); w = torch.randn(81, 1, 64, device=dev) * 0.02
    for _ in range(3): op.fwd(x, w, act=ops.ACT_TANH)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): op.fwd(x, w, act=ops.ACT_TANH)
    e1.record(); torch.cuda.synchronize()
    print("bf16 head B=%d: %.1f us" % (B, e0.elapsed_time(e1) * 100))
PY
cat gpurun_out/r04_t12.log
