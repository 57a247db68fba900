"""Worker of tests/test_hip_model.py::test_data_parallel_two_ranks (spawned, one process per rank).  On a box with one GPU both
ranks sit on cuda:0 with the gloo backend (RCCL refuses two ranks on one device, gloo moves the same buckets through the host);
with two or more GPUs visible it is the deployment layout: one rank per device, RCCL ("nccl"), the bf16 x 3 GEMMs left on."""
import os
import sys

import torch


def run(rank, world, port, steps, out_dir, overlap, global_berhu=False):
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    for p in (root, os.path.join(root, "gdn-pytorch_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    multi = torch.cuda.device_count() >= world           # (counting devices does not initialise the GPU)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank) if multi else "0",
                      WORLD_SIZE=str(world), GDN_OVERLAP_ALLREDUCE="1" if overlap else "0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import gdn_amd.AE_model_unet as M
    from gdn_amd import distributed as D
    from gdn_amd import ops
    from gdn_amd import utils as U
    from gdn_amd.optim import Adam
    from oracle import gdn_oracle as O
    D.init(backend="nccl" if multi else "gloo")
    U.GLOBAL_BERHU = bool(global_berhu)
    dev = torch.device("cuda", rank if multi else 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(0 if rank == 0 else 123)          # rank 1 starts from different weights: the broadcast must fix it
    model = M.AutoEncoder_DtoD(input_dim=1, height=32, width=64).to(dev).train()
    x0 = O.synthetic_batch(2, 32, 64, seed=100 + rank)[0].to(dev)
    model(x0, istrain=False)                             # builds the arena
    D.broadcast_parameters(model, src=0)
    opt = Adam(model.parameters(), 2e-4, [0.9, 0.999], eps=1e-08, weight_decay=5e-4)
    losses, trace = [], []             # (trace: per-step hashes for tests/diag/dp_debug.py, GDN_DP_DEBUG=1)
    import hashlib
    hsh = lambda t: hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:12]
    for s in range(steps):
        depth, _, sparse = [t.to(dev) for t in O.synthetic_batch(2, 32, 64, seed=10 * s + rank)]
        out = model(depth, istrain=False)
        loss, _, _ = U.dtod_loss(out, depth, sparse)
        opt.zero_grad()
        loss.backward()
        if os.environ.get("GDN_DP_DEBUG"):
            trace.append(("grad_local", s, hsh(model._gdn_param_arena.grad)))
        D.sync_gradients(model, opt)
        if os.environ.get("GDN_DP_DEBUG"):
            trace.append(("grad_reduced", s, hsh(model._gdn_param_arena.grad)))
        opt.step()
        if os.environ.get("GDN_DP_DEBUG"):
            trace.append(("weights", s, hsh(model._gdn_param_arena.data)))
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    torch.save({"sd": {k: v.cpu() for k, v in model.state_dict().items()}, "losses": losses, "trace": trace,
                "reducer": getattr(model, "_gdn_reducer", None) is not None,
                "x3": "1" if ops.x3_enabled() else "0", "shared": D.SHARED_GPU_RANKS, "backend": str(torch.distributed.get_backend())},
               os.path.join(out_dir, "rank%d.pt" % rank))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
