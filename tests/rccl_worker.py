"""Worker of tests/test_hip_model.py::test_rccl_single_rank_reducer_is_bitwise_identity.  A fresh process (the process group
must not leak into pytest) that trains a few DtoD steps either plainly or through the data-parallel machinery with a
1-rank RCCL group (GDN_FORCE_DIST=1): async bucketed all-reduce on RCCL's stream, GradReducer overlap with backward,
work.wait() ordering against the compute stream, grad_scale = 1/world in the fused Adam."""
import os
import sys

import torch


def main(argv):
    out_path, use_dist, steps = argv[0], argv[1] == "1", int(argv[2])
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    for p in (root, os.path.join(root, "gdn-pytorch_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    if use_dist:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.update(GDN_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(port))
        os.environ.pop("GDN_DIST_BACKEND", None)              # "nccl" == RCCL
    import gdn_amd.AE_model_unet as M
    from gdn_amd import distributed as D
    from gdn_amd import utils as U
    from gdn_amd.optim import Adam
    from oracle import gdn_oracle as O
    rank, local_rank, world = D.init()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    info = {"active": D.active(), "backend": torch.distributed.get_backend() if torch.distributed.is_initialized() else None}
    torch.manual_seed(0)
    H, W = 64, 96
    model = M.AutoEncoder_DtoD(input_dim=1, height=H, width=W).to(dev).train()
    D.broadcast_parameters(model)
    opt = Adam(model.parameters(), 2e-4, [0.9, 0.999], eps=1e-08, weight_decay=5e-4)
    losses, fired_early = [], []
    for s in range(steps):
        depth, _, sparse = [t.to(dev) for t in O.synthetic_batch(2, H, W, seed=20 + s)]
        out = model(depth, istrain=False)
        loss, _, _ = U.dtod_loss(out, depth, sparse)
        opt.zero_grad()
        loss.backward()
        red = getattr(model, "_gdn_reducer", None)
        if red is not None and red.active:
            fired_early.append(sum(red.fired))                 # buckets already in flight when backward returned
        D.sync_gradients(model, opt)
        opt.step()
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    info.update(losses=losses, fired_early=fired_early, n_buckets=len(red.buckets) if red is not None else 0,
                sd={k: v.cpu() for k, v in model.state_dict().items()})
    torch.save(info, out_path)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1:])
