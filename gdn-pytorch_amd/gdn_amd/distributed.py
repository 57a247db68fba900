"""Data parallelism: one process per GPU, RCCL all-reduce of the gradient arena over xGMI.

Replaces the reference's single-process ``nn.DataParallel`` (GDN_main.py:153,
163,169): no per-forward parameter broadcast, no scatter/gather of activations;
each rank trains its own batch shard with rank-local BatchNorm statistics (what
DataParallel's replicas do) and the only exchange is one bucketed SUM
all-reduce of the flat gradient arena per step, divided by world size inside the
fused Adam kernel.
"""
import os

import torch
import torch.distributed as dist


def env_rank():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init(backend=None):
    """Initialise torch.distributed from the launcher's environment (no-op for world size 1)."""
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" is RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


# xGMI links are point to point (~153 GB/s each); a handful of large buckets keeps
# each ring step bandwidth-bound while letting the tail overlap with Adam setup.
BUCKET_ELEMS = 16 * 1024 * 1024     # 64 MB fp32


def flat_buckets(flat, bucket_elems=BUCKET_ELEMS):
    n = flat.numel()
    return [flat[o:min(n, o + bucket_elems)] for o in range(0, n, bucket_elems)]


def allreduce_flat(flat, bucket_elems=BUCKET_ELEMS, async_op=True):
    """SUM all-reduce a flat buffer in buckets; returns the work handles (already waited if not async)."""
    if world_size() == 1:
        return []
    works = [dist.all_reduce(b, op=dist.ReduceOp.SUM, async_op=True) for b in flat_buckets(flat, bucket_elems)]
    if not async_op:
        for w in works:
            w.wait()
        return []
    return works


def sync_gradients(model, optimizer=None):
    """All-reduce the model's gradient arena; the mean is applied by the optimizer's grad_scale."""
    ws = world_size()
    ar = getattr(model, "_gdn_param_arena", None)
    if ws == 1:
        return
    if ar is None:
        raise RuntimeError("sync_gradients: model has no gradient arena yet (run a forward/backward first)")
    for w in allreduce_flat(ar.grad):
        w.wait()
    if optimizer is not None and hasattr(optimizer, "grad_scale"):
        optimizer.grad_scale = 1.0 / ws
    else:
        ar.grad.mul_(1.0 / ws)


def broadcast_parameters(model, src=0):
    """Make every rank start from rank `src`'s weights and BN buffers (once, at start)."""
    if world_size() == 1:
        return
    ar = getattr(model, "_gdn_param_arena", None)
    if ar is not None:
        dist.broadcast(ar.data, src)
    else:
        for p in model.parameters():
            dist.broadcast(p.data, src)
    for b in model.buffers():
        dist.broadcast(b, src)


def allreduce_max_scalar(t):
    """4-byte MAX all-reduce (the optional global BerHu threshold)."""
    if world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t
