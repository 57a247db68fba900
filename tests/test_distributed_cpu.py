"""World-size-2 gloo tests (CPU) of the data-parallel path: bucketed all-reduce of the flat
gradient arena, parameter broadcast, and the semantics statement of SURVEY 8(e): each rank ==
one reference run on its own shard (rank-local BatchNorm statistics and BerHu threshold),
gradients averaged over ranks."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    import pathlib
    root = pathlib.Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root)); sys.path.insert(0, str(root / "gdn-pytorch_amd"))
    torch.set_num_threads(2)
    from gdn_amd import distributed as D
    from gdn_amd import engine as E
    import gdn_amd.AE_model_unet as M
    from oracle import gdn_oracle as O
    try:
        r, lr, w = D.init(backend="gloo")
        assert (r, w) == (rank, world) and D.world_size() == world and D.rank() == rank
        # --- bucketed all-reduce of a flat buffer (several buckets, ragged tail) ---
        flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)
        for wk in D.allreduce_flat(flat, bucket_elems=300):
            wk.wait()
        assert torch.equal(flat, torch.arange(1000, dtype=torch.float32) * 3)
        assert [b.numel() for b in D.flat_buckets(flat, 300)] == [300, 300, 300, 100]
        # --- one training step per rank on its own shard, CPU oracle arithmetic ---
        torch.manual_seed(0)                       # identical init on every rank
        model = M.AutoEncoder_DtoD(height=32, width=64)
        if rank == 1:                               # perturb, then check the broadcast repairs it
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(1.0)
        arena = E.ParamArena(model, torch.device("cpu"))
        model._gdn_param_arena = arena
        D.broadcast_parameters(model, src=0)
        sd0 = O.init_state_dict("AutoEncoder_DtoD", seed=0)
        assert all(torch.equal(v, sd0[k]) for k, v in model.state_dict().items())
        shards = [O.synthetic_batch(1, 32, 64, seed=10 + i) for i in range(world)]
        grads = []
        for i in range(world):                      # every rank computes all shards to know the expected mean
            res = O.train_step("DtoD", {k: v.clone() for k, v in sd0.items()}, shards[i], {})
            grads.append(res["grads"])
        arena.bind_grads()
        for k, p in model.named_parameters():
            p.grad.copy_(grads[rank][k])            # what this rank's backward would have written
        D.sync_gradients(model, None)               # SUM all-reduce, then 1/world
        for k, p in model.named_parameters():
            want = sum(g[k] for g in grads) / world
            torch.testing.assert_close(p.grad, want, rtol=1e-5, atol=1e-7)
        # --- overlapped reduction: buckets fire as soon as their last parameter is marked ---
        red = D.GradReducer(arena, bucket_elems=200000)
        assert len(red.buckets) > 4 and sum(b[2] for b in red.buckets) == len(arena.items)
        for k, p in model.named_parameters():
            p.grad.copy_(grads[rank][k])
        red.begin()
        order = [p for p in model.parameters()][::-1]          # backward visits parameters roughly in reverse
        red.mark(order[:len(order) // 2])
        assert any(red.fired) and not all(red.fired)
        red.mark(order[len(order) // 2:-3])                    # three parameters never get marked
        assert red.finish() and not red.active
        for k, p in model.named_parameters():
            want = sum(g[k] for g in grads)
            torch.testing.assert_close(p.grad, want, rtol=1e-5, atol=1e-7)
        t = torch.tensor([float(rank)])
        assert D.allreduce_max_scalar(t).item() == world - 1
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))
    finally:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_gradient_allreduce_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=540) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
