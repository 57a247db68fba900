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
        assert torch.equal(flat, torch.arange(1000, dtype=torch.float32) * (world * (world + 1) // 2))
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
        # every rank needs all shards' gradients to know the expected mean: two ranks compute them all themselves; eight
        # compute their own and exchange them with all_gather_object (a different collective from the all-reduce under test)
        if world <= 2:
            grads = [O.train_step("DtoD", {k: v.clone() for k, v in sd0.items()}, shards[i], {})["grads"] for i in range(world)]
        else:
            import torch.distributed as dist
            mine = O.train_step("DtoD", {k: v.clone() for k, v in sd0.items()}, shards[rank], {})["grads"]
            grads = [None] * world
            dist.all_gather_object(grads, {k: v.clone() for k, v in mine.items()})
        arena.bind_grads()
        for k, p in model.named_parameters():
            p.grad.copy_(grads[rank][k])            # what this rank's backward would have written
        D.sync_gradients(model, None)               # SUM all-reduce, then 1/world
        for k, p in model.named_parameters():
            want = sum(g[k] for g in grads) / world
            # (gloo sums the ranks in ring order, this loop in rank order: fp32 summation-order noise grows with the rank count)
            torch.testing.assert_close(p.grad, want, rtol=5e-6 * world, atol=5e-8 * world + 2.5e-7 * world * float(want.abs().max()))
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
            torch.testing.assert_close(p.grad, want, rtol=5e-6 * world, atol=5e-8 * world + 2.5e-7 * world * float(want.abs().max()))
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
@pytest.mark.parametrize("world", [2, 4])
def test_gradient_allreduce_gloo(world):
    """world = 4: more than two ranks on the gloo backend (eight -- BASELINE configs[3] -- pass too but take 2.5 minutes of start-up on
    this 8-core box) -- parameter
    broadcast, the whole-arena and the bucketed / overlapped gradient reduction against the mean of all shards' gradients."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=540) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(r, "ok") for r in range(world)], res


def test_launch_ranks_spawns_one_process_per_device(tmp_path):
    """`GDN_main --gpu_num 0,1,2` (the reference's multi-GPU idiom, README.md:82) -> distributed.launch_ranks: one fresh
    process per listed device with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set; the ranks find each other (gloo here)."""
    import json
    import pathlib
    import sys
    root = pathlib.Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root / "gdn-pytorch_amd"))
    from gdn_amd import distributed as D
    env = {"PYTHONPATH": os.pathsep.join([str(root / "tests"), str(root / "gdn-pytorch_amd"), str(root)]),
           "GDN_DIST_BACKEND": "gloo"}
    rc = D.launch_ranks([str(tmp_path), "-1"], ["0", "1", "2"], module="spawn_worker", extra_env=env, timeout=120)
    assert rc == 0
    seen = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(3)]
    assert [s["rank"] for s in seen] == [0, 1, 2] and [s["local_rank"] for s in seen] == [0, 1, 2]
    assert all(s["world"] == 3 and s["sum"] == 6.0 and s["visible"] == "0,1,2" for s in seen)


def test_launch_ranks_propagates_failure(tmp_path):
    """A rank that exits non-zero ends the job with that code; the surviving ranks (blocked in the collective) are stopped."""
    import pathlib
    import sys
    root = pathlib.Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root / "gdn-pytorch_amd"))
    from gdn_amd import distributed as D
    env = {"PYTHONPATH": os.pathsep.join([str(root / "tests"), str(root / "gdn-pytorch_amd"), str(root)]),
           "GDN_DIST_BACKEND": "gloo"}
    rc = D.launch_ranks([str(tmp_path), "1"], ["0", "1"], module="spawn_worker", extra_env=env, timeout=120)
    assert rc == 7


def test_gdn_main_gpu_num_list_goes_through_the_launcher(monkeypatch):
    """main() with --gpu_num 0,1,2,3 and no launcher environment hands the SAME argv to launch_ranks with the four devices
    and never touches the GPU itself; with RANK set (a child, or torch.distributed.run) it runs in-process."""
    import pathlib
    import sys
    root = pathlib.Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root / "gdn-pytorch_amd"))
    from gdn_amd import GDN_main
    calls = []
    monkeypatch.setattr(GDN_main.D, "launch_ranks", lambda argv, devices, **kw: calls.append((list(argv), list(devices))) or 0)
    monkeypatch.setattr(GDN_main, "run", lambda args, *a, **k: calls.append("run") or "ran")
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    argv = ["DATA", "--mode", "DtoD", "--gpu_num", "0,1,2,3", "--synthetic"]
    assert GDN_main.main(argv) is None
    assert calls == [(argv, ["0", "1", "2", "3"])]
    monkeypatch.setenv("RANK", "2")
    monkeypatch.setenv("WORLD_SIZE", "4")
    assert GDN_main.main(argv) == "ran" and calls[-1] == "run"
    monkeypatch.delenv("RANK")
    monkeypatch.delenv("WORLD_SIZE")
    assert GDN_main.main(["DATA", "--gpu_num", "3", "--synthetic"]) == "ran"      # a single device: in-process as before


def test_rank_sharded_loader_order():
    """Every rank shuffles with the same order seed and takes order[rank::world]: the shards are disjoint, equally long and
    cover the epoch once (DistributedSampler semantics)."""
    import pathlib
    import sys
    root = pathlib.Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root / "gdn-pytorch_amd"))
    from gdn_amd.datasets import GpuAugmentLoader
    ds = list(range(103))
    loaders = [GpuAugmentLoader(ds, 5, "cpu", train=True, seed=7 + r, rank=r, world=4, order_seed=99, drop_last=True)
               for r in range(4)]
    orders = [ld._epoch_order() for ld in loaders]
    assert all(len(o) == 25 for o in orders) and all(len(ld) == 5 for ld in loaders)
    flat = [i for o in orders for i in o]
    assert len(set(flat)) == 100 and set(flat) <= set(ds)
    single = GpuAugmentLoader(ds, 5, "cpu", train=True, seed=7, order_seed=99)._epoch_order()
    assert sorted(single) == ds and [single[r::4][:25] for r in range(4)] == orders


def _run_bench(args, env_extra=None, drop=("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")):
    import pathlib
    import subprocess
    import sys
    root = pathlib.Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(root / "bench.py"), *args], env=env, capture_output=True, text=True, timeout=300)


@pytest.mark.timeout(400)
def test_bench_gpus_n_self_launches_ranks_gloo():
    """`python bench.py --gpus 2` as a plain command (no launcher environment) starts two fresh rank processes itself --
    the reference's one-command multi-GPU idiom (GDN_main.py:24,150-173) applied to the program the driver runs for the
    scaling curve -- and forwards rank 0's JSON record as the LAST line of its own stdout.  (--selftest-launch: the rank
    plumbing without GPU work; gloo.)"""
    import json
    r = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--selftest-launch"], {"GDN_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    rec = json.loads(lines[-1])
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["dist_backend"] == "gloo"
    assert rec["token_sum"] == 3.0 and rec["spawned"] is True and rec["steps"] == 3
    assert "rank 1 of 2 up" in r.stderr                      # the other rank ran too, and its output stayed off stdout


def test_bench_gpus_mismatch_is_an_error():
    """--gpus N that the launcher environment does not honour is rc 2 with a message, not a warning; and asking for more
    GPUs than the machine has fails before anything is launched, naming the device count."""
    r = _run_bench(["--gpus", "2", "--selftest-launch"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"}, drop=())
    assert r.returncode == 2 and "WORLD_SIZE is 1" in r.stderr and r.stdout.strip() == ""
    import torch
    if torch.cuda.device_count() < 2:
        r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"])
        assert r.returncode == 2 and "%d GPU(s) are visible" % torch.cuda.device_count() in r.stderr
        assert r.stdout.strip() == ""


def test_visible_devices_maps_through_an_outer_restriction(monkeypatch):
    """launch_ranks must not overwrite a scheduler's HIP_VISIBLE_DEVICES with raw indices: the requested
    devices index the visible list; an index beyond it is refused."""
    import pathlib
    import sys
    root = pathlib.Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root / "gdn-pytorch_amd"))
    from gdn_amd import distributed as D
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
    assert D.visible_devices(["0", "1", "2"]) == "0,1,2"
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "4,5,6,7")
    assert D.visible_devices([0, 1]) == "4,5" and D.visible_devices(["3", "0"]) == "7,4"
    with pytest.raises(RuntimeError, match="exposes only 4"):
        D.visible_devices(["4"])


def test_launch_ranks_leaves_no_child_behind_on_timeout(tmp_path):
    """The wait loop's timeout (like an exception or a signal in the parent) ends every child: terminate, then kill."""
    import pathlib
    import subprocess
    import sys
    import time
    root = pathlib.Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root / "gdn-pytorch_amd"))
    from gdn_amd import distributed as D
    script = tmp_path / "sleeper.py"
    script.write_text("import os, sys, time, signal\n"
                      "signal.signal(signal.SIGTERM, signal.SIG_IGN)\n"          # needs the kill escalation
                      "open(sys.argv[1] + '/pid%s' % os.environ['RANK'], 'w').write(str(os.getpid()))\n"
                      "time.sleep(600)\n")
    t0 = time.time()
    rc = D.launch_ranks([str(tmp_path)], [None, None], script=str(script), timeout=3)
    assert rc == 124 and time.time() - t0 < 60
    for r in range(2):
        pid = int((tmp_path / ("pid%d" % r)).read_text())
        assert subprocess.run(["kill", "-0", str(pid)], capture_output=True).returncode != 0, "rank %d survived" % r


def _shared_gpu_worker(rank, world, port, q, same):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.pop("GDN_X3", None)
    import sys
    import pathlib
    import warnings
    root = pathlib.Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root)); sys.path.insert(0, str(root / "gdn-pytorch_amd"))
    from gdn_amd import distributed as D
    try:
        # (no GPU here: the identity of "the device this rank drives" is injected)
        D.device_identity = lambda local_rank: "box/GPU-0" if same else "box/GPU-%d" % local_rank
        with warnings.catch_warnings(record=True) as seen:
            warnings.simplefilter("always")
            D.init(backend="gloo")
        from gdn_amd import ops
        # the switch is a Python-side setting handed to the library as GDN_HINT_NO_X3, not an environment variable
        q.put((rank, D.SHARED_GPU_RANKS, "1" if ops.x3_enabled() else "0", sum("share one GPU" in str(w.message) for w in seen),
               os.environ.get("GDN_X3")))
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    except Exception as e:      # noqa: BLE001
        q.put((rank, "error", repr(e), 0))


@pytest.mark.parametrize("same", [True, False])
def test_ranks_sharing_one_gpu_switch_the_bf16x3_gemms_off(same):
    """distributed.init: ranks whose device identities coincide (a 1-GPU box under the gloo hook, an oversubscribed
    launch) switch the bf16 x 3 GEMMs off (ops.set_x3(False) -> GDN_HINT_NO_X3 on every geometry) -- the measured cross-process interference of barrier-paced bf16 matrix bursts with a
    neighbour's FFT kernels -- and rank 0 says so once; ranks on different GPUs change nothing."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_shared_gpu_worker, args=(r, 2, port, q, same)) for r in range(2)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    assert all(g[1] != "error" for g in got), got
    if same:
        assert [(g[1], g[2]) for g in got] == [(2, "0"), (2, "0")] and got[0][3] == 1 and got[1][3] == 0
    else:
        assert [(g[1], g[2], g[3]) for g in got] == [(1, "1", 0), (1, "1", 0)]
    assert all(g[4] is None for g in got), "the guard must not touch the environment"


def _accum_worker(rank, world, port, q, overlap):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), GDN_OVERLAP_ALLREDUCE="1" if overlap else "0")
    import sys
    import pathlib
    root = pathlib.Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root)); sys.path.insert(0, str(root / "gdn-pytorch_amd"))
    torch.set_num_threads(1)
    from gdn_amd import distributed as D
    from gdn_amd import engine as E
    try:
        D.init(backend="gloo")
        torch.manual_seed(0)
        model = torch.nn.Sequential(torch.nn.Conv2d(4, 8, 3, bias=False), torch.nn.BatchNorm2d(8), torch.nn.Conv2d(8, 8, 1, bias=False),
                                    torch.nn.BatchNorm2d(8), torch.nn.Conv2d(8, 2, 3, bias=False))
        arena = E.ParamArena(model, torch.device("cpu"))
        model._gdn_param_arena = arena
        params = list(model.parameters())

        def g(step, r):                                   # the "local gradient" of rank r in backward number `step`
            gen = torch.Generator().manual_seed(1000 * step + r)
            return [torch.randn(p.shape, generator=gen) for p in params]

        def backward(step):
            """What _Bridge.backward does around the tape, with a tape that writes this rank's gradients in reverse order."""
            ctx = E.Ctx(record=True, arena=arena)
            pending = E.begin_backward(model, arena, ctx)
            for p, gp in reversed(list(zip(params, g(step, rank)))):
                p.grad.copy_(gp)
                ctx.grads_done(p)
            E.end_backward(arena, pending)
            return ctx.reducer is not None

        def zero():
            for p in params:
                p.grad = None

        def want(steps):
            return [sum(g(s, r)[i] for s in steps for r in range(world)) for i in range(len(params))]

        def check(steps, what):
            """sync_gradients(model, None) leaves the MEAN over the ranks of the accumulated gradient"""
            for p, w in zip(params, want(steps)):
                torch.testing.assert_close(p.grad * world, w, rtol=1e-5, atol=1e-6, msg=lambda m: what + ": " + m)

        if overlap:
            assert D.attach_reducer(model) is not None
            model._gdn_reducer = D.GradReducer(arena, bucket_elems=200)       # several buckets for this tiny arena
            assert len(model._gdn_reducer.buckets) > 2
        # (a) backward, backward, sync: the second backward meets reductions of the first still in flight
        assert backward(1) == overlap
        assert backward(2) is False                        # an accumulating backward never overlaps
        D.sync_gradients(model, None)
        check((1, 2), "backward, backward, sync")
        # (b) backward, sync, backward, sync: the first sum must not be reduced (or averaged) a second time
        zero()
        backward(3); D.sync_gradients(model, None)
        check((3,), "backward, sync")
        backward(4); D.sync_gradients(model, None)
        check((3, 4), "backward, sync, backward, sync")
        # (c) three backwards, one sync; then (d) a fresh step after zero_grad overlaps again
        zero()
        backward(5); backward(6); backward(7)
        D.sync_gradients(model, None)
        check((5, 6, 7), "three backwards, one sync")
        zero()
        assert backward(8) == overlap
        D.sync_gradients(model, None)
        check((8,), "fresh step after zero_grad")
        # (e) with an optimizer the arena keeps SUMS and the 1/world rides in its grad_scale
        class _Opt:
            grad_scale = 1.0
        opt = _Opt()
        zero()
        backward(9); D.sync_gradients(model, opt); backward(10); D.sync_gradients(model, opt)
        assert opt.grad_scale == 1.0 / world
        for p, w in zip(params, want((9, 10))):
            torch.testing.assert_close(p.grad, w, rtol=1e-5, atol=1e-6)
        assert arena.carry_reduced is None and arena.reduced
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))
    finally:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("overlap", [True, False])
def test_gradient_accumulation_under_data_parallelism(overlap):
    """backward without zero_grad under data parallelism.  Every rank's contributions must be summed over the
    ranks exactly once, whichever way backwards and sync_gradients interleave, with and without the overlapped reducer."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_accum_worker, args=(r, world, port, q, overlap)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def _store_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    import pathlib
    root = pathlib.Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root / "gdn-pytorch_amd"))
    from gdn_amd import distributed as D
    import torch.distributed as dist
    try:
        calls = []
        real = dist.all_gather_object
        dist.all_gather_object = lambda *a, **k: calls.append("all_gather_object") or real(*a, **k)
        D.device_identity = lambda lr: None if lr == 2 else "host/uuid-%d/0:%d:0" % (lr // 2, lr // 2)
        D.init(backend="gloo")
        ids = D.exchange_through_store(rank, world, "second-%d" % rank, key="gdn/test")
        q.put((rank, D.SHARED_GPU_RANKS, ids, calls))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:      # noqa: BLE001
        q.put((rank, "error", repr(e), []))


def test_device_identities_travel_through_the_store_not_a_collective():
    """the start-up guard compares (host, device) identities through the rendezvous store; no
    collective of the group's backend runs before the first real one.  Three ranks: 0 and 1 share a device, rank 2's
    identity is unknown (None) and therefore shares with nobody."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_store_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    assert all(g[1] != "error" for g in got), got
    assert [g[1] for g in got] == [2, 2, 1]
    assert all(g[2] == ["second-0", "second-1", "second-2"] and g[3] == [] for g in got), got


@pytest.mark.parametrize("sig", ["HUP", "KILL"])
def test_launcher_death_takes_the_ranks_along(tmp_path, sig):
    """the ranks lead sessions of their own, so SIGHUP to the launcher (a dropped terminal) must be passed on, and
    a SIGKILLed launcher -- which runs no clean-up -- must still not leave GPU ranks behind (PR_SET_PDEATHSIG)."""
    import pathlib
    import signal
    import subprocess
    import sys
    import time
    root = pathlib.Path(__file__).resolve().parent.parent
    sleeper = tmp_path / "sleeper.py"
    sleeper.write_text("import os, sys, time\n"
                       "open(sys.argv[1] + '/pid%s' % os.environ['RANK'], 'w').write(str(os.getpid()))\n"
                       "time.sleep(600)\n")
    launcher = tmp_path / "launcher.py"
    launcher.write_text("import sys\nsys.path.insert(0, %r)\nfrom gdn_amd import distributed as D\n"
                        "sys.exit(D.launch_ranks([%r], [None, None], script=%r, timeout=300))\n"
                        % (str(root / "gdn-pytorch_amd"), str(tmp_path), str(sleeper)))
    lp = subprocess.Popen([sys.executable, str(launcher)])
    try:
        t0 = time.time()
        while not all((tmp_path / ("pid%d" % r)).exists() and (tmp_path / ("pid%d" % r)).read_text() for r in range(2)):
            assert time.time() - t0 < 120 and lp.poll() is None
            time.sleep(0.1)
        pids = [int((tmp_path / ("pid%d" % r)).read_text()) for r in range(2)]
        lp.send_signal(getattr(signal, "SIG" + sig))
        lp.wait(timeout=60)
        t0 = time.time()
        while any(subprocess.run(["kill", "-0", str(p)], capture_output=True).returncode == 0 for p in pids):
            assert time.time() - t0 < 30, "ranks %s survived SIG%s of the launcher" % (pids, sig)
            time.sleep(0.1)
    finally:
        if lp.poll() is None:
            lp.kill()
