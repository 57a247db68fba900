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


def forced():
    """GDN_FORCE_DIST=1: run the data-parallel machinery (process group, bucketed async all-reduce, GradReducer overlap)
    even at world size 1.  A 1-rank RCCL all-reduce is the identity, but it goes through RCCL's own stream, its
    async_op work handles and their ordering against the kernels just enqueued on the compute stream -- the part of the
    path that gloo does not model and a 1-GPU box can still exercise."""
    return os.environ.get("GDN_FORCE_DIST") == "1"


def init(backend=None):
    """Initialise torch.distributed from the launcher's environment (no-op for world size 1 unless GDN_FORCE_DIST=1)."""
    rank, local_rank, world = env_rank()
    if (world > 1 or forced()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # "nccl" is RCCL on ROCm.  GDN_DIST_BACKEND=gloo is a test hook: RCCL refuses two ranks on one device, gloo
            # moves the same buckets through the host, so the multi-rank code paths can be exercised on a 1-GPU box.
            backend = os.environ.get("GDN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        _guard_shared_gpu(rank, local_rank, world)
    return rank, local_rank, world


SHARED_GPU_RANKS = 1       # ranks of this job (this one included) that drive the same physical GPU as this rank


def device_identity(local_rank):
    """What makes two ranks users of ONE physical GPU: host name + the device's UUID + its PCI address.  None when it
    cannot be told (no GPU, or a runtime that reports neither a UUID nor a PCI address: an all-zero identity would make
    every rank of a healthy multi-GPU host look like a neighbour).  A GDN_SINGLE_DEVICE run (test hook: every rank on
    device 0) resolves device 0."""
    import socket
    if not torch.cuda.is_available():
        return None
    n = torch.cuda.device_count()
    idx = 0 if os.environ.get("GDN_SINGLE_DEVICE") == "1" else (local_rank % n if n else 0)
    pr = torch.cuda.get_device_properties(idx)
    uuid = str(getattr(pr, "uuid", "") or "")
    if not uuid.strip("0-") or uuid == "None":           # all zeros / absent: not an identity
        uuid = ""
    pci = [getattr(pr, a, None) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id")]
    pci = "" if any(v is None for v in pci) or (not uuid and not any(pci)) else "%x:%x:%x" % tuple(pci)
    if not uuid and not pci:
        return None
    return "%s/%s/%s" % (socket.gethostname(), uuid, pci)


def exchange_through_store(rank, world, value, key="gdn/dev", timeout_s=120.0):
    """Every rank's `value` (a short string), gathered through the rendezvous STORE of the default process group -- host
    TCP traffic the group's construction has already exercised -- rather than through a collective of the group's backend:
    at start-up nothing has yet proven that RCCL can run one, and a device-identity check must not be the first to try."""
    from datetime import timedelta
    store = dist.distributed_c10d._get_default_store()
    store.set("%s/%d" % (key, rank), value if value is not None else "")
    keys = ["%s/%d" % (key, r) for r in range(world)]
    store.wait(keys, timedelta(seconds=timeout_s))
    out = []
    for k in keys:
        v = store.get(k)
        v = v.decode("utf-8", "replace") if isinstance(v, (bytes, bytearray)) else str(v)
        out.append(v or None)
    return out


def _guard_shared_gpu(rank, local_rank, world):
    """Several ranks on ONE MI355X (a 1-GPU test box with the gloo hook, or an oversubscribed launch): the bf16 x 3 GEMMs
    are switched off for this process (ops.set_x3(False): every layer planned from now on carries GDN_HINT_NO_X3 and its
    Winograd per-bin GEMMs return to the fp32 matrix instruction).
    Measured on this pool (profiles/r03_neighbour_mfma.txt): a kernel that alternates bursts of bf16
    matrix instructions with workgroup barriers -- which is what a tiled bf16 GEMM is -- perturbs FFT-type kernels of
    ANOTHER process that shares the GPU (16-lane pieces of their results change; reproduced with a 30-line kernel as the
    neighbour and rocFFT as the victim), while one process per GPU -- the deployment this package is built for -- and the
    fp32 matrix instruction are not affected.  An explicit choice (GDN_X3 in the environment) is respected.
    The identities travel through the rendezvous store, not through a collective (exchange_through_store)."""
    global SHARED_GPU_RANKS
    if world < 2:
        return
    me = device_identity(local_rank)
    try:
        ids = exchange_through_store(rank, world, me)
    except Exception as e:      # noqa: BLE001 -- the guard must never be what stops a job
        import warnings
        warnings.warn("could not compare the ranks' devices (%r); assuming one process per GPU" % (e,), RuntimeWarning)
        return
    SHARED_GPU_RANKS = sum(1 for i in ids if i is not None and i == me) if me is not None else 1
    if SHARED_GPU_RANKS > 1:
        from . import ops
        if not ops.x3_explicit():
            ops.set_x3(False)
            if rank == 0:
                import warnings
                warnings.warn("%d ranks share one GPU: bf16 x 3 GEMMs disabled for this job (GDN_HINT_NO_X3); one process "
                              "per GPU is the supported layout" % SHARED_GPU_RANKS, RuntimeWarning)


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def active():
    """True when gradients go through the all-reduce path."""
    return dist.is_initialized() and (dist.get_world_size() > 1 or forced())


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
    if not active():
        return []
    works = [dist.all_reduce(b, op=dist.ReduceOp.SUM, async_op=True) for b in flat_buckets(flat, bucket_elems)]
    if not async_op:
        for w in works:
            w.wait()
        return []
    return works


class _Stats:
    """What the all-reduce cost a step, for bench.py's multi-rank record: off unless stats_begin()."""

    def __init__(self):
        self.on = False
        self.reset()

    def reset(self):
        self.syncs = self.buckets = self.bytes = self.fired_early = 0
        self.host_wait_s = 0.0
        self.events = []               # (before, after) event pairs on the compute stream around the waits

    def waits(self, works, arena_grad):
        """Wait for `works`; account the host time spent (gloo blocks the host) and, on a GPU, how long the COMPUTE stream
        stood still for them (RCCL's wait() is a stream dependency, not a host wait): the exposed all-reduce time."""
        import time
        ev = None
        if self.on and arena_grad.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        t0 = time.perf_counter()
        for w in works:
            w.wait()
        if self.on:
            self.host_wait_s += time.perf_counter() - t0
            if ev is not None:
                ev[1].record()
                self.events.append(ev)


STATS = _Stats()


def stats_begin():
    STATS.reset()
    STATS.on = True


def stats_report(steps):
    """Per-step means since stats_begin() (call after torch.cuda.synchronize())."""
    STATS.on = False
    dev_ms = sum(a.elapsed_time(b) for a, b in STATS.events) if STATS.events else None
    n = max(1, steps)
    return {"allreduce_exposed_ms": None if dev_ms is None else round(dev_ms / n, 4),
            "allreduce_host_wait_ms": round(STATS.host_wait_s * 1e3 / n, 4),
            "buckets": STATS.buckets // max(1, STATS.syncs), "buckets_in_flight_before_backward_returned": STATS.fired_early // max(1, STATS.syncs),
            "bytes_reduced": STATS.bytes // max(1, STATS.syncs), "syncs": STATS.syncs,
            "overlap": os.environ.get("GDN_OVERLAP_ALLREDUCE", "1") != "0"}


class GradReducer:
    """Overlaps the gradient all-reduce with backward.

    The gradient arena is cut into contiguous buckets; the engine reports every parameter whose
    gradient has just been written (``mark``), and a bucket is all-reduced (async, on RCCL's own
    stream) as soon as its last parameter is done.  Backward reaches the decoder and the 512-channel
    bottleneck first -- 80 % of the bytes -- so only the encoder's tail is exposed after backward.
    """

    def __init__(self, arena, bucket_elems=BUCKET_ELEMS // 2):
        self.arena = arena
        self.buckets = []            # [start, end, n_params]
        self.bucket_of = {}
        cur = None
        for p, off, n, _ in arena.items:
            if cur is None or off + n - cur[0] > bucket_elems:
                cur = [off, off + n, 0]
                self.buckets.append(cur)
            cur[1] = off + n
            cur[2] += 1
            self.bucket_of[id(p)] = len(self.buckets) - 1
        self.active = False
        self.pending, self.fired, self.works = [], [], []

    def begin(self):
        self.pending = [b[2] for b in self.buckets]
        self.fired = [False] * len(self.buckets)
        self.works = []
        self.active = True

    def _fire(self, i):
        s, e, _ = self.buckets[i]
        self.fired[i] = True
        self.works.append(dist.all_reduce(self.arena.grad[s:e], op=dist.ReduceOp.SUM, async_op=True))

    def mark(self, params):
        if not self.active:
            return
        for p in params:
            i = self.bucket_of.get(id(p))
            if i is None:
                continue
            self.pending[i] -= 1
            if self.pending[i] == 0 and not self.fired[i]:
                self._fire(i)

    def finish(self):
        """Reduce whatever has not been sent yet (parameters that got no gradient this step) and wait."""
        if not self.active:
            return False
        if STATS.on:
            STATS.fired_early += sum(1 for f in self.fired if f)
            STATS.buckets += len(self.buckets)
            STATS.bytes += 4 * sum(e - s_ for s_, e, _ in self.buckets)
        for i, f in enumerate(self.fired):
            if not f:
                self._fire(i)
        STATS.waits(self.works, self.arena.grad)
        self.works = []
        self.active = False
        return True


def attach_reducer(model):
    """Enable overlapped gradient reduction for `model` (no-op for world size 1)."""
    if not active() or os.environ.get("GDN_OVERLAP_ALLREDUCE", "1") == "0":
        return None
    ar = getattr(model, "_gdn_param_arena", None)
    if ar is None:
        return None
    red = getattr(model, "_gdn_reducer", None)
    if red is None or red.arena is not ar:
        red = GradReducer(ar)
        model._gdn_reducer = red
    return red


def sync_gradients(model, optimizer=None):
    """All-reduce the model's gradient arena; the mean is applied by the optimizer's grad_scale.

    If a GradReducer overlapped the reduction with backward, this only waits for its tail."""
    ws = world_size()
    ar = getattr(model, "_gdn_param_arena", None)
    if not active():
        return
    if ar is None:
        raise RuntimeError("sync_gradients: model has no gradient arena yet (run a forward/backward first)")
    red = attach_reducer(model)          # active from the NEXT backward on
    if STATS.on:
        STATS.syncs += 1
    if not (red is not None and red.arena is ar and red.finish()):
        works = allreduce_flat(ar.grad)
        if STATS.on:
            STATS.buckets += len(works)
            STATS.bytes += 4 * ar.grad.numel()
        STATS.waits(works, ar.grad)
    if optimizer is not None and hasattr(optimizer, "grad_scale"):
        optimizer.grad_scale = 1.0 / ws          # the arena holds SUMS over the ranks
        scale = 1.0
    else:
        ar.grad.mul_(1.0 / ws)                   # the arena holds MEANS
        scale = 1.0 / ws
    if getattr(ar, "carry_reduced", None) is not None:
        # gradient accumulation across a sync: the earlier, already reduced gradient was kept out of the arena (as a sum)
        # while the new local contribution was reduced (engine.ParamArena.finish_grads); they meet here, each summed over the
        # ranks exactly once
        ar.grad.add_(ar.carry_reduced, alpha=scale)
        ar.carry_reduced = None
    ar.reduced, ar.reduced_scale = True, scale


def broadcast_parameters(model, src=0):
    """Make every rank start from rank `src`'s weights and BN buffers (once, at start)."""
    if not active():
        return
    ar = getattr(model, "_gdn_param_arena", None)
    if ar is not None:
        dist.broadcast(ar.data, src)
        ar.touch()
    else:
        for p in model.parameters():
            dist.broadcast(p.data, src)
    for b in model.buffers():
        dist.broadcast(b, src)


def allreduce_max_scalar(t):
    """4-byte MAX all-reduce (the optional global BerHu threshold)."""
    if active():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


# ----------------------------------------------------------------------------
# The reference's multi-GPU idiom: `python GDN_main.py DATA --gpu_num 0,1,2,3` (README.md:82, GDN_main.py:24,150-173)
# trains on four GPUs from ONE command (nn.DataParallel).  Here that command becomes one process per listed device.
# ----------------------------------------------------------------------------
def _host_store(world):
    """The rendezvous store of a self-launched job, hosted by the PARENT on a port the kernel picks (port 0): no window
    between choosing a free port and rank 0 binding it.  The children connect as clients (torch's env:// rendezvous does
    that for every rank when TORCHELASTIC_USE_AGENT_STORE=True -- what torch.distributed.run's agent sets).  CPU only: the
    parent never touches the GPU."""
    from datetime import timedelta
    store = dist.TCPStore("127.0.0.1", 0, world, is_master=True, timeout=timedelta(seconds=1800), wait_for_workers=False)
    return store, store.port


def visible_devices(devices):
    """HIP_VISIBLE_DEVICES for the children of launch_ranks.  `devices` index the devices THIS process may use: if the
    scheduler already restricted them (HIP_VISIBLE_DEVICES set), the list is mapped through that restriction instead of
    overwriting it with indices that would name someone else's GPUs; an index beyond it is refused."""
    devices = [str(d) for d in devices]
    outer = os.environ.get("HIP_VISIBLE_DEVICES")
    if outer is None or outer.strip() == "":
        return ",".join(devices)
    allowed = [d.strip() for d in outer.split(",") if d.strip() != ""]
    mapped = []
    for d in devices:
        if not d.isdigit() or int(d) >= len(allowed):
            raise RuntimeError("device %s requested but HIP_VISIBLE_DEVICES=%s exposes only %d device(s)"
                               % (d, outer, len(allowed)))
        mapped.append(allowed[int(d)])
    return ",".join(mapped)


def _stop(procs, grace=5.0):
    """terminate -> wait -> kill for every child still alive (each child leads its own session / process group, so its own
    helpers go with it)."""
    import signal
    import time
    alive = [p for p in procs if p.poll() is None]
    for p in alive:
        try:
            os.killpg(p.pid, signal.SIGTERM)
        except (ProcessLookupError, PermissionError, OSError):
            p.terminate()
    t0 = time.time()
    while alive and time.time() - t0 < grace:
        alive = [p for p in alive if p.poll() is None]
        time.sleep(0.05)
    for p in alive:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except (ProcessLookupError, PermissionError, OSError):
            p.kill()
    for p in procs:
        try:
            p.wait(timeout=grace)
        except Exception:  # noqa: BLE001
            pass


def launch_ranks(argv, devices, module="gdn_amd.GDN_main", extra_env=None, timeout=None, script=None, capture_rank0=False):
    """Start one fresh child process per entry of `devices` running ``python -m <module> <argv>`` (or ``python <script>
    <argv>``) with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, wait for all of them and return the first non-zero exit
    code (0 if all succeeded; the others are stopped as soon as one fails).  The parent must not have touched the GPU:
    the children are started with subprocess (fork + exec of a new interpreter), never by exec-ing over an initialised
    process.  LOCAL_RANK indexes the visible-device list, which is set to exactly `devices` (mapped through an outer
    HIP_VISIBLE_DEVICES restriction) for every child: RCCL needs to see its peers' devices for xGMI peer access.

    capture_rank0: rank 0's stdout is piped and returned as text -> (rc, text); the other ranks' stdout goes to this
    process's stderr, so a caller can forward rank 0's result line as the last line of its own stdout.

    Whatever ends the wait -- success, a failed rank, the timeout, an exception, SIGINT / SIGTERM / SIGHUP / SIGQUIT in the
    parent -- no child survives it: terminate, a grace period, then kill (each child is a session leader, signalled as a
    group; further signals are ignored while that runs).  A launcher that is SIGKILLed takes its ranks with it through
    PR_SET_PDEATHSIG."""
    import signal
    import subprocess
    import sys
    import threading
    import time
    world = len(devices)
    vis = visible_devices(devices) if devices and all(d is not None for d in devices) else None
    store, port = _host_store(world)
    procs, rank0_out, reader = [], [], None
    prev = {}

    def on_signal(signum, _frame):
        raise KeyboardInterrupt("signal %d" % signum)

    in_main = threading.current_thread() is threading.main_thread()
    if in_main:
        # SIGHUP / SIGQUIT too: the children lead sessions of their own, so a terminal or ssh drop no longer reaches them
        # through the process group -- the parent has to pass it on
        for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP, signal.SIGQUIT):
            prev[sg] = signal.signal(sg, on_signal)

    # a SIGKILLed launcher cannot run its `finally`: the kernel delivers SIGKILL to the rank when the launcher's thread ends.
    # the hook runs between fork and exec, possibly beside other threads of the launcher -- it must not import or
    # dlopen anything there (a loader / import lock held by another thread at fork time would deadlock the child), so libc's
    # prctl is resolved HERE, once; and a launcher that died before the prctl ran is noticed by the parent-pid check.
    try:
        import ctypes
        _prctl = ctypes.CDLL("libc.so.6", use_errno=True).prctl
    except Exception:  # noqa: BLE001
        _prctl = None
    launcher_pid, kill_sig = os.getpid(), int(signal.SIGKILL)

    def die_with_parent():
        if _prctl is not None:
            _prctl(1, kill_sig, 0, 0, 0)      # PR_SET_PDEATHSIG
            if os.getppid() != launcher_pid:
                os._exit(1)
    rc = 0
    try:
        for r in range(world):
            env = dict(os.environ)
            env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1",
                        "MASTER_PORT": str(port), "TORCHELASTIC_USE_AGENT_STORE": "True",
                        "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
            if vis is not None:
                env["HIP_VISIBLE_DEVICES"] = vis
            env["GDN_SPAWNED"] = "1"
            if extra_env:
                env.update(extra_env)
            cmd = [sys.executable, script, *argv] if script else [sys.executable, "-m", module, *argv]
            out = None
            if capture_rank0:
                out = subprocess.PIPE if r == 0 else sys.stderr
            procs.append(subprocess.Popen(cmd, env=env, stdout=out, start_new_session=True, preexec_fn=die_with_parent))
        if capture_rank0:
            def pump():
                for line in procs[0].stdout:
                    rank0_out.append(line.decode("utf-8", "replace"))
            reader = threading.Thread(target=pump, daemon=True)
            reader.start()
        t0 = time.time()
        alive = list(procs)
        while alive:
            for pr in list(alive):
                code = pr.poll()
                if code is None:
                    continue
                alive.remove(pr)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 128 - code     # killed by signal n: the shell's 128 + n
                    _stop(alive)                      # a dead rank would leave the others hanging in a collective
            if timeout is not None and time.time() - t0 > timeout and alive:
                _stop(alive, grace=1.0)
                rc = rc or 124
            time.sleep(0.05)
    finally:
        if in_main:                      # a second signal must not abort the clean-up itself
            for sg in prev:
                signal.signal(sg, signal.SIG_IGN)
        _stop(procs)
        if reader is not None:
            reader.join(timeout=5.0)
        for sg, h in prev.items():
            signal.signal(sg, h)
        del store
    return (rc, "".join(rank0_out)) if capture_rank0 else rc
