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
    return rank, local_rank, world


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
        for i, f in enumerate(self.fired):
            if not f:
                self._fire(i)
        for w in self.works:
            w.wait()
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
    if not (red is not None and red.arena is ar and red.finish()):
        for w in allreduce_flat(ar.grad):
            w.wait()
    if optimizer is not None and hasattr(optimizer, "grad_scale"):
        optimizer.grad_scale = 1.0 / ws
    else:
        ar.grad.mul_(1.0 / ws)


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
def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(argv, devices, module="gdn_amd.GDN_main", extra_env=None, timeout=None):
    """Start one fresh child process per entry of `devices` running ``python -m <module> <argv>`` with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, wait for all of them and return the first non-zero exit code (0 if
    all succeeded; the others are terminated as soon as one fails).  The parent must not have touched the GPU: the
    children are started with subprocess (fork + exec of a new interpreter), never by exec-ing over an initialised
    process.  LOCAL_RANK indexes the visible-device list, which is set to exactly `devices` for every child (RCCL needs
    to see its peers' devices for xGMI peer access)."""
    import subprocess
    import sys
    import time
    world = len(devices)
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        if devices and all(d is not None for d in devices):
            env["HIP_VISIBLE_DEVICES"] = ",".join(str(d) for d in devices)
        env["GDN_SPAWNED"] = "1"
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable, "-m", module, *argv], env=env))
    rc, t0 = 0, time.time()
    alive = list(procs)
    while alive:
        for pr in list(alive):
            code = pr.poll()
            if code is None:
                continue
            alive.remove(pr)
            if code != 0 and rc == 0:
                rc = code
                for other in alive:               # a dead rank would leave the others hanging in a collective
                    other.terminate()
        if timeout is not None and time.time() - t0 > timeout and alive:
            for other in alive:
                other.kill()
            rc = rc or 124
        time.sleep(0.05)
    return rc
