"""Whole-training-step hipGraph capture.

The engine's forward, the fused loss kernels, the tape replay of backward and the fused Adam are all stream-ordered,
allocation-free (beyond torch's caching allocator) and sync-free, so one training step -- ~570 launches, 10-17 ms of
host enqueue time -- can be captured once and replayed as a single graph launch.  What has to live in device memory
for that: the optimizer's step counter / bias corrections / learning rate (``optim.Adam(capturable=True)`` ->
``gdn_adam_step_dev``) and the batch (static input buffers, refilled before every replay).

Single-process only: with data parallelism the RCCL all-reduce stays outside any capture, so ``GraphedTrainStep``
refuses world sizes > 1 and the eager path is used there.
"""
import torch

from . import distributed as D
from . import engine as E
from ._lib import GdnError


class GraphedTrainStep:
    """Capture ``step_fn(*static_inputs)`` -- forward, losses, zero_grad, backward, optimizer.step -- and replay it.

    step_fn must use the tensors it is handed (they are the graph's static input buffers) and return a tensor or a tuple
    of device tensors (e.g. the loss terms); the same static output tensors are returned by every replay.
    optimizer must be ``gdn_amd.optim.Adam(..., capturable=True)``."""

    def __init__(self, step_fn, example_inputs, optimizer, warmup=3):
        if D.world_size() > 1:
            raise GdnError("GraphedTrainStep is single-process: the RCCL gradient all-reduce is not captured")
        if not getattr(optimizer, "capturable", False):
            raise GdnError("GraphedTrainStep needs optim.Adam(..., capturable=True) (device-side step counter)")
        self.optimizer = optimizer
        self.static_inputs = [t.clone() if torch.is_tensor(t) else t for t in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):            # warm-up: allocates every workspace, the arena, the optimizer state
            for _ in range(max(1, warmup)):
                out = step_fn(*self.static_inputs)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.warmup_steps = max(1, warmup)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            out = step_fn(*self.static_inputs)
        self.static_outputs = out
        self.replays = 0
        E.bump_graph_epoch()

    def __call__(self, *inputs):
        for dst, src in zip(self.static_inputs, inputs):
            if torch.is_tensor(dst):
                dst.copy_(src, non_blocking=True)
        self.optimizer.refresh_hyper()           # a learning-rate decay since the last replay reaches the device here
        self.graph.replay()
        self.replays += 1
        E.bump_graph_epoch()                     # BN running statistics changed behind torch's version counters
        return self.static_outputs
