"""Fused Adam over the model's flat parameter arena.

Same update rule as the reference's ``optim.Adam(model.parameters(), lr,
[momentum, beta], eps=1e-08, weight_decay=5e-4)`` (GDN_main.py:157,173:
coupled L2 weight decay on every parameter, BN affine included), executed as
ONE kernel launch over the arena instead of 124-136 per-tensor launches.
"""
import torch

from . import ops
from ._lib import GdnError


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._flat = {}        # id(arena) -> {"m","v","step"}
        self.grad_scale = 1.0  # set to 1/world_size by the data-parallel wrapper

    def _arena_groups(self, group):
        """Split a param group into (arena, covers_whole_arena) and stragglers."""
        arenas, loose = {}, []
        for p in group["params"]:
            ar = getattr(p, "_gdn_arena", None)
            if ar is not None and ar.intact():
                arenas.setdefault(id(ar), (ar, []))[1].append(p)
            else:
                loose.append(p)
        return arenas, loose

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            arenas, loose = self._arena_groups(group)
            for ar, ps in arenas.values():
                if len(ps) != len(ar.items):
                    loose.extend(ps)      # partial coverage: fall back to per-tensor launches
                    continue
                if any(p.grad is None for p in ps):
                    raise GdnError("fused Adam: some parameters have no gradient; run backward first")
                st = self._flat.get(id(ar))
                if st is None:
                    st = {"m": ops.zeros((ar.numel,), ar.device), "v": ops.zeros((ar.numel,), ar.device), "step": 0}
                    self._flat[id(ar)] = st
                st["step"] += 1
                ops.adam_step(ar.data, ar.grad, st["m"], st["v"], group["lr"], b1, b2, group["eps"],
                              group["weight_decay"], st["step"], self.grad_scale)
                ar.touch()        # the kernel wrote the parameters behind torch's back: bf16 shadows are stale
            for p in loose:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise GdnError("fused Adam needs GPU parameters: the HIP path has no CPU fallback")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["m"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["v"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                g = p.grad
                if g.stride() != p.stride():
                    g = torch.empty_like(p, memory_format=torch.preserve_format).copy_(g)
                ops.adam_step(p.data, g, st["m"], st["v"], group["lr"], b1, b2, group["eps"], group["weight_decay"],
                              st["step"], self.grad_scale)
                if getattr(p, "_gdn_arena", None) is not None:
                    p._gdn_arena.touch()
        return loss
