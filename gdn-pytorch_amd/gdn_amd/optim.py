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
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, capturable=False):
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._flat = {}        # id(arena) -> {"m","v","step"}
        self.grad_scale = 1.0  # set to 1/world_size by the data-parallel wrapper
        # capturable: the step counter, beta^t and the hyper-parameters live in device memory (gdn_adam_step_dev), so
        # step() can be captured in a hipGraph and replayed; call refresh_hyper() after changing lr outside a capture
        self.capturable = bool(capturable)

    def refresh_hyper(self):
        """Push (lr, betas, eps, weight_decay, grad_scale) to the device buffers of the capturable path if they changed."""
        for group in self.param_groups:
            for st in list(self._flat.values()) + [self.state[p] for p in group["params"] if p in self.state]:
                if "hyper" in st and st.get("group") is group:
                    self._push_hyper(st, group)

    def _push_hyper(self, st, group):
        vals = (float(group["lr"]), float(group["betas"][0]), float(group["betas"][1]), float(group["eps"]),
                float(group["weight_decay"]), float(self.grad_scale))
        if st.get("hyper_host") != vals:
            if torch.cuda.is_current_stream_capturing():
                raise GdnError("optimizer hyper-parameters changed inside a graph capture; call refresh_hyper() before it")
            st["hyper"].copy_(torch.tensor(vals, dtype=torch.float32))
            st["hyper_host"] = vals

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

    @staticmethod
    def _dev_state(group, t, device):
        """{double beta1^t, double beta2^t, int32 t, float bc1, float bc2s} for `t` steps TAKEN so far (the kernel advances it
        before it uses it)."""
        import struct
        if torch.cuda.is_current_stream_capturing():
            raise GdnError("capturable Adam: a device step counter would be created (and reset by every replay) inside a graph "
                           "capture; run one eager step with the same gradient coverage first")
        b1, b2 = group["betas"]
        raw = struct.pack("<ddiff", float(b1) ** t, float(b2) ** t, t, 0.0, 0.0)
        return torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)

    def _ensure_dev(self, st, group, device, steps_taken):
        """Device-side step state of the capturable path for one state store: hyper float32[6] and the step state of
        _dev_state(); `steps_taken` = the host-side count of updates this store has had (a resume folds it in by advancing the
        powers on the host first)."""
        if "hyper" not in st:
            st["hyper"] = torch.zeros(6, dtype=torch.float32, device=device)
            st["state"] = self._dev_state(group, steps_taken, device)
            st["group"] = group
        self._push_hyper(st, group)

    def _apply(self, pdata, grad, st, group, device):
        b1, b2 = group["betas"]
        if self.capturable:
            self._ensure_dev(st, group, device, st["step"] - 1)      # (step() has counted this update already)
            st["full_dev"] = True         # the store's device counter has taken every update so far: a valid thing to copy
            ops.adam_step_dev(pdata, grad, st["m"], st["v"], st["hyper"], st["state"])
        else:
            ops.adam_step(pdata, grad, st["m"], st["v"], group["lr"], b1, b2, group["eps"], group["weight_decay"],
                          st["step"], self.grad_scale)

    def _step_partial(self, ar, st, group):
        """Per-tensor updates ON SLICES OF THE FLAT MOMENTS for the parameters that have a gradient (torch.optim.Adam skips
        the others).  Host path: per-parameter step counts (`pstep`), back to the one-launch update once every parameter
        THAT TAKES GRADIENTS is level again.  Capturable path: the step counter is device memory, so each parameter gets a
        device state of its own -- a stream-ordered copy of the arena's when the one-launch path ran before -- and the
        arena stays on per-tensor launches from then on (replays advance those counters behind the host's back, so the
        host cannot tell when they are level)."""
        if st["pstep"] is None:
            st["pstep"] = {id(p): st["step"] for p, _, _, _ in ar.items}
        b1, b2 = group["betas"]
        if self.capturable:
            # the arena's device counter may be copied only if the one-launch path has driven it (then it also holds
            # the updates that graph replays made behind the host's back); when the very first step is already partial, there
            # is no such counter and every parameter starts from its host count
            had_dev = bool(st.get("full_dev"))
            self._ensure_dev(st, group, ar.device, st["step"])
            pdev = st.setdefault("pdev", {})
        for p, o, n, tr in ar.items:
            if p.grad is None:
                continue
            g = p.grad
            gslice = ar.grad[o:o + n]
            if g.data_ptr() != gslice.data_ptr():
                # a caller-owned gradient: bring it into the parameter's physical (tap-major) order
                gslice = torch.empty_like(p, memory_format=torch.preserve_format).copy_(g)
            st["pstep"][id(p)] += 1
            if self.capturable:
                ds = pdev.get(id(p))
                if ds is None:
                    if had_dev:
                        if torch.cuda.is_current_stream_capturing():
                            raise GdnError("capturable Adam: a per-parameter step counter would be created inside a graph capture "
                                           "(every replay would reset it); run one eager step with this gradient coverage first")
                        ds = st["state"].clone()          # the arena's device counter so far (no host read)
                    else:
                        ds = self._dev_state(group, st["pstep"][id(p)] - 1, ar.device)
                    pdev[id(p)] = ds
                ops.adam_step_dev(ar.data[o:o + n], gslice, st["m"][o:o + n], st["v"][o:o + n], st["hyper"], ds)
            else:
                ops.adam_step(ar.data[o:o + n], gslice, st["m"][o:o + n], st["v"][o:o + n], group["lr"], b1, b2, group["eps"],
                              group["weight_decay"], st["pstep"][id(p)], self.grad_scale)
        if self.capturable:
            return
        counts = {st["pstep"][id(p)] for p, _, _, _ in ar.items if p.requires_grad}     # (a frozen parameter never levels)
        if len(counts) == 1:              # everyone who steps is level again: back to the one-launch update
            lvl = counts.pop()
            if all(st["pstep"][id(p)] == lvl for p, _, _, _ in ar.items):
                st["step"], st["pstep"] = lvl, None

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
                if getattr(ar, "carry_reduced", None) is not None:
                    # (backward, sync_gradients, backward) under data parallelism leaves the first micro-batch's
                    # REDUCED gradient aside until the next sync_gradients() adds it back; stepping now would drop it silently
                    raise GdnError("optimizer.step() with an accumulated, already all-reduced gradient pending: call "
                                   "sync_gradients() after the last backward of an accumulation")
                if len(ps) != len(ar.items):
                    loose.extend(ps)      # partial coverage: fall back to per-tensor launches
                    continue
                if all(p.grad is None for p in ps):
                    continue              # nothing ran backward for this model: skip it, like torch.optim.Adam
                st = self._flat.get(id(ar))
                if st is None:
                    # ONE state store per arena: flat moments; `pstep` is None while every parameter has taken the same
                    # number of steps (st["step"]) and becomes a per-parameter count once coverage has been partial
                    st = {"m": ops.zeros((ar.numel,), ar.device), "v": ops.zeros((ar.numel,), ar.device), "step": 0,
                          "pstep": None}
                    self._flat[id(ar)] = st
                if st["pstep"] is None and all(p.grad is not None for p in ps):
                    st["step"] += 1
                    self._apply(ar.data, ar.grad, st, group, ar.device)
                else:
                    # part of the model got no gradient this step (frozen sub-modules, an unused branch): torch.optim.Adam
                    # skips those parameters -- the one-launch update would apply weight decay and stale moments to them.
                    # The others are updated per tensor ON SLICES OF THE SAME FLAT MOMENTS with their own step counts, so a
                    # parameter never alternates between two sets of moments when coverage changes between steps.
                    self._step_partial(ar, st, group)
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
                self._apply(p.data, g, st, group, p.device)
                if getattr(p, "_gdn_arena", None) is not None:
                    p._gdn_arena.touch()
        return loss
