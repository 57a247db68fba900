"""Host-side execution engine: explicit forward/backward over the HIP kernels.

The reference relies on torch autograd + cuDNN for every layer.  Here each
block's forward launches the HIP kernels directly and, when gradients are
needed, pushes one closure on a tape; ``backward`` replays the tape in reverse.
Parameters keep the reference's names/shapes but live, tap-major, in one flat
arena per model (so Adam and the RCCL all-reduce see a single buffer).
"""
import os

import torch

from . import ops
from ._lib import GdnError

_ALIGN = 64  # floats
_FUSE_EVAL_BN = os.environ.get("GDN_FUSE_EVAL_BN", "1") != "0"     # A/B switch for measurements
# fp32 stride-1 zero-padded layers with a window of at least this size run in the frequency domain
# (csrc/conv_fft.hip); 0 disables.  5x5 on 256 channels is the break-even neighbourhood.
_FFT_MIN_K = int(os.environ.get("GDN_FFT_MIN_K", "5"))
# fp32 3x3 stride-1 zero-padded layers on 64..512 channels run as Winograd F(2x2,3x3) (csrc/conv_wino.hip)
_WINOGRAD = os.environ.get("GDN_WINOGRAD", "1") != "0"
# train-mode BatchNorm fusion (A/B switch): scale/shift/ReLU of a ResidualBlock's first half applied in the consumer's
# loader, BatchNorm-backward reductions emitted by the data-gradient epilogues
_FUSE_TRAIN_BN = os.environ.get("GDN_FUSE_TRAIN_BN", "1") != "0"
# per-site A/B switches of that fusion (measurement; all on by default unless a site measured slower)
_FUSE = {k: os.environ.get("GDN_FUSE_" + k.upper(), d) != "0" for k, d in
         (("fft_in", "1"), ("fft_dyb", "1"), ("fft_bnb", "1"), ("wino_in", "1"), ("wino_bnb", "1"), ("ring_bnb", "1"))}
# fp32 4x4 stride-2 pad-1 Conv2d / ConvTranspose2d layers run as Winograd F(3x3,2x2) over the polyphase images
# (csrc/conv_wino2.hip) when both channel counts reach this value (0 disables): the transforms move ~1.8x the
# layer's activations (measured: a gain on every such layer of G, the smallest at 64 channels, tests/diag/wino2_time.py)
_WINO2_MIN_C = int(os.environ.get("GDN_WINO2_MIN_C", "64"))
# fp32 1 <-> 64 channel 9x9 layers (G's first convolution, the heads' backward) on csrc/conv_c1.hip (A/B switch)
_C1 = os.environ.get("GDN_C1", "1") != "0"
# x2 bilinear upsampling folded into the consumer convolution's loader and its backward's fold pass (north_star "bilinear-interp
# ... fused"; csrc/up2x.h): A/B switch
_FUSE_UP2X = os.environ.get("GDN_FUSE_UP2X", "1") != "0"
# bf16 form of the same fusion (round 5, row N1): LDS-DMA operands cannot be interpolated on load, so the PRODUCER writes the
# upsampled tensor -- BatchNorm-apply (+ residual) and the interpolation are one pass (gdn_bn_apply_up2x) -- and the consumer's
# reflection fold applies the adjoint (gdn_conv_dgrad dx_up2x): no stand-alone upsample2x kernel in either direction
_FUSE_UP2X_BF16 = os.environ.get("GDN_FUSE_UP2X_BF16", "1") != "0"
_GRAPH_EPOCH = 0


def bump_graph_epoch():
    """A graph replay changed parameters / BN buffers without touching torch's version counters: drop derived caches."""
    global _GRAPH_EPOCH
    _GRAPH_EPOCH += 1


# ----------------------------------------------------------------------------
# Parameter layout: logical torch shape, physical tap-major [kh*kw, Cout, Cin]
# ----------------------------------------------------------------------------
def _perm(transposed):
    # logical -> physical permutation and its inverse
    return ((2, 3, 1, 0), (3, 2, 0, 1)) if transposed else ((2, 3, 0, 1), (2, 3, 0, 1))


def tap_view(t, transposed):
    """[kh*kw, Cout, Cin] view of a conv weight (or its grad) stored tap-major; None if it is not."""
    fwd, _ = _perm(transposed)
    v = t.permute(*fwd)
    if not v.is_contiguous():
        return None
    kh, kw, co, ci = v.shape
    return v.view(kh * kw, co, ci)


class ParamArena:
    """Flat fp32 storage for all parameters of a model (+ a matching gradient arena)."""

    def __init__(self, module, device):
        self.items = []      # (param, offset, numel, transposed or None)
        convt = {id(m.weight) for m in module.modules() if isinstance(m, torch.nn.ConvTranspose2d)}
        off = 0
        for p in module.parameters():
            tr = None
            if p.dim() == 4:
                tr = id(p) in convt
            self.items.append((p, off, p.numel(), tr))
            off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = off
        self.device = device
        self.data = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=device)
        for p, o, n, tr in self.items:
            src = p.data.to(device=device, dtype=torch.float32)
            p.data = self._view(self.data, o, src.shape, tr)
            p.data.copy_(src)
            p._gdn_arena = self
        self.ptr0 = self.items[0][0].data_ptr() if self.items else 0
        self.generation = 0
        self.data16, self._key16, self._v16 = None, None, {}
        # data parallelism (distributed.py): `reduced` -- the gradient arena holds sums over the ranks (an all-reduce ran
        # since the last fresh backward) rather than this rank's local gradients; `carry_reduced` -- an already reduced
        # gradient carried across a further backward (accumulation without zero_grad), kept OUT of the arena until the new
        # local contribution has been reduced too (sync_gradients adds it back), so nothing is summed over the ranks twice
        self.reduced = False
        self.reduced_scale = 1.0       # 1: the reduced arena holds sums over the ranks; 1/world: means (sync_gradients without an optimizer)
        self.carry_reduced = None      # always kept as a SUM over the ranks

    @staticmethod
    def _view(flat, off, shape, tr):
        n = 1
        for s in shape:
            n *= s
        sl = flat[off:off + n]
        if tr is None:
            return sl.view(shape)
        fwd, inv = _perm(tr)
        phys = [shape[i] for i in fwd]
        return sl.view(phys).permute(*inv)

    def touch(self):
        """Called by writers that bypass torch (the fused Adam kernel): parameters changed."""
        self.generation += 1

    def versions(self):
        """Sum of the parameters' own torch version counters: `p.data = view` gives every parameter a counter of its own,
        so load_state_dict / p.copy_() / a stock torch optimizer bump THESE and not the arena tensor's."""
        return sum(p._version for p, _, _, _ in self.items)

    def bf16_data(self):
        """bf16 shadow of the parameter arena (same offsets/layout), re-cast when the master changed."""
        key = (self.generation, self.data._version, self.versions(), _GRAPH_EPOCH)
        if self.data16 is None:
            self.data16 = torch.empty(self.numel, dtype=torch.bfloat16, device=self.device)
            self._key16 = None
        if self._key16 != key or torch.cuda.is_current_stream_capturing():
            ops.cast(self.data, out=self.data16)
            self._key16 = key
        return self.data16

    def bf16_view(self, p):
        v = self._v16.get(id(p))
        if v is None:
            for q, o, n, tr in self.items:
                if q is p:
                    v = self._view(self.bf16_data(), o, p.shape, tr)
                    self._v16[id(p)] = v
                    break
            else:
                raise KeyError("parameter not in arena")
        return v

    def intact(self):
        return all(p.data_ptr() == self.data.data_ptr() + 4 * o and p.device == self.data.device
                   for p, o, _, _ in self.items)

    def grad_view(self, p):
        for q, o, n, tr in self.items:
            if q is p:
                return self._view(self.grad, o, p.shape, tr)
        raise KeyError("parameter not in arena")

    def bind_grads(self):
        """Point every .grad at its arena slice for one backward.  Returns the params whose existing, caller-owned grad
        must be accumulated afterwards.  A .grad that already IS the arena slice (a second backward without zero_grad:
        gradient accumulation) is carried over: the kernels overwrite their slices, finish_grads() adds the carry back."""
        self._gv = {}
        self._written = set()
        self._bound_before = set()
        self._carry = None
        accumulate = []
        for p, o, n, tr in self.items:
            gv = self._view(self.grad, o, p.shape, tr)
            if p.grad is not None:
                if p.grad.data_ptr() == gv.data_ptr():
                    self._bound_before.add(id(p))
                else:
                    accumulate.append((p, p.grad))
            p.grad = gv
            self._gv[id(p)] = gv
        if self._bound_before:
            self._carry = self.grad.clone()
            for p, o, n, tr in self.items:
                if id(p) not in self._bound_before:
                    self._carry[o:o + n].zero_()
        else:
            self.reduced = False               # a fresh backward overwrites the arena with this rank's local gradients
            self.carry_reduced = None
        return accumulate

    def mark_written(self, params):
        self._written.update(id(p) for p in params)

    def finish_grads(self):
        """After the tape ran: parameters no kernel wrote a gradient for (requires_grad False, an unused branch) get the
        state autograd would leave -- their previous gradient, or None -- instead of whatever an earlier step left in the
        arena; a carried gradient (see bind_grads) is added back."""
        for p, o, n, tr in self.items:
            if id(p) in self._written:
                continue
            if id(p) in self._bound_before:
                self.grad[o:o + n].zero_()            # the carry holds the old value
            else:
                p.grad = None
        if self._carry is not None:
            if self.reduced:
                # the carry is a sum over the ranks, what the tape just wrote is local: they meet after the next all-reduce
                if self.reduced_scale != 1.0:
                    self._carry.mul_(1.0 / self.reduced_scale)
                self.carry_reduced = self._carry if self.carry_reduced is None else self.carry_reduced.add_(self._carry)
                self.reduced = False
            else:
                self.grad.add_(self._carry)
            self._carry = None


def begin_backward(module, arena, ctx, trainable=True):
    """Gradient bookkeeping before a model's tape replays (the bridge's backward; also driven directly by the CPU tests of
    the data-parallel accumulation semantics).  Returns the caller-owned gradients to accumulate afterwards."""
    red = getattr(module, "_gdn_reducer", None)
    mine = red is not None and red.arena is arena
    if mine and red.active:
        # a previous backward's overlapped all-reduces are still in flight (backward, backward, ... without a
        # sync_gradients in between): wait for them before bind_grads() reads the arena; it then holds reduced sums
        red.finish()
        arena.reduced, arena.reduced_scale = True, 1.0
    pending = arena.bind_grads() if trainable else []      # a frozen network (the guide) only passes dx through
    # the overlapped reducer hands buckets to async all-reduces while the tape is still running; a carried gradient
    # (backward without zero_grad: finish_grads adds it AFTER the tape) or a caller-owned .grad would be added behind
    # those reductions' backs -- such a backward leaves its local gradients in the arena and sync_gradients reduces the
    # whole arena afterwards (an already REDUCED carry is kept aside until then: ParamArena.carry_reduced)
    if mine and trainable and not pending and not arena._bound_before:
        red.begin()
        ctx.reducer = red
    return pending


def end_backward(arena, pending, trainable=True):
    if trainable:
        arena.finish_grads()
    for p, old in pending:       # a caller-owned .grad existed: accumulate like autograd would
        if p.grad is None:
            p.grad = old
        else:
            p.grad.add_(old)


def ensure_arena(module, device):
    ar = getattr(module, "_gdn_param_arena", None)
    if ar is None or ar.device != device or not ar.intact():
        ar = ParamArena(module, device)
        module._gdn_param_arena = ar
    return ar


# ----------------------------------------------------------------------------
# Tape
# ----------------------------------------------------------------------------
class Ctx:
    def __init__(self, record, arena=None, input_needs_grad=False, dtype=torch.float32):
        self.record = record
        self.dtype = dtype                     # storage dtype of activations / activation gradients
        self.tape = []
        self.arena = arena
        self.grads = {}
        self.input = None                      # the model's NHWC input buffer
        self.input_needs_grad = input_needs_grad
        self.reducer = None                    # distributed.GradReducer while a backward is running
        self.bn_src = {}                       # id(activation) -> BnOut of the train-mode BatchNorm that produced it
        self.up_src = {}                       # id(upsampled tensor) -> (its low-resolution source, up2x mode): conv_bn_act(up_out=)

    def grads_done(self, *params):
        """Parameters whose gradient has just been written (lets the all-reduce start early)."""
        if self.arena is not None:
            self.arena.mark_written(params)
        if self.reducer is not None:
            self.reducer.mark(params)

    def claim(self, x):
        """Called by whatever consumes activation x FIRST in the forward.  That consumer's backward runs after every other
        consumer's, so its data gradient (with the accumulated `addsrc`) is the final value of dL/dx: if x came out of a
        train-mode BatchNorm, the kernel writing it can also emit that BatchNorm's backward reduction (BnOut.partial).
        Returns the BnOut (once), or None."""
        if isinstance(x, BnOut):
            return x if x.claim() else None
        info = self.bn_src.pop(id(x), None)
        return info if info is not None and info.claim() else None

    def wants_dx(self, x):
        """Data gradients stop at the network input unless the caller asked for them."""
        return x is not self.input or self.input_needs_grad

    # gradient slots are keyed by tensor identity
    def add_grad(self, t, g):
        k = id(t)
        old = self.grads.get(k)
        if old is None:
            self.grads[k] = (t, g)
        else:
            self.grads[k] = (t, _add(old[1], g, t.dtype))

    def pop_grad_as(self, t, dtype):
        """pop_grad, converted to `dtype` (the head's fp32 input gradient meeting a bf16 consumer)."""
        g = self.pop_grad(t)
        if g is not None and g.dtype != dtype:
            g = ops.cast(_dense(g), dtype)
        return g

    def pop_grad(self, t):
        e = self.grads.pop(id(t), None)
        return None if e is None else e[1]

    def backward(self):
        if self.tape is None:
            raise GdnError("this forward's tape was already consumed (retain_graph is not supported on the HIP path)")
        for fn in reversed(self.tape):
            fn()
        self.tape = None      # drop the saved activations


def _chan_slice(t):
    """True for a pure channel slice of a dense NHWC buffer: channel stride 1, one pixel pitch, rows and images laid out
    back to back at that pitch -- the only non-contiguous shape gdn_add_pitched addresses correctly (a spatial crop or a
    strided batch view has other row / image strides and takes the generic copy)."""
    return (t.dim() == 4 and t.shape[3] % 4 == 0 and t.stride(3) == 1 and t.stride(1) == t.shape[2] * t.stride(2)
            and t.stride(0) == t.shape[1] * t.stride(1))


def _dense(t):
    """Dense copy of a channel-slice view (the gradient of a torch.cat half); contiguous tensors pass through."""
    if t.is_contiguous():
        return t
    if _chan_slice(t):
        return ops.add_pitched(t)
    out = torch.empty(t.shape, dtype=t.dtype, device=t.device)
    out.copy_(t)
    return out


def _add(a, b, out_dtype):
    if a.is_contiguous() and b.is_contiguous():
        return ops.add(a, b, out_dtype=out_dtype)
    if _chan_slice(a) and _chan_slice(b):
        return ops.add_pitched(a, b, out_dtype=out_dtype)
    return ops.add(_dense(a), _dense(b), out_dtype=out_dtype)


class BnOut:
    """What a train-mode conv + BatchNorm (+ReLU) layer leaves behind for its backward: the raw conv output y, the
    coefficients co = [scale, shift, mean, invstd] and, once the kernel that writes the final dL/d(output) has run, that
    kernel's per-slot partial sums of the BatchNorm backward reduction (`partial`, or None: run the reduce pass).

    It doubles as the DEFERRED activation of a ResidualBlock's first half: `a = relu(bn1(conv1 x))` has exactly one
    consumer (conv2), whose patch loader applies scale / shift / ReLU on the fly, so `a` is never written to memory
    (AE_model_unet.py:49-54; north_star "conv+BN+ReLU fused").  A consumer that cannot do that calls dense()."""

    def __init__(self, y, co, relu):
        self.y, self.co, self.relu = y, co, relu
        self.partial = None
        self._claimed = False
        self._dense = None
        self.shape, self.dtype = y.shape, y.dtype

    def claim(self):
        first, self._claimed = not self._claimed, True
        return first

    def dense(self, out_dtype=None):
        if self._dense is None:
            self._dense = ops.bn_apply(self.y, self.co[0], self.co[1], self.relu, None, out_dtype=out_dtype)
        return self._dense


class Up2x:
    """Deferred x2 bilinear upsampling of `src` (F.interpolate(scale_factor=2, mode='bilinear'), AE_model_unet.py:336-359):
    the one consumer -- a ConvBlock -- interpolates while its transform kernel gathers the patches and its backward's fold
    pass applies the adjoint, so the upsampled tensor (4x src) and its gradient are never written.  A consumer whose path
    has no such loader calls dense(ctx): the stand-alone kernels, with their tape entry."""

    def __init__(self, src, align_corners):
        self.src, self.align = src, bool(align_corners)
        B, H, W, C = src.shape
        self.shape, self.dtype = torch.Size((B, 2 * H, 2 * W, C)), src.dtype
        self.mode = 2 if align_corners else 1          # in_up2x / dx_up2x of the C ABI

    def dense(self, ctx):
        x, align = self.src, self.align
        y = ops.upsample2x(x, align)
        if ctx.record:
            def bwd():
                dy = ctx.pop_grad(y)
                if dy is None:
                    return
                ctx.add_grad(x, ops.upsample2x_bwd(_dense(dy), align))
            ctx.tape.append(bwd)
        return y


def _conv_op(mod, reflect):
    op = getattr(mod, "_gdn_op", None)
    if op is None:
        tr = isinstance(mod, torch.nn.ConvTranspose2d)
        op = ops.Conv(mod.in_channels, mod.out_channels, mod.kernel_size[0], mod.stride[0],
                      mod.padding[0] if not reflect else reflect, reflect=bool(reflect), transposed=tr)
        mod._gdn_op = op
    return op


def _w_tap(mod):
    tr = isinstance(mod, torch.nn.ConvTranspose2d)
    v = tap_view(mod.weight.data, tr)
    if v is None:
        raise GdnError("conv weight is not in tap-major layout; call the model once on its device first")
    return v, tr


def _w_for(ctx, mod, dtype):
    """Tap-major weight in the layer's compute dtype (bf16: view into the arena's bf16 shadow)."""
    w, tr = _w_tap(mod)
    if dtype == torch.float32:
        return w, tr
    if ctx.arena is None:
        raise GdnError("bf16 compute needs the parameter arena")
    ctx.arena.bf16_data()                       # refresh the shadow if the master changed
    v = tap_view(ctx.arena.bf16_view(mod.weight), tr)
    return v, tr


def _layer_dtype(ctx, conv):
    """bf16 layers need 64-channel reduction slabs; the image-input conv stays fp32."""
    if ctx.dtype == torch.bfloat16 and conv.in_channels % 64 == 0 and conv.out_channels % 64 == 0:
        return torch.bfloat16
    return torch.float32


def _wgrad_into(ctx, mod, x, dy, x2=None):
    """Weight gradient of a conv module straight into its arena slice."""
    op = mod._gdn_op
    tr = isinstance(mod, torch.nn.ConvTranspose2d)
    gv = tap_view(mod.weight.grad, tr)
    if gv is None:
        raise GdnError("weight.grad is not tap-major")
    op.wgrad(x, dy, gv, 0)
    if x2 is not None:
        op.wgrad(x2, dy, gv, x.shape[3])


def _eval_coeffs(bn):
    """scale/shift of an eval-mode BatchNorm, cached until its tensors change."""
    ar = getattr(bn.weight, "_gdn_arena", None)       # the fused Adam writes gamma / beta behind torch's version counters
    key = (_GRAPH_EPOCH, getattr(bn, "_gdn_stats_ver", 0), bn.running_mean._version, bn.running_var._version,
           bn.weight._version, bn.bias._version, bn.running_mean.data_ptr(), bn.weight.data_ptr(),
           None if ar is None else ar.generation)
    c = getattr(bn, "_gdn_eval_cache", None)
    if c is None or c[0] != key:
        co = ops.bn_eval_coeffs(bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var, bn.eps)
        c = (key, co)
        bn._gdn_eval_cache = c
    return c[1]


def _conv_instnorm_train(ctx, x, conv, inorm, relu, op, w, reflect, need_dx):
    """Conv -> train-mode InstanceNorm2d(affine=True, track_running_stats=True) -> [ReLU]: the standalone
    ConvBlock / ConvTBlock(norm='Instance') of the reference (AE_model_unet.py:70-75, :88-92; R and G never build one).
    Per-instance statistics are batch statistics of a batch of one, so the BatchNorm kernels run once per image on that
    image's slice: the convolution writes image b's output with its sum / sum-of-squares partials, finalize (momentum 1 into
    scratch buffers) yields that image's coefficients, mean and unbiased variance, apply / backward use them; the running
    statistics take the batch mean of the per-instance values (what F.instance_norm does).  A rarely used path: direct
    kernels, B small launches."""
    B, H, W, _ = x.shape
    C = conv.out_channels
    mom = 0.1 if inorm.momentum is None else inorm.momentum
    ys, cos, outs = [], [], []
    rm = torch.zeros((B, C), dtype=torch.float32, device=x.device)
    rv = torch.ones((B, C), dtype=torch.float32, device=x.device)
    for b in range(B):
        yb, st = op.fwd(x[b:b + 1], w, stats=True)
        co = ops.bn_finalize_train(st, yb.shape[1] * yb.shape[2], inorm.weight.data, inorm.bias.data, rm[b], rv[b], 1.0, inorm.eps)
        ys.append(yb)
        cos.append(co)
        outs.append(ops.bn_apply(yb, co[0], co[1], relu, None, out_dtype=ctx.dtype))
    if inorm.track_running_stats:
        inorm.running_mean.mul_(1.0 - mom).add_(rm.mean(0), alpha=mom)
        inorm.running_var.mul_(1.0 - mom).add_(rv.mean(0), alpha=mom)
        # (torch's InstanceNorm2d leaves num_batches_tracked untouched: F.instance_norm does not take it)
        inorm._gdn_stats_ver = getattr(inorm, "_gdn_stats_ver", 0) + 1
    a = torch.cat(outs, 0)
    if ctx.record:
        in_hw = (H, W)

        def bwd():
            da = ctx.pop_grad(a)
            if da is None:
                return
            da = _dense(da)
            frozen = not (conv.weight.requires_grad or inorm.weight.requires_grad or inorm.bias.requires_grad)
            dg = torch.zeros(C, dtype=torch.float32, device=da.device)
            db = torch.zeros(C, dtype=torch.float32, device=da.device)
            dys = []
            for b in range(B):
                dgb, dbb = torch.empty_like(dg), torch.empty_like(db)
                dys.append(ops.bn_bwd(da[b:b + 1], ys[b], inorm.weight.data, cos[b], relu, dgb, dbb))
                dg += dgb
                db += dbb
            dy = torch.cat(dys, 0)
            if not frozen:
                inorm.weight.grad.copy_(dg)
                inorm.bias.grad.copy_(db)
                _wgrad_into(ctx, conv, x, dy)
                ctx.grads_done(inorm.weight, inorm.bias, conv.weight)
            if need_dx and ctx.wants_dx(x):
                wt = ops.transpose_taps(_w_tap(conv)[0], dtype=torch.float32)
                ctx.grads[id(x)] = (x, op.dgrad(dy, wt, in_hw, addsrc=ctx.pop_grad_as(x, torch.float32)))
        ctx.tape.append(bwd)
    return a


def conv_bn_act(ctx, x, conv, bn, relu, residual=None, x2=None, reflect=0, need_dx=True, defer=False, up_out=None):
    """[relu](BN(conv(cat(x, x2)))) (+ residual): ConvBlock / ResidualBlock halves / ConvTBlock.

    up_out (None, or the align_corners flag): the caller also wants F.interpolate(output, scale_factor=2, mode='bilinear') and
    gets (output, upsampled) back -- on the bf16 path the BatchNorm-apply pass writes both (one kernel), on the fp32 path the
    upsampled one is the deferred Up2x of upsample().

    x may be a BnOut (the deferred activation of the previous layer): the transform-domain paths apply its scale / shift /
    ReLU while loading; any other path materialises it first.  defer=True (the caller guarantees a single consumer)
    returns this layer's output as a BnOut instead of running the BatchNorm-apply pass, when the layer is a train-mode
    fp32 one without a residual."""
    op = _conv_op(conv, reflect)
    ldt = _layer_dtype(ctx, conv)
    up = None
    if isinstance(x, Up2x):
        # deferred x2 upsampling: the frequency-domain / Winograd F(2x2,3x3) loaders interpolate on the fly; training needs
        # the fold pass of a reflection-padded layer for the adjoint.  Everything else gets the materialised tensor.
        k, s_ = conv.kernel_size[0], conv.stride[0]
        B_, H_, W_ = x.shape[0], x.shape[1], x.shape[2]
        fusable = (ldt == torch.float32 and x2 is None and s_ == 1 and (not ctx.record or reflect)
                   and ((_FFT_MIN_K > 0 and k >= _FFT_MIN_K and op.fft_ok(B_, H_, W_, backward=ctx.record))
                        or (_WINOGRAD and k == 3 and op.wino_ok(B_, H_, W_))))
        if fusable:
            up = x
        else:
            x = x.dense(ctx)
    xin = ctx.claim(x)                   # BnOut of the train-mode BatchNorm that produced x, if we are its first consumer
    lazy = isinstance(x, BnOut)
    if x2 is not None:
        ctx.claim(x2)
    if x.dtype != ldt:
        raise GdnError("layer %d->%d computes in %s but its input is %s" % (conv.in_channels, conv.out_channels, ldt, x.dtype))
    w, tr = _w_for(ctx, conv, ldt)
    if bn.training and isinstance(bn, torch.nn.InstanceNorm2d):
        if lazy or up is not None or x2 is not None or residual is not None or ldt != torch.float32:
            raise GdnError("train-mode InstanceNorm is implemented for the standalone fp32 ConvBlock / ConvTBlock only")
        a = _conv_instnorm_train(ctx, x, conv, bn, relu, op, w, reflect, need_dx)
        return a if up_out is None else (a, upsample(ctx, a, bool(up_out)))
    # GDN_HINT_TRAIN: a trained layer in train mode (forward + backward + weight gradient) -- the frequency-domain path then
    # tiles for the sum of both passes (40-point tiles on the 9x9 layers); frozen / eval-mode layers keep the forward-optimal plan
    fft_train = bool(ctx.record and bn.training and conv.weight.requires_grad)
    use_fft = (_FFT_MIN_K > 0 and ldt == torch.float32 and x2 is None and conv.kernel_size[0] >= _FFT_MIN_K
               and conv.stride[0] == 1
               and op.fft_ok(x.shape[0], x.shape[1], x.shape[2], backward=ctx.record, train=fft_train))
    if use_fft and ctx.dtype == torch.bfloat16:
        # A model that computes in bf16 runs barrier-paced 16-bit matrix kernels (conv_*_bf16) and must never have a
        # frequency-domain kernel in flight beside them (measured interference on this hardware; the
        # frequency-domain backward uses a second stream).  No layer of a bf16 model qualifies today (ldt is fp32 only for the
        # 3-channel input conv); this keeps it that way by construction (tools/check_no_mfma16_beside_fft.py checks traces).
        raise GdnError("internal: a bf16 model selected a frequency-domain layer")
    use_wino = (not use_fft and _WINOGRAD and ldt == torch.float32 and x2 is None
                and conv.kernel_size[0] == 3 and conv.stride[0] == 1 and op.wino_ok(x.shape[0], x.shape[1], x.shape[2]))
    use_wino2 = (not use_fft and not use_wino and _WINO2_MIN_C > 0 and ldt == torch.float32 and x2 is None
                 and conv.kernel_size[0] == 4 and conv.stride[0] == 2 and not isinstance(x, BnOut)
                 and min(conv.in_channels, conv.out_channels) >= _WINO2_MIN_C and op.wino2_ok(x.shape[0], x.shape[1], x.shape[2]))
    # the transform-domain paths share one call shape: forward (+ saved state), backward from that state
    alt_fwd = op.fft_fwd if use_fft else op.wino_fwd if use_wino else op.wino2_fwd if use_wino2 else None
    alt_bwd = op.fft_bwd if use_fft else op.wino_bwd if use_wino else op.wino2_bwd if use_wino2 else None
    # slots of the producer BatchNorm's backward partials the data-gradient pass can emit (wino2 epilogues do not emit them)
    bnb_slots = (op.wino_bnb_slots if use_wino
                 else (lambda B_, H_, W_: op.fft_bnb_slots(B_, H_, W_, train=fft_train)) if (use_fft and not use_wino2 and _FUSE["fft_bnb"])
                 else (lambda *a: 0))
    state_kw = "spectrum" if use_fft else "state"
    bstate_kw = "xf" if use_fft else "state"
    use_fft_only = use_fft
    use_fft = use_fft or use_wino or use_wino2
    in_kw = {}
    xt = x                               # the tensor the conv kernels read
    hint_kw = {"train": fft_train} if use_fft_only else {}
    if up is not None:
        if not (use_fft and not use_wino2):
            raise GdnError("internal: deferred upsampling reached a layer without a fused loader")
        in_kw = dict(up2x=up.mode)
        xt = up.src
    if lazy:
        if use_fft and not use_wino2 and _FUSE_TRAIN_BN and _FUSE["fft_in" if use_fft_only else "wino_in"]:
            in_kw = dict(in_affine=(x.co[0], x.co[1]), in_relu=x.relu)
            xt = x.y
        else:
            xt = x.dense(ctx.dtype)      # (the direct kernels' loaders go straight to LDS: materialise)
    xf = None
    keep_xf = use_fft and ctx.record and conv.weight.requires_grad
    use_c1 = (_C1 and not use_fft and ldt == torch.float32 and x2 is None and not lazy and conv.in_channels in (1, 3)
              and not isinstance(conv, torch.nn.ConvTranspose2d)
              and ops.c1_ok(x, conv.out_channels, conv.kernel_size[0], conv.stride[0], reflect or conv.padding[0], rgb=True))
    if bn.training:
        if use_fft:
            r = alt_fwd(xt, w, stats=True, **{state_kw: keep_xf}, **in_kw, **hint_kw)
            y, st = r[0], r[1]
            xf = r[2] if keep_xf else None
        elif use_c1:
            y, st = ops.conv_c1_fwd(xt, w, reflect=bool(reflect), stats=True)
        else:
            y, st = op.fwd(xt, w, x2=x2, stats=True)
        count = y.shape[0] * y.shape[1] * y.shape[2]
        mom = 0.1 if bn.momentum is None else bn.momentum
        co = ops.bn_finalize_train(st, count, bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var, mom, bn.eps,
                                   num_batches_tracked=bn.num_batches_tracked)
        bn._gdn_stats_ver = getattr(bn, "_gdn_stats_ver", 0) + 1
    else:
        co = _eval_coeffs(bn)
    a_up = None
    fused = (_FUSE_EVAL_BN and not bn.training and ldt == ctx.dtype and not (relu and residual is not None)
             and (residual is None or residual.dtype == ldt) and conv.out_channels > 1)
    out_info = None
    if fused:
        # eval-mode BN folded into the conv epilogue: conv + scale/shift + ReLU (+ residual) in one pass
        if use_fft:
            y = a = alt_fwd(xt, w, affine=(co[0], co[1]), act=ops.ACT_RELU if relu else ops.ACT_NONE, addsrc=residual, **in_kw,
                            **hint_kw)
        elif use_c1:
            y = a = ops.conv_c1_fwd(xt, w, reflect=bool(reflect), affine=(co[0], co[1]),
                                    act=ops.ACT_RELU if relu else ops.ACT_NONE, addsrc=residual)
        else:
            y = a = op.fwd(xt, w, x2=x2, affine=(co[0], co[1]), act=ops.ACT_RELU if relu else ops.ACT_NONE, addsrc=residual)
    else:
        if not bn.training:
            y = (alt_fwd(xt, w, **in_kw, **hint_kw) if use_fft else ops.conv_c1_fwd(xt, w, reflect=bool(reflect)) if use_c1
                 else op.fwd(xt, w, x2=x2))
        else:
            out_info = BnOut(y, co, relu)
        if (defer and out_info is not None and residual is None and _FUSE_TRAIN_BN and ldt == torch.float32
                and ctx.dtype == torch.float32):
            a = out_info                                  # scale / shift / ReLU happen in the consumer's loader
        else:
            if (up_out is not None and _FUSE_UP2X_BF16 and ctx.dtype == torch.bfloat16 and y.is_contiguous()
                    and (residual is None or residual.is_contiguous()) and y.shape[0] * 2 * y.shape[1] <= 65535
                    and y.shape[3] % 4 == 0):         # (the fused kernel's channel-vector width; other shapes take the two-pass form)
                a, a_up = ops.bn_apply_up2x(y, co[0], co[1], relu, residual, align_corners=bool(up_out), out_dtype=ctx.dtype)
            else:
                a = ops.bn_apply(y, co[0], co[1], relu, residual, out_dtype=ctx.dtype)
            if out_info is not None and ctx.record and _FUSE_TRAIN_BN:
                ctx.bn_src[id(a)] = out_info
    # the layer's input is an upsampled tensor whose producer kept the low-resolution source (up_out above): a reflection-padded
    # layer's fold pass can write dL/d(source) directly
    up_in = None
    if (not isinstance(x, (BnOut, Up2x)) and x2 is None and reflect and not use_fft and conv.in_channels % 4 == 0
            and x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0):
        up_in = ctx.up_src.get(id(x))
    if ctx.record:
        in_hw = (x.shape[1], x.shape[2])
        bn_training = bn.training

        def bwd():
            da = ctx.pop_grad(a)
            if da is None:
                return
            frozen = not (conv.weight.requires_grad or bn.weight.requires_grad or bn.bias.requires_grad)
            if not bn_training and not frozen:
                raise GdnError("backward through an eval-mode BatchNorm is only implemented for frozen layers "
                               "(requires_grad False on the conv and BN parameters): the guide network of --latent_grad")
            if residual is not None:
                ctx.add_grad(residual, da)
            dyb = None
            if (bn_training and use_fft_only and _FUSE_TRAIN_BN and _FUSE["fft_dyb"] and da.is_contiguous()
                    and da.dtype == torch.float32):
                # frequency-domain layer: dy has one reader (the dy transform), which applies pass 3 of the BatchNorm
                # backward while loading -- dy is never written
                kk = ops.bn_bwd_coeffs(da, y, co, relu, bn.weight.grad if not frozen else None,
                                       bn.bias.grad if not frozen else None, partial=out_info.partial)
                out_info.partial = None
                dy, dyb = da, (y, co, kk, relu)
            elif bn_training:
                dy = ops.bn_bwd(da, y, bn.weight.data, co, relu, bn.weight.grad if not frozen else None,
                                bn.bias.grad if not frozen else None, out_dtype=ldt, partial=out_info.partial)
                out_info.partial = None
            else:
                dy = ops.bn_eval_bwd(da, y, co, (2 if fused else 1) if relu else 0, out_dtype=ldt)
            want_dx = need_dx and ctx.wants_dx(x)
            if use_fft:
                # one transform of dy feeds both gradients; the forward's input spectrum is reused for dw
                gv = None
                if not frozen:
                    gv = tap_view(conv.weight.grad, isinstance(conv, torch.nn.ConvTranspose2d))
                    if gv is None:
                        raise GdnError("weight.grad is not tap-major")
                if gv is not None or want_dx:
                    bnb = None
                    if (xin is not None and want_dx and xin.y.dtype == torch.float32
                            and _FUSE["fft_bnb" if use_fft_only else "wino_bnb"]):
                        # x = [relu](BN_train(xin.y)) and this data gradient is its final gradient: emit the producer's
                        # BatchNorm-backward partial sums from the epilogue that writes dx
                        slots = bnb_slots(x.shape[0], x.shape[1], x.shape[2])
                        if slots > 0:
                            part = torch.empty((slots, 2, conv.in_channels), dtype=torch.float32, device=dy.device)
                            bnb = (xin.y, xin.co, xin.relu, part)
                    extra = {"dyb": dyb} if dyb is not None else {}
                    if bnb is not None:
                        extra["bnb"] = bnb
                    gx = x                                  # the tensor whose gradient this layer's dx is
                    if up is not None:
                        extra["up2x"] = up.mode             # ... the low-resolution source of the deferred upsampling
                        gx = up.src
                    dx = alt_bwd(dy, w, in_hw, dw_tap=gv, need_dx=want_dx, **{bstate_kw: xf},
                                 addsrc=ctx.pop_grad_as(gx, ldt) if want_dx else None, **extra, **hint_kw)
                    if want_dx:
                        ctx.grads[id(gx)] = (gx, dx)
                        if bnb is not None:
                            xin.partial = bnb[3]
                if not frozen:
                    ctx.grads_done(bn.weight, bn.bias, conv.weight)
                return
            if not frozen:
                if use_c1 and dy.is_contiguous() and conv.in_channels == 1:
                    ops.conv_c1_wgrad(xt, dy, tap_view(conv.weight.grad, False), reflect=bool(reflect))
                else:
                    _wgrad_into(ctx, conv, xt, dy, x2)
                ctx.grads_done(bn.weight, bn.bias, conv.weight)
            if want_dx:
                wt = ops.transpose_taps(_w_tap(conv)[0], dtype=ldt)
                if x2 is None:
                    bnb = None
                    if (xin is not None and ldt == torch.bfloat16 and xin.y.dtype == ldt and _FUSE_TRAIN_BN and _FUSE["ring_bnb"]
                            and xin.y.is_contiguous()):
                        # x = [relu](BN_train(xin.y)) and this data gradient is its final gradient (we are its first consumer;
                        # the other consumers' gradients arrive as addsrc): the LDS-DMA ring kernel's epilogue emits the
                        # producer's BatchNorm-backward partial sums, as the Winograd path does for the fp32 layers
                        slots = op.dgrad_bnb_slots(x.shape[0], x.shape[1], x.shape[2], ldt)
                        if slots > 0:
                            part = torch.empty((slots, 2, conv.in_channels), dtype=torch.float32, device=dy.device)
                            bnb = (xin.y, xin.co, xin.relu, part)
                    if up_in is not None and bnb is None and id(x) not in ctx.grads:
                        lo, mode = up_in
                        ctx.grads[id(lo)] = (lo, op.dgrad(dy, wt, in_hw, addsrc=ctx.pop_grad_as(lo, ldt), up2x=mode))
                        return
                    dx = op.dgrad(dy, wt, in_hw, addsrc=ctx.pop_grad_as(x, ldt), bnb=bnb)
                    ctx.grads[id(x)] = (x, dx)
                    if bnb is not None:
                        xin.partial = bnb[3]
                else:
                    dcat = op.dgrad(dy, wt, in_hw)
                    c1 = x.shape[3]
                    ctx.add_grad(x, dcat[..., :c1])
                    ctx.add_grad(x2, dcat[..., c1:])
        ctx.tape.append(bwd)
    if up_out is None:
        return a
    if a_up is None:
        return a, upsample(ctx, a, bool(up_out))
    ctx.claim(a)                         # (as upsample() does: the interpolation is this activation's first consumer)
    if ctx.record:
        align = bool(up_out)
        ctx.up_src[id(a_up)] = (a, 2 if align else 1)

        def up_bwd():                    # (after this tape entry in the forward = before the layer's own backward)
            dup = ctx.pop_grad(a_up)     # None when the consumer's fold pass already wrote dL/da (the usual case)
            if dup is not None:
                ctx.add_grad(a, ops.upsample2x_bwd(_dense(dup), align))
        ctx.tape.append(up_bwd)
    return a, a_up


def conv_head_tanh(ctx, x, conv):
    """Final 9x9 (transposed) conv to one channel + tanh (AE_model_unet.py:362-363, :570-571).
    On the bf16 path only x is bf16: weights, the depth map and this layer's backward results are fp32 (the 1 <-> 64 channel
    kernels read the bf16 x / the bf16 gradient already accumulated for x directly)."""
    op = _conv_op(conv, 0)
    ctx.claim(x)
    w, tr = _w_tap(conv)
    out = op.fwd(x, w, act=ops.ACT_TANH)
    if ctx.record:
        in_hw = (x.shape[1], x.shape[2])

        def bwd():
            do = ctx.pop_grad(out)
            if do is None:
                return
            dpre = ops.tanh_bwd(do.contiguous(), out)
            # 64 -> 1 heads: both gradients are 1 <-> 64 channel correlations with the single-channel d(pre-tanh) as the
            # staged image (csrc/conv_c1.hip); a Conv2d head flips the taps, a ConvTranspose2d head does not
            c1 = (_C1 and x.is_contiguous() and conv.out_channels == 1
                  and ops.c1_ok(dpre, conv.in_channels, conv.kernel_size[0], conv.stride[0], conv.padding[0]))
            if conv.weight.requires_grad:
                if c1:
                    ops.conv_c1_wgrad(dpre, x, tap_view(conv.weight.grad, tr), flip=not tr)
                else:
                    _wgrad_into(ctx, conv, x, dpre)
                ctx.grads_done(conv.weight)
            if c1:
                dx = ops.conv_c1_fwd(dpre, w, flip=not tr, addsrc=ctx.pop_grad(x))      # fp32, like the generic path's
            else:
                wt = ops.transpose_taps(w)
                dx = op.dgrad(dpre, wt, in_hw, addsrc=ctx.pop_grad_as(x, torch.float32))
            ctx.grads[id(x)] = (x, dx)
        ctx.tape.append(bwd)
    return out


def conv_plain(ctx, x, conv, x2=None):
    """Bare convolution without norm/activation (legacy AutoEncoder 1x1 after cat, :210)."""
    op = _conv_op(conv, 0)
    ctx.claim(x)
    w, tr = _w_for(ctx, conv, _layer_dtype(ctx, conv))
    y = op.fwd(x, w, x2=x2)
    if ctx.record:
        raise GdnError("legacy AutoEncoder is inference-only on the HIP path")
    return y


def upsample(ctx, x, align_corners=False):
    """F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=...).  fp32: returned DEFERRED (Up2x) -- the consumer
    ConvBlock's loader interpolates; see conv_bn_act."""
    ctx.claim(x)
    if isinstance(x, BnOut):
        x = x.dense(ctx.dtype)
    if _FUSE_UP2X and ctx.dtype == torch.float32 and x.dtype == torch.float32 and x.is_contiguous():
        return Up2x(x, align_corners)
    return Up2x(x, align_corners).dense(ctx)


# ----------------------------------------------------------------------------
# Boundary: NCHW torch tensors <-> NHWC device buffers
# ----------------------------------------------------------------------------
def to_nhwc(x):
    if x.dim() != 4:
        raise GdnError("expected a 4-D NCHW tensor")
    if not x.is_cuda:
        raise GdnError("input must live on the GPU: the HIP path has no CPU fallback")
    x = x.detach()
    if x.dtype != torch.float32:
        x = x.float()
    B, C, H, W = x.shape
    if C == 1 and x.is_contiguous():
        return x.view(B, H, W, 1)
    p = x.permute(0, 2, 3, 1)
    if p.is_contiguous():       # already channels_last in memory
        return p
    return ops.nchw_to_nhwc(x.contiguous())


def to_nchw_view(t):
    """Logical NCHW view (channels_last strides) of an NHWC buffer -- zero copy."""
    return t.permute(0, 3, 1, 2)


def grad_to_nhwc(g):
    """Incoming NCHW-shaped gradient -> dense NHWC buffer (copy only if the memory order differs)."""
    g = g.detach()
    if g.dtype not in (torch.float32, torch.bfloat16):
        g = g.float()
    B, C, H, W = g.shape
    if C == 1:
        return g.contiguous().view(B, H, W, 1)
    p = g.permute(0, 2, 3, 1)
    if p.is_contiguous():
        return p
    return ops.nchw_to_nhwc(g.contiguous())
