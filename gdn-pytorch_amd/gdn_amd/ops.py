"""Thin tensor-level wrappers over the C ABI (one Python function per entry point).

Activations are torch tensors used purely as device buffers: shape [B,H,W,C],
fp32, channel stride 1; a channel slice of a wider tensor is allowed (pixel
pitch ``ld`` = stride(2)).  Everything launches on torch's current stream.
"""
import ctypes
import os

import torch

from ._lib import ConvGeom, GdnError, lib

ACT_NONE, ACT_TANH, ACT_RELU = 0, 1, 2       # bit flags
BN_EPS, BN_MOMENTUM = 1e-5, 0.1


def stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _is_bf16(t):
    return 1 if t is not None and t.dtype == torch.bfloat16 else 0


def _ld(t):
    if t.shape[-1] == 1:          # single channel: the pixel pitch is all that matters
        return t.stride(-2) if t.shape[-2] > 1 else 1
    if t.stride(-1) != 1:
        raise GdnError("activation must have channel stride 1, got strides %s" % (t.stride(),))
    return t.stride(-2)


def _chk(t, name="tensor", bf16_ok=False):
    ok = t.dtype == torch.float32 or (bf16_ok and t.dtype == torch.bfloat16)
    if not ok or not t.is_cuda:
        raise GdnError("%s must be a CUDA/HIP float32 tensor (got %s on %s); the HIP path has no CPU fallback"
                       % (name, t.dtype, t.device))
    return t


CFG_BF16 = 0x10000      # GDN_CFG_BF16
HINT_TRAIN, HINT_NO_X3, HINT_NO_WINO_F4, HINT_FFT_NP32, HINT_FFT_NP40 = 1, 2, 4, 8, 16      # GDN_HINT_* bits of gdn_conv_geom.hints


def plan_override_hints():
    """Test / measurement hooks of THIS BINDING (the C library reads no environment variable): GDN_PLAN_BATCH=<n>,
    GDN_RING_CUS=<n>, GDN_FFT_NP=<32|40> become the plan-override fields of gdn_conv_geom.hints of every geometry built while
    they are set."""
    h = 0
    e = os.environ
    if e.get("GDN_PLAN_BATCH"):
        n = int(e["GDN_PLAN_BATCH"])
        if not 1 <= n <= 255:           # an 8-bit field: masking would silently plan for another batch
            raise GdnError("GDN_PLAN_BATCH=%d is outside 1..255" % n)
        h |= n << 8
    if e.get("GDN_RING_CUS"):
        n = int(e["GDN_RING_CUS"])
        if not 8 <= n <= 2040 or n % 8:
            raise GdnError("GDN_RING_CUS=%d must be a multiple of 8 in 8..2040" % n)
        h |= (n // 8) << 16
    if e.get("GDN_FFT_NP") in ("32", "40"):
        h |= HINT_FFT_NP32 if e["GDN_FFT_NP"] == "32" else HINT_FFT_NP40
    return h

# bf16 x 3 split products for the Winograd per-bin GEMMs.  The switch lives HERE, not in the library: the
# environment variable GDN_X3 is read once, at import; distributed._guard_shared_gpu (ranks sharing one GPU), tests and
# measurements call set_x3().  Every geometry handed to the C ABI carries the decision as GDN_HINT_NO_X3, and a forward's
# saved state remembers the hints it was written with, so its backward reads the weight set in the form it was written in
# even if the switch moved in between.
_x3 = os.environ.get("GDN_X3", "1")[:1] != "0"
_x3_explicit = "GDN_X3" in os.environ


def x3_enabled():
    return _x3


def x3_explicit():
    """True when the user chose (GDN_X3 in the environment at import, or set_x3(explicit=True))."""
    return _x3_explicit


def set_x3(on, explicit=False):
    """Switch the bf16 x 3 GEMMs on / off for every layer planned from now on; returns the previous setting."""
    global _x3, _x3_explicit
    prev, _x3 = _x3, bool(on)
    _x3_explicit = _x3_explicit or explicit
    return prev


def _state_x3(state):
    """The switch a forward wrote `state` under (None: stateless call, use the current one)."""
    return getattr(state, "_gdn_x3", None) if state is not None else None


# Winograd F(4x4,3x3) for the zero-padded 3x3 layers: GDN_WINO_F4=0 in the environment at import, or
# set_wino_f4(False), keeps F(2x2,3x3) -- carried to the library as GDN_HINT_NO_WINO_F4 in every geometry, like the switch above
# (the saved state of a forward is laid out for the plan it was written under; its backward passes the same hint).
_f4 = os.environ.get("GDN_WINO_F4", "1")[:1] != "0"


def wino_f4_enabled():
    return _f4


def set_wino_f4(on):
    """Switch the F(4x4,3x3) plan on / off for every layer planned from now on; returns the previous setting."""
    global _f4
    prev, _f4 = _f4, bool(on)
    return prev


def _state_f4(state):
    return getattr(state, "_gdn_f4", None) if state is not None else None


def _bf(t):
    """dtype-mask bit of a tensor argument: 1 for bfloat16, 0 for float32 / None."""
    if t is None or t.dtype == torch.float32:
        return 0
    if t.dtype == torch.bfloat16:
        return 1
    raise GdnError("unsupported dtype %s (float32 or bfloat16)" % t.dtype)


def _mask(*ts):
    m = 0
    for i, t in enumerate(ts):
        m |= _bf(t) << i
    return m


def _same_dtype(ref, *others):
    for o in others:
        if o is not None and o.dtype != ref.dtype:
            raise GdnError("mixed dtypes in one conv call: %s vs %s" % (ref.dtype, o.dtype))


_ws_cache = {}


def workspace(nbytes, device, tag="ws"):
    """Grow-only scratch buffer per (device, tag, current stream): calls on one stream are ordered, so they may share
    scratch memory; two streams (or threads driving their own streams) never do.  Inside a hipGraph capture a missing or
    too small buffer is allocated as a plain temporary instead (it lives in the graph's private pool and must not be
    handed to later eager calls)."""
    key = (device, tag, torch.cuda.current_stream(device).cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        if not torch.cuda.is_current_stream_capturing():
            _ws_cache[key] = buf
    return buf


_side_streams = {}
_FFT_OVERLAP = os.environ.get("GDN_FFT_NO_OVERLAP") is None


def side_stream(device):
    """The caller-owned second stream gdn_fftconv_bwd's weight-gradient chain runs on (one per device; every use is a
    fork from / join into the caller's current stream, so sharing it only serialises)."""
    st = _side_streams.get(device)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _side_streams[device] = st
    return st


class ShaderClock:
    """Measurement aid: the shader clock the chip holds while the launches inside the `with` block run on the current stream
    (gdn_clock_probe_*: one sleeping wave on a second stream reads s_memtime / s_memrealtime at both ends of the window).

        with ops.ShaderClock(dev) as clk:
            for _ in range(reps): launch()
        torch.cuda.synchronize(); clk.ghz()      # None if no stream was found that runs beside the current one

    The watcher only works on a stream whose hardware queue is not the current stream's: HIP deals its streams onto a handful
    of hardware queues, and a watcher that shares the queue of the stream it watches sits IN FRONT of the launches it is
    meant to bracket (the r06 full-suite runs: the watcher ran into its tick limit after earlier tests had created a few
    dozen streams).  So the side stream is chosen by trial -- a 1 ms watcher that must see a stop issued right behind it --
    among a high-priority stream and up to eight fresh ones, once per (device, current stream)."""

    _side = {}            # (device index, current stream handle) -> torch.cuda.Stream or None

    def __init__(self, device, max_s=2.0):
        # host-pinned, device-mapped: the flag is polled across XCDs whose L2s are not coherent for device memory (gdn_hip.h)
        self.buf = torch.zeros(4, dtype=torch.int64).pin_memory()
        self.max_ticks = int(min(max_s, 10.0) * 1e8)
        self.device = torch.device(device)
        self.side = self._find_side()

    def _trial(self, side, ticks):
        main = torch.cuda.current_stream(self.device)
        self.buf.zero_()
        lib.gdn_clock_probe_arm(self.buf.data_ptr(), main.cuda_stream)
        side.wait_stream(main)
        lib.gdn_clock_probe_watch(self.buf.data_ptr(), ticks, side.cuda_stream)
        lib.gdn_clock_probe_stop(self.buf.data_ptr(), main.cuda_stream)
        main.wait_stream(side)
        torch.cuda.synchronize(self.device)
        return bool(self.buf[3].item())

    def _find_side(self):
        main = torch.cuda.current_stream(self.device)
        key = (self.device.index or 0, main.cuda_stream)
        if key not in ShaderClock._side:
            found = None
            for i in range(9):
                cand = torch.cuda.Stream(device=self.device, priority=-1 if i == 0 else 0)
                if self._trial(cand, 100000):            # 1 ms: ended by the flag only if it ran beside the stop
                    found = cand
                    break
            ShaderClock._side[key] = found
        return ShaderClock._side[key]

    def __enter__(self):
        self.buf.zero_()
        if self.side is not None:
            main = torch.cuda.current_stream(self.device)
            lib.gdn_clock_probe_arm(self.buf.data_ptr(), main.cuda_stream)
            self.side.wait_stream(main)
            lib.gdn_clock_probe_watch(self.buf.data_ptr(), self.max_ticks, self.side.cuda_stream)
        return self

    def __exit__(self, *exc):
        if self.side is not None:
            main = torch.cuda.current_stream(self.device)
            lib.gdn_clock_probe_stop(self.buf.data_ptr(), main.cuda_stream)
            main.wait_stream(self.side)
        return False

    def read(self):
        """(shader cycles, 100 MHz ticks, ended_by_flag) of the window; call after a synchronize."""
        _, cyc, ticks, ended = [int(v) for v in self.buf.tolist()]
        return cyc, ticks, bool(ended)

    def ghz(self, min_ticks=1000):
        cyc, ticks, ended = self.read()
        if not ended or ticks < min_ticks or cyc <= 0:
            return None
        return cyc / ticks * 0.1


class Conv:
    """Geometry + launch helper for one Conv2d / ConvTranspose2d layer.

    Weights are tap-major [k*k, Cout, Cin] (see AE_model_unet._tap_view)."""

    def __init__(self, cin, cout, k, stride=1, pad=0, reflect=False, transposed=False):
        self.cin, self.cout, self.k, self.stride, self.pad = cin, cout, k, stride, pad
        self.reflect, self.transposed = bool(reflect and pad > 0), transposed
        self._geom = {}
        self._fwd_ws = {}
        self._slots = {}

    def geom(self, B, H, W, hints=0, x3=None, f4=None):
        """x3 / f4: None = the current switch; a saved state's backward passes what its forward used."""
        if not (_x3 if x3 is None else x3):
            hints |= HINT_NO_X3
        if not (_f4 if f4 is None else f4):
            hints |= HINT_NO_WINO_F4
        hints |= plan_override_hints()
        key = (B, H, W, hints)
        g = self._geom.get(key)
        if g is None:
            cg = ConvGeom(B, H, W, self.cin, self.cout, self.k, self.stride, self.pad,
                          1 if self.reflect else 0, 1 if self.transposed else 0, hints)
            ho, wo = ctypes.c_int32(), ctypes.c_int32()
            lib.gdn_conv_out_dims(ctypes.byref(cg), ctypes.byref(ho), ctypes.byref(wo))
            g = (cg, ctypes.byref(cg), ho.value, wo.value)
            self._geom[key] = g
        return g

    def stats_slots(self, B, H, W, tile_cfg=0):
        _, ref, _, _ = self.geom(B, H, W)
        return int(lib.gdn_conv_stats_slots(ref, tile_cfg))

    def fwd(self, x, w_tap, x2=None, stats=False, act=ACT_NONE, addsrc=None, tile_cfg=0, out=None, stats_out=None,
            affine=None):
        """y = tanh?(relu?(conv(cat(x, x2)) * scale + shift) + addsrc); returns y, or (y, stats_partials) when stats.
        affine = (scale, shift) per output channel (an eval-mode BatchNorm folded into the epilogue) or None."""
        _chk(x, "x", bf16_ok=True)
        bf = x.dtype == torch.bfloat16
        head_mixed = bf and self.cout == 1       # 1-channel heads: bf16 x, fp32 weights and depth map
        if head_mixed:
            _chk(w_tap, "head weight")
            if x2 is not None or addsrc is not None or stats or affine is not None:
                raise GdnError("the bf16 head takes no concat / addsrc / stats / affine")
        else:
            _same_dtype(x, x2, w_tap, addsrc, out)
        if bf:
            tile_cfg |= CFG_BF16
        B, H, W, C1 = x.shape
        _, ref, Ho, Wo = self.geom(B, H, W)
        ph = plan_override_hints()       # part of every cache key below: an override changes the plan, hence slot count and workspace
        c2 = 0 if x2 is None else x2.shape[3]
        if C1 + c2 != self.cin:
            raise GdnError("conv expects %d input channels, got %d" % (self.cin, C1 + c2))
        y = out if out is not None else torch.empty((B, Ho, Wo, self.cout),
                                                    dtype=torch.float32 if head_mixed else x.dtype, device=x.device)
        st = None
        if stats:
            st = stats_out
            slots = self._slots.get((B, H, W, tile_cfg, ph))
            if slots is None:
                slots = int(lib.gdn_conv_stats_slots(ref, tile_cfg))
                self._slots[(B, H, W, tile_cfg, ph)] = slots
            if st is None:
                st = torch.empty((slots, 2, self.cout), dtype=torch.float32, device=x.device)
            elif (st.dtype != torch.float32 or not st.is_contiguous() or st.dim() != 3 or st.shape[0] != slots
                  or tuple(st.shape[1:]) != (2, self.cout)):
                # the kernel writes one slot per tile of the configuration it picks: a buffer sized for another tile_cfg is
                # overrun (or leaves slots unwritten that the finalize pass then sums)
                raise GdnError("conv fwd: stats_out must be a dense float32 [%d, 2, %d] for this geometry and tile_cfg, got %s"
                               % (slots, self.cout, tuple(st.shape)))
        nb = self._fwd_ws.get((B, H, W, tile_cfg, ph))
        if nb is None:
            nb = int(lib.gdn_conv_fwd_workspace_bytes(ref, tile_cfg))
            self._fwd_ws[(B, H, W, tile_cfg, ph)] = nb
        ws = workspace(nb, x.device, "splitk") if nb else None
        try:
            lib.gdn_conv_fwd(ref, _p(x), _ld(x), _p(x2), 0 if x2 is None else _ld(x2), C1, _p(w_tap), _p(y), _ld(y),
                             _p(addsrc), 0 if addsrc is None else _ld(addsrc), _p(st),
                             None if affine is None else _p(affine[0]), None if affine is None else _p(affine[1]),
                             act, tile_cfg, _p(ws), nb, stream())
        except GdnError as e:
            raise GdnError("%s [conv %d->%d k%d s%d p%d reflect=%s transposed=%s, x %s ld %d, x2 %s]" % (
                e, self.cin, self.cout, self.k, self.stride, self.pad, self.reflect, self.transposed,
                tuple(x.shape), _ld(x), None if x2 is None else tuple(x2.shape))) from None
        return (y, st) if stats else y

    def dgrad_bnb_slots(self, B, H, W, dtype, tile_cfg=0):
        """Slots of the BatchNorm-backward partials dgrad's epilogue can emit for this layer (0: it cannot)."""
        _, ref, _, _ = self.geom(B, H, W)
        return int(lib.gdn_conv_dgrad_bnb_slots(ref, tile_cfg | (CFG_BF16 if dtype == torch.bfloat16 else 0)))

    def dgrad(self, dy, wt_tap, in_hw, addsrc=None, tile_cfg=0, bnb=None, up2x=0):
        """dx = dgrad(dy) (+ addsrc).  wt_tap: [k*k, Cin, Cout]; in_hw: layer input (H, W).
        bnb = (y, co, relu, partial): dx is the final gradient of [relu](BN_train(y)); `partial` [dgrad_bnb_slots, 2, Cin]
        receives that BatchNorm's backward partial sums (as wino_bwd's bnb).
        up2x (1: align_corners False, 2: True): the layer's input was the x2 bilinear upsampling of a tensor t; dx (and addsrc)
        are dL/dt [B, H/2, W/2, Cin] (reflection-padded layers: the fold pass applies the adjoint interpolation)."""
        _chk(dy, "dy", bf16_ok=True)
        _same_dtype(dy, wt_tap, addsrc)
        if dy.dtype == torch.bfloat16:
            tile_cfg |= CFG_BF16
        B = dy.shape[0]
        H, W = in_hw
        _, ref, Ho, Wo = self.geom(B, H, W)
        if tuple(dy.shape[1:]) != (Ho, Wo, self.cout):
            raise GdnError("dgrad: dy shape %s does not match layer output (%d,%d,%d)" % (tuple(dy.shape), Ho, Wo, self.cout))
        dx = torch.empty((B, H // 2, W // 2, self.cin) if up2x else (B, H, W, self.cin), dtype=dy.dtype, device=dy.device)
        if up2x and addsrc is not None and tuple(addsrc.shape) != tuple(dx.shape):
            raise GdnError("dgrad(up2x): addsrc must be the low-resolution gradient %s" % (tuple(dx.shape),))
        nb = int(lib.gdn_conv_dgrad_workspace_bytes(ref, tile_cfg))
        ws = workspace(nb, dy.device, "dgrad") if nb else None
        by, bco, brelu, bpart = bnb if bnb is not None else (None, None, False, None)
        if by is not None and by.dtype != dy.dtype:
            raise GdnError("dgrad: the BatchNorm input of the fused backward reduction must have the gradient's dtype")
        lib.gdn_conv_dgrad(ref, _p(dy), _ld(dy), _p(wt_tap), _p(dx), _ld(dx), _p(addsrc),
                           0 if addsrc is None else _ld(addsrc), _p(by), 0 if by is None else _ld(by), _p(bco),
                           1 if brelu else 0, _p(bpart), int(up2x), _p(ws), nb, tile_cfg, stream())
        return dx

    # ---- FFT-domain path (csrc/conv_fft.hip): stride-1 zero-padded fp32 layers with 64..256 channels ----
    def fft_ok(self, B, H, W, backward=False, train=False):
        """True when gdn_fftconv_fwd (and, with `backward`, gdn_fftconv_bwd) supports this layer at this input size.
        Stride-1 ConvTranspose2d layers are forward-only.  train (here and in fft_fwd / fft_bwd, the same value for one
        layer instance): GDN_HINT_TRAIN -- forward + backward of a trained layer, tiled for the sum of both."""
        _, ref, _, _ = self.geom(B, H, W, 1 if train else 0)
        if backward:
            return int(lib.gdn_fftconv_bwd_workspace_bytes(ref)) > 0
        return int(lib.gdn_fftconv_spectrum_bytes(ref)) > 0

    def fft_stats_slots(self, B, H, W, train=False):
        _, ref, _, _ = self.geom(B, H, W, 1 if train else 0)
        return int(lib.gdn_fftconv_stats_slots(ref))

    def fft_fwd(self, x, w_tap, stats=False, addsrc=None, spectrum=False, out=None, stats_out=None, affine=None,
                act=ACT_NONE, in_affine=None, in_relu=False, up2x=0, train=False):
        """y = conv(x) (+ addsrc) through the frequency domain; returns y, then the BatchNorm partials when `stats`,
        then the input spectrum (opaque uint8 buffer for fft_bwd) when `spectrum`.
        in_affine = (scale, shift): x is a raw conv output and the layer input is [relu](x*scale + shift), applied on load.
        up2x (1: align_corners False, 2: True): x is the low-resolution tensor, upsampled x2 (bilinear) on load."""
        _chk(x, "x"); _chk(w_tap, "w")
        B, H, W, C1 = x.shape
        if up2x:
            H, W = 2 * H, 2 * W
        _, ref, Ho, Wo = self.geom(B, H, W, 1 if train else 0)
        nb = int(lib.gdn_fftconv_fwd_workspace_bytes(ref))
        if nb == 0 or C1 != self.cin:
            raise GdnError("fftconv: unsupported layer k=%d stride=%d Cin=%d Cout=%d" % (self.k, self.stride, C1, self.cout))
        y = out if out is not None else torch.empty((B, Ho, Wo, self.cout), dtype=torch.float32, device=x.device)
        st = None
        if stats:
            st = stats_out if stats_out is not None else torch.empty(
                (int(lib.gdn_fftconv_stats_slots(ref)), 2, self.cout), dtype=torch.float32, device=x.device)
        xf = torch.empty(int(lib.gdn_fftconv_spectrum_bytes(ref)), dtype=torch.uint8, device=x.device) if spectrum else None
        ws = workspace(nb, x.device, "fft")
        lib.gdn_fftconv_fwd(ref, _p(x), _ld(x), _p(w_tap), _p(y), _ld(y), _p(addsrc), 0 if addsrc is None else _ld(addsrc),
                            _p(st), _p(affine[0]) if affine else None, _p(affine[1]) if affine else None, int(act),
                            _p(in_affine[0]) if in_affine else None, _p(in_affine[1]) if in_affine else None,
                            1 if in_relu else 0, int(up2x), _p(xf), _p(ws), nb, stream())
        res = (y,) + ((st,) if stats else ()) + ((xf,) if spectrum else ())
        return res if len(res) > 1 else y

    def wino_bnb_slots(self, B, H, W):
        _, ref, _, _ = self.geom(B, H, W)
        return int(lib.gdn_winoconv_bnb_slots(ref))

    def fft_bnb_slots(self, B, H, W, train=False):
        """Slots of the BatchNorm-backward partial sums fft_bwd's data-gradient gather can emit (`bnb`); 0: not available."""
        _, ref, _, _ = self.geom(B, H, W, 1 if train else 0)
        return int(lib.gdn_fftconv_bnb_slots(ref))

    def fft_bwd(self, dy, w_tap, in_hw, xf=None, dw_tap=None, need_dx=True, addsrc=None, dyb=None, up2x=0, train=False,
                bnb=None):
        """Data gradient (returned; + addsrc) and / or weight gradient (into dw_tap, needs the forward's saved state xf:
        input + weight spectra) from one transform of dy.  w_tap is the FORWARD tap-major weight [k*k, Cout, Cin]; it is
        only read when xf is None.
        dyb = (y_raw, coeffs[4,Cout], kk[2,Cout], relu): `dy` is dout of THIS layer's train-mode BatchNorm; the dy transform
        applies scale*(dz - k1 - xhat*k2) while loading (kk from bn_bwd_coeffs).
        up2x: the forward upsampled a low-resolution x on load; dx (and addsrc) are that tensor's gradient [B,H/2,W/2,Cin]
        (in_hw stays the convolution's input extent).
        bnb = (y_in, coeffs[4,Cin], relu, partial[fft_bnb_slots,2,Cin]): dx is the final gradient of this layer's input
        [relu](BN_train(y_in)); the pass that writes dx fills `partial` with that BatchNorm's backward sums (bn_bwd `partial`)."""
        _chk(dy, "dy")
        B = dy.shape[0]
        H, W = in_hw
        _, ref, Ho, Wo = self.geom(B, H, W, 1 if train else 0)
        nb = int(lib.gdn_fftconv_bwd_workspace_bytes(ref))
        if nb == 0:
            raise GdnError("fftconv: unsupported layer k=%d stride=%d" % (self.k, self.stride))
        by, bco, brelu, bpart = bnb if (bnb is not None and need_dx) else (None, None, False, None)
        if bpart is not None and (bpart.dtype != torch.float32 or not bpart.is_contiguous()
                                  or tuple(bpart.shape) != (int(lib.gdn_fftconv_bnb_slots(ref)), 2, self.cin)):
            raise GdnError("fft_bwd: bnb partial must be a dense float32 [%d, 2, %d]" % (int(lib.gdn_fftconv_bnb_slots(ref)), self.cin))
        if tuple(dy.shape[1:]) != (Ho, Wo, self.cout):
            raise GdnError("fft_bwd: dy shape %s does not match layer output" % (tuple(dy.shape),))
        dx = torch.empty((B, H // 2, W // 2, self.cin) if up2x else (B, H, W, self.cin), dtype=torch.float32,
                         device=dy.device) if need_dx else None
        ws = workspace(nb, dy.device, "fft")
        yy, yco, ykk, yrelu = dyb if dyb is not None else (None, None, None, False)

        def call(phases, st):
            lib.gdn_fftconv_bwd(ref, _p(dy), _ld(dy), _p(w_tap), _p(xf), _p(dx),
                                0 if dx is None else _ld(dx), _p(addsrc), 0 if addsrc is None else _ld(addsrc),
                                _p(dw_tap), _p(yy), 0 if yy is None else _ld(yy), _p(yco), _p(ykk), 1 if yrelu else 0,
                                _p(by), 0 if by is None else _ld(by), _p(bco), 1 if brelu else 0, _p(bpart),
                                int(up2x), phases, _p(ws), nb, st)
        if dw_tap is not None and need_dx and _FFT_OVERLAP:
            # the two chains only share the spectrum of dy and are each latency-bound: the weight-gradient chain runs on a
            # second stream of OURS next to the data-gradient chain (fork after the transform, join before returning)
            main = torch.cuda.current_stream(dy.device)
            side = side_stream(dy.device)
            call(1, main.cuda_stream)
            side.wait_stream(main)
            call(2, side.cuda_stream)
            call(4, main.cuda_stream)
            main.wait_stream(side)
        else:
            call(0, stream())
        return dx

    # ---- Winograd F(2x2,3x3) path (csrc/conv_wino.hip): 3x3 stride-1 zero-padded fp32 layers with 64..512 channels ----
    def fft_cgemm_only(self, B, H, W, which, ws=None, train=False, st=None):
        """Measurement hook: only the per-bin complex GEMMs of this layer's frequency-domain plan (which: 0 forward, 1 data
        gradient, 2 weight-gradient reduction) on a scratch workspace; returns (workspace, bins, tiles, points)."""
        _, ref, _, _ = self.geom(B, H, W, hints=HINT_TRAIN if train else 0)
        nb = int(lib.gdn_fftconv_cgemm_workspace_bytes(ref))
        if nb == 0:
            raise GdnError("fftconv: unsupported layer k=%d stride=%d" % (self.k, self.stride))
        if ws is None:
            ws = torch.empty(nb // 4, dtype=torch.float32, device="cuda").normal_()
        lib.gdn_fftconv_cgemm(ref, int(which), _p(ws), nb, stream() if st is None else st)
        bins, M, npnt = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        lib.gdn_fftconv_cgemm_shape(ref, ctypes.byref(bins), ctypes.byref(M), ctypes.byref(npnt))
        return ws, bins.value, M.value, npnt.value

    def wino_ok(self, B, H, W):
        _, ref, _, _ = self.geom(B, H, W)
        return int(lib.gdn_winoconv_state_bytes(ref)) > 0

    def wino_fwd(self, x, w_tap, stats=False, addsrc=None, state=False, affine=None, act=ACT_NONE, in_affine=None,
                 in_relu=False, up2x=0):
        """y = conv3x3(x) (+ epilogue) by Winograd F(2x2,3x3); returns y, then the BatchNorm partials when `stats`, then
        the transformed input (opaque buffer for wino_bwd) when `state`.  up2x: as for fft_fwd."""
        _chk(x, "x"); _chk(w_tap, "w")
        B, H, W, C1 = x.shape
        if up2x:
            H, W = 2 * H, 2 * W
        _, ref, Ho, Wo = self.geom(B, H, W)
        nb = int(lib.gdn_winoconv_fwd_workspace_bytes(ref))
        if nb == 0 or C1 != self.cin:
            raise GdnError("winoconv: unsupported layer k=%d stride=%d Cin=%d Cout=%d" % (self.k, self.stride, C1, self.cout))
        y = torch.empty((B, Ho, Wo, self.cout), dtype=torch.float32, device=x.device)
        st = torch.empty((int(lib.gdn_winoconv_stats_slots(ref)), 2, self.cout), dtype=torch.float32,
                         device=x.device) if stats else None
        sv = torch.empty(int(lib.gdn_winoconv_state_bytes(ref)), dtype=torch.uint8, device=x.device) if state else None
        if sv is not None:
            sv._gdn_x3 = _x3                 # the form of the saved weight set (fp32 / bf16 x 3 panels) follows the switch
            sv._gdn_f4 = _f4                 # ... and the layout of the saved input transform (16 / 36 bins) the plan
        ws = workspace(nb, x.device, "fft")
        lib.gdn_winoconv_fwd(ref, _p(x), _ld(x), _p(w_tap), _p(y), _ld(y), _p(addsrc), 0 if addsrc is None else _ld(addsrc),
                             _p(st), _p(affine[0]) if affine else None, _p(affine[1]) if affine else None, int(act),
                             _p(in_affine[0]) if in_affine else None, _p(in_affine[1]) if in_affine else None,
                             1 if in_relu else 0, int(up2x), _p(sv), _p(ws), nb, stream())
        res = (y,) + ((st,) if stats else ()) + ((sv,) if state else ())
        return res if len(res) > 1 else y

    def wino_gemm_only(self, V, U, Mo, B, H, W):
        """Measurement hook: the 16 per-bin GEMMs of one forward on already transformed operands."""
        _, ref, _, _ = self.geom(B, H, W)
        lib.gdn_winoconv_gemm(ref, _p(V), _p(U), _p(Mo), stream())

    def wino_bwd(self, dy, w_tap, in_hw, state=None, dw_tap=None, need_dx=True, addsrc=None, bnb=None, up2x=0):
        """Data gradient (returned; + addsrc) and / or weight gradient (into dw_tap, needs the forward's `state`).
        w_tap is the FORWARD tap-major weight [9, Cout, Cin]."""
        _chk(dy, "dy")
        B = dy.shape[0]
        H, W = in_hw
        _, ref, Ho, Wo = self.geom(B, H, W, x3=_state_x3(state), f4=_state_f4(state))
        nb = int(lib.gdn_winoconv_bwd_workspace_bytes(ref))
        if nb == 0:
            raise GdnError("winoconv: unsupported layer k=%d stride=%d" % (self.k, self.stride))
        if tuple(dy.shape[1:]) != (Ho, Wo, self.cout):
            raise GdnError("wino_bwd: dy shape %s does not match layer output" % (tuple(dy.shape),))
        dx = torch.empty((B, H // 2, W // 2, self.cin) if up2x else (B, H, W, self.cin), dtype=torch.float32,
                         device=dy.device) if need_dx else None
        ws = workspace(nb, dy.device, "fft")
        by, bco, brelu, bpart = bnb if (bnb is not None and need_dx) else (None, None, False, None)
        lib.gdn_winoconv_bwd(ref, _p(dy), _ld(dy), _p(w_tap), _p(state), _p(dx), 0 if dx is None else _ld(dx), _p(addsrc),
                             0 if addsrc is None else _ld(addsrc), _p(dw_tap), _p(by), 0 if by is None else _ld(by), _p(bco),
                             1 if brelu else 0, _p(bpart), int(up2x), _p(ws), nb, stream())
        return dx

    # ---- Winograd F(3x3,2x2) path (csrc/conv_wino2.hip): 4x4 stride-2 pad-1 Conv2d / ConvTranspose2d, fp32, 64..512 channels ----
    def wino2_ok(self, B, H, W):
        _, ref, _, _ = self.geom(B, H, W)
        return int(lib.gdn_wino2conv_fwd_workspace_bytes(ref)) > 0

    def wino2_fwd(self, x, w_tap, stats=False, addsrc=None, state=False, affine=None, act=ACT_NONE):
        """y = conv / conv-transpose (k4, s2, p1) (+ epilogue) by Winograd F(3x3,2x2) over the polyphase images; returns y,
        then the BatchNorm partials when `stats`, then the saved state for wino2_bwd when `state` (the transformed input of a
        Conv2d; the input tensor itself for a ConvTranspose2d, whose backward transforms dy instead)."""
        _chk(x, "x"); _chk(w_tap, "w")
        B, H, W, C1 = x.shape
        _, ref, Ho, Wo = self.geom(B, H, W)
        nb = int(lib.gdn_wino2conv_fwd_workspace_bytes(ref))
        if nb == 0 or C1 != self.cin:
            raise GdnError("wino2conv: unsupported layer k=%d stride=%d Cin=%d Cout=%d" % (self.k, self.stride, C1, self.cout))
        y = torch.empty((B, Ho, Wo, self.cout), dtype=torch.float32, device=x.device)
        st = torch.empty((int(lib.gdn_wino2conv_stats_slots(ref)), 2, self.cout), dtype=torch.float32,
                         device=x.device) if stats else None
        sv = None
        if state:
            sb = int(lib.gdn_wino2conv_state_bytes(ref))
            sv = torch.empty(sb, dtype=torch.uint8, device=x.device) if sb else x
        ws = workspace(nb, x.device, "fft")
        lib.gdn_wino2conv_fwd(ref, _p(x), _ld(x), _p(w_tap), _p(y), _ld(y), _p(addsrc), 0 if addsrc is None else _ld(addsrc),
                              _p(st), _p(affine[0]) if affine else None, _p(affine[1]) if affine else None, int(act),
                              _p(sv) if (sv is not None and sv is not x) else None, _p(ws), nb, stream())
        res = (y,) + ((st,) if stats else ()) + ((sv,) if state else ())
        return res if len(res) > 1 else y

    def wino2_bwd(self, dy, w_tap, in_hw, state=None, dw_tap=None, need_dx=True, addsrc=None, bnb=None):
        """Data gradient (returned; + addsrc) and / or weight gradient (into dw_tap, needs the forward's `state`)."""
        _chk(dy, "dy")
        B = dy.shape[0]
        H, W = in_hw
        _, ref, Ho, Wo = self.geom(B, H, W)
        nb = int(lib.gdn_wino2conv_bwd_workspace_bytes(ref))
        if nb == 0:
            raise GdnError("wino2conv: unsupported layer k=%d stride=%d" % (self.k, self.stride))
        if tuple(dy.shape[1:]) != (Ho, Wo, self.cout):
            raise GdnError("wino2_bwd: dy shape %s does not match layer output" % (tuple(dy.shape),))
        dx = torch.empty((B, H, W, self.cin), dtype=torch.float32, device=dy.device) if need_dx else None
        ws = workspace(nb, dy.device, "fft")
        xin = state if self.transposed else None
        sv = None if self.transposed else state
        lib.gdn_wino2conv_bwd(ref, _p(dy), _ld(dy), _p(w_tap), _p(xin), 0 if xin is None else _ld(xin), _p(sv), _p(dx),
                              0 if dx is None else _ld(dx), _p(addsrc), 0 if addsrc is None else _ld(addsrc), _p(dw_tap),
                              _p(ws), nb, stream())
        return dx

    def wgrad(self, x, dy, dw_tap, ci_off=0, cfg=0):
        """dw_tap[tap][co][ci_off + ci] = wgrad over the channel slice x (Cx = x.shape[3]).
        bf16 x/dy take the bf16 MFMA kernel; dw_tap is fp32 either way."""
        _chk(x, "x", bf16_ok=True)
        _chk(dy, "dy", bf16_ok=True)
        _chk(dw_tap, "dw")
        B, H, W, Cx = x.shape
        _, ref, _, _ = self.geom(B, H, W)
        if x.dtype == torch.bfloat16 and dy.dtype == torch.bfloat16:
            nb = int(lib.gdn_conv_wgrad_bf16_workspace_bytes(ref, Cx, cfg))
            if nb == 0:
                raise GdnError("bf16 wgrad: unsupported geometry k=%d stride=%d Cx=%d Cout=%d" % (self.k, self.stride, Cx, self.cout))
            ws = workspace(nb, x.device, "wgrad")
            lib.gdn_conv_wgrad_bf16(ref, _p(x), _ld(x), Cx, _p(dy), _ld(dy), _p(dw_tap), self.cin, ci_off, _p(ws), nb, cfg,
                                    stream())
            return
        nb = int(lib.gdn_conv_wgrad_workspace_bytes(ref, Cx))
        if nb == 0:
            raise GdnError("wgrad: unsupported geometry k=%d stride=%d Cx=%d" % (self.k, self.stride, Cx))
        ws = workspace(nb, x.device, "wgrad")
        lib.gdn_conv_wgrad(ref, _p(x), _ld(x), Cx, _p(dy), _ld(dy), _p(dw_tap), self.cin, ci_off, _p(ws), nb,
                           _mask(x, dy), stream())


def c1_ok(x1, n, k, stride, pad, rgb=False):
    """The 1 <-> 64 channel 9x9 layers of csrc/conv_c1.hip: a dense fp32 single-channel image against 64 channels
    (rgb: the forward also takes a 3-channel image, R's first layer)."""
    return (x1.dtype == torch.float32 and x1.dim() == 4 and (x1.shape[3] == 1 or (rgb and x1.shape[3] == 3)) and x1.is_contiguous()
            and n == 64 and k == 9 and stride == 1 and pad == 4)


def conv_c1_fwd(x1, w81, reflect=False, flip=False, stats=False, addsrc=None, affine=None, act=ACT_NONE, out_dtype=torch.float32):
    """y[B,H,W,64] = sum_{tap,c} x1[p + tap - 4][c] * w81[tap][:][c]; x1 [B,H,W,Cin], Cin = 1 or 3, w81 any tensor of 81*64*Cin
    floats in [tap][64][Cin] order.  y (out_dtype) and addsrc may be bf16: a bf16 model's head data gradient."""
    _chk(x1, "x1"); _chk(w81, "w")
    if x1.dtype != torch.float32 or w81.dtype != torch.float32:
        raise GdnError("conv_c1_fwd: the image and the weights are fp32")
    B, H, W, cin = x1.shape
    if cin not in (1, 3) or w81.numel() != 81 * 64 * cin or not w81.is_contiguous() or not x1.is_contiguous():
        raise GdnError("conv_c1_fwd: dense [B,H,W,1|3] image and 81 x 64 x Cin contiguous weights expected")
    y = torch.empty((B, H, W, 64), dtype=out_dtype, device=x1.device)
    st = torch.empty((int(lib.gdn_conv_c1_stats_slots(B, H, W)), 2, 64), dtype=torch.float32, device=x1.device) if stats else None
    dtypes = _is_bf16(y) | (_is_bf16(addsrc) << 1)
    lib.gdn_conv_c1_fwd(_p(x1), cin, B, H, W, 64, 9, 4, 1 if reflect else 0, 1 if flip else 0, _p(w81), _p(y), 64, _p(addsrc),
                        0 if addsrc is None else _ld(addsrc), _p(st), _p(affine[0]) if affine else None,
                        _p(affine[1]) if affine else None, int(act), dtypes, stream())
    return (y, st) if stats else y


def conv_c1_wgrad(x1, gw, dw81, reflect=False, flip=False):
    """dw81[tap][:] = sum_p gw[p][:] * x1[p + tap - 4] (written at the flipped tap when flip); gw [B,H,W,64], fp32 or bf16."""
    _chk(x1, "x1"); _chk(gw, "gw", bf16_ok=True); _chk(dw81, "dw")
    B, H, W, _ = x1.shape
    if dw81.numel() != 81 * 64 or not dw81.is_contiguous() or tuple(gw.shape) != (B, H, W, 64):
        raise GdnError("conv_c1_wgrad: bad shapes")
    nb = int(lib.gdn_conv_c1_wgrad_workspace_bytes())
    ws = workspace(nb, x1.device, "wgrad")
    lib.gdn_conv_c1_wgrad(_p(x1), _p(gw), _ld(gw), _is_bf16(gw), B, H, W, 64, 9, 4, 1 if reflect else 0, 1 if flip else 0, _p(dw81),
                          _p(ws), nb, stream())


def gemm_x3_pack(Bm):
    """fp32 [bins, rows, K] -> packed bf16 x 3 panels (opaque uint8 buffer for gemm_x3_nt)."""
    _chk(Bm, "B")
    bins, rows, K = Bm.shape
    nb = int(lib.gdn_gemm_x3_packed_bytes(bins, rows, K))
    if nb == 0 or not Bm.is_contiguous():
        raise GdnError("gemm_x3_pack: needs a dense [bins, rows, K] tensor with K a multiple of 32")
    out = torch.empty(nb, dtype=torch.uint8, device=Bm.device)
    lib.gdn_gemm_x3_pack(_p(Bm), _p(out), bins, rows, K, stream())
    return out


def gemm_x3_nt(A, Bp, N, out=None):
    """C[bin] = A[bin] @ B[bin]^T with fp32 A [bins, M, K], packed B (gemm_x3_pack of [bins, N, K]); fp32 C [bins, M, N]."""
    _chk(A, "A")
    bins, M, K = A.shape
    C = out if out is not None else torch.empty((bins, M, N), dtype=torch.float32, device=A.device)
    lib.gdn_gemm_x3_nt(_p(A), _p(Bp), _p(C), bins, M, N, K, stream())
    return C


def gemm_x3_tn(A, Bm, nsplit=1):
    """P[split, bin] = A[bin]^T @ B[bin] over the split's rows; A [bins, T, NI], B [bins, T, NJ] fp32 -> [nsplit, bins, NI, NJ]."""
    _chk(A, "A"); _chk(Bm, "B")
    bins, T, NI = A.shape
    NJ = Bm.shape[2]
    P = torch.empty((nsplit, bins, NI, NJ), dtype=torch.float32, device=A.device)
    lib.gdn_gemm_x3_tn(_p(A), _p(Bm), _p(P), bins, T, NI, NJ, nsplit, stream())
    return P


def transpose_taps(w_tap, out=None, dtype=None):
    """[T, R, C] -> [T, C, R]; dtype (or out.dtype) may differ from w_tap's: fp32 master -> bf16 copy."""
    T, R, C = w_tap.shape
    wt = out if out is not None else torch.empty((T, C, R), dtype=dtype or w_tap.dtype, device=w_tap.device)
    lib.gdn_transpose_taps(_p(w_tap), _p(wt), T, R, C, _mask(w_tap, wt), stream())
    return wt


def cast(src, dtype=None, out=None):
    """Dense dtype conversion fp32 <-> bf16 (one streaming kernel)."""
    if not src.is_contiguous():
        raise GdnError("cast needs a dense tensor")
    dst = out if out is not None else torch.empty(src.shape, dtype=dtype, device=src.device)
    lib.gdn_cast(_p(src), _p(dst), src.numel(), _mask(src, dst), stream())
    return dst


def weight_to_tapmajor(w, transposed):
    """torch-layout conv weight -> tap-major [k*k, Cout, Cin] (w contiguous)."""
    A, Bc, kh, kw = w.shape
    cout, cin = (Bc, A) if transposed else (A, Bc)
    out = torch.empty((kh * kw, cout, cin), dtype=torch.float32, device=w.device)
    lib.gdn_weight_to_tapmajor(_p(w.contiguous()), _p(out), cout, cin, kh * kw, 0 if transposed else 1, stream())
    return out


def weight_from_tapmajor(w_tap, k, transposed):
    T, cout, cin = w_tap.shape
    shape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
    out = torch.empty(shape, dtype=torch.float32, device=w_tap.device)
    lib.gdn_weight_from_tapmajor(_p(w_tap), _p(out), cout, cin, T, 0 if transposed else 1, stream())
    return out


def bn_finalize_train(stats, count, gamma, beta, running_mean, running_var, momentum=BN_MOMENTUM, eps=BN_EPS,
                      num_batches_tracked=None):
    slots, _, C = stats.shape
    co = torch.empty((4, C), dtype=torch.float32, device=stats.device)   # scale, shift, mean, invstd
    if num_batches_tracked is not None and (num_batches_tracked.dtype != torch.int64 or not num_batches_tracked.is_cuda):
        raise GdnError("num_batches_tracked must be a device int64 tensor")
    lib.gdn_bn_finalize_train(_p(stats), slots, C, int(count), _p(gamma), _p(beta), _p(running_mean), _p(running_var),
                              momentum, eps, _p(co[0]), _p(co[1]), _p(co[2]), _p(co[3]), _p(num_batches_tracked), stream())
    return co


def bn_eval_coeffs(gamma, beta, running_mean, running_var, eps=BN_EPS):
    C = running_mean.numel()
    co = torch.empty((2, C), dtype=torch.float32, device=running_mean.device)
    lib.gdn_bn_eval_coeffs(_p(gamma), _p(beta), _p(running_mean), _p(running_var), eps, C, _p(co[0]), _p(co[1]), stream())
    return co


def bn_apply(y, scale, shift, relu, residual=None, out=None, out_dtype=None):
    B, H, W, C = y.shape
    o = out if out is not None else torch.empty((B, H, W, C), dtype=out_dtype or y.dtype, device=y.device)
    lib.gdn_bn_apply(_p(y), _ld(y), _p(scale), _p(shift), _p(residual), 0 if residual is None else _ld(residual),
                     _p(o), _ld(o), B * H * W, C, 1 if relu else 0, _mask(y, residual, o), stream())
    return o


def bn_bwd(dout, y, gamma, coeffs, relu, dgamma, dbeta, out_dtype=None, partial=None):
    """coeffs = [scale, shift, mean, invstd] from bn_finalize_train. Returns dy.
    partial [slots,2,C]: the reduce pass was done by the epilogue that wrote dout (wino_bwd `bnb`)."""
    B, H, W, C = y.shape
    npix = B * H * W
    dy = torch.empty((B, H, W, C), dtype=out_dtype or y.dtype, device=y.device)
    nb = int(lib.gdn_bn_bwd_workspace_bytes(npix, C))
    ws = workspace(nb, y.device, "bnbwd")
    lib.gdn_bn_bwd(_p(dout), _ld(dout), _p(y), _ld(y), _p(gamma), _p(coeffs[0]), _p(coeffs[1]), _p(coeffs[2]),
                   _p(coeffs[3]), _p(dy), _ld(dy), _p(dgamma), _p(dbeta), npix, C, 1 if relu else 0,
                   _p(partial), 0 if partial is None else partial.shape[0], _p(ws), nb, _mask(dout, y, dy), stream())
    return dy


def bn_bwd_coeffs(dout, y, coeffs, relu, dgamma, dbeta, partial=None):
    """Passes 1 + 2 of the BatchNorm backward: writes dgamma / dbeta (may be None), returns kk [2,C] = mean(dz), mean(dz*xhat)
    for a consumer that applies pass 3 in its loader (Conv.fft_bwd dyb)."""
    B, H, W, C = y.shape
    npix = B * H * W
    kk = torch.empty((2, C), dtype=torch.float32, device=y.device)
    nb = int(lib.gdn_bn_bwd_workspace_bytes(npix, C))
    ws = workspace(nb, y.device, "bnbwd")
    lib.gdn_bn_bwd_coeffs(_p(dout), _ld(dout), _p(y), _ld(y), _p(coeffs[0]), _p(coeffs[1]), _p(coeffs[2]), _p(coeffs[3]),
                          _p(dgamma), _p(dbeta), _p(kk), npix, C, 1 if relu else 0, _p(partial),
                          0 if partial is None else partial.shape[0], _p(ws), nb, _mask(dout, y), stream())
    return kk


def bn_eval_bwd(dout, y, coeffs, relu, out_dtype=None):
    """Backward through an eval-mode BN (+ReLU): coeffs = [scale, shift] from bn_eval_coeffs. Returns dy.
    relu: False/0 none, True/1 y is the raw conv output, 2 y is the activated output of a fused epilogue."""
    B, H, W, C = y.shape
    dy = torch.empty((B, H, W, C), dtype=out_dtype or y.dtype, device=y.device)
    lib.gdn_bn_eval_bwd(_p(dout), _ld(dout), _p(y), _ld(y), _p(coeffs[0]), _p(coeffs[1]), _p(dy), _ld(dy), B * H * W, C,
                        int(relu), _mask(dout, y, dy), stream())
    return dy


def upsample2x(x, align_corners=False):
    B, H, W, C = x.shape
    if not x.is_contiguous():
        raise GdnError("upsample2x needs a dense NHWC tensor")
    y = torch.empty((B, 2 * H, 2 * W, C), dtype=x.dtype, device=x.device)
    lib.gdn_upsample2x_fwd(_p(x), _p(y), B, H, W, C, 1 if align_corners else 0, _mask(x, y), stream())
    return y


def bn_apply_up2x(y, scale, shift, relu, residual, align_corners=False, out_dtype=None, need_low=True):
    """(low, up): low = [relu](y * scale + shift) (+ residual) and up = its x2 bilinear upsampling, one pass (gdn_bn_apply_up2x);
    low is None when need_low is False."""
    B, H, W, C = y.shape
    if not y.is_contiguous() or (residual is not None and not residual.is_contiguous()):
        raise GdnError("bn_apply_up2x needs dense NHWC tensors")
    odt = out_dtype or y.dtype
    low = torch.empty((B, H, W, C), dtype=odt, device=y.device) if need_low else None
    up = torch.empty((B, 2 * H, 2 * W, C), dtype=odt, device=y.device)
    dt = _mask(y, residual, up, up)              # (low is stored in up's type, also when it is not written)
    lib.gdn_bn_apply_up2x(_p(y), _p(scale), _p(shift), _p(residual), _p(low), _p(up), B, H, W, C, 1 if relu else 0,
                          1 if align_corners else 0, dt, stream())
    return low, up


def upsample2x_bwd(dy, align_corners=False):
    B, H2, W2, C = dy.shape
    if not dy.is_contiguous():
        raise GdnError("upsample2x_bwd needs a dense NHWC tensor")
    dx = torch.empty((B, H2 // 2, W2 // 2, C), dtype=dy.dtype, device=dy.device)
    lib.gdn_upsample2x_bwd(_p(dy), _p(dx), B, H2 // 2, W2 // 2, C, 1 if align_corners else 0, _mask(dy, dx), stream())
    return dx


def nchw_to_nhwc(x, dtype=None):
    B, C, H, W = x.shape
    y = torch.empty((B, H, W, C), dtype=dtype or x.dtype, device=x.device)
    lib.gdn_nchw_to_nhwc(_p(x), _p(y), B, C, H, W, _mask(x, y), stream())
    return y


def nhwc_to_nchw(x, dtype=None):
    B, H, W, C = x.shape
    y = torch.empty((B, C, H, W), dtype=dtype or x.dtype, device=x.device)
    lib.gdn_nhwc_to_nchw(_p(x), _p(y), B, C, H, W, _mask(x, y), stream())
    return y


def add(a, b, out_dtype=None):
    if not (a.is_contiguous() and b.is_contiguous()):
        raise GdnError("add needs dense tensors")
    o = torch.empty(a.shape, dtype=out_dtype or a.dtype, device=a.device)
    lib.gdn_add(_p(a), _p(b), _p(o), a.numel(), _mask(a, b, o), stream())
    return o


def add_pitched(a, b=None, out_dtype=None):
    """Dense a (+ b) of [B,H,W,C] tensors that may be channel slices of wider ones (b None: compacting copy)."""
    B, H, W, C = a.shape
    o = torch.empty((B, H, W, C), dtype=out_dtype or a.dtype, device=a.device)
    lib.gdn_add_pitched(_p(a), _ld(a), _p(b), 0 if b is None else _ld(b), _p(o), C, B * H * W, C, _mask(a, b, o), stream())
    return o


def copy_rows(src, dst):
    """dst[...] = src for two dense fp32 tensors of equal size whose last dimension is a multiple of 4 (gdn_add_pitched as a copy)."""
    n = src.numel()
    w = src.shape[-1]
    lib.gdn_add_pitched(_p(src), w, None, 0, _p(dst), w, n // w, w, 0, stream())
    return dst


def scale_dev(x, s):
    """x * s with s a 0-dim device tensor (no host sync, no torch kernel)."""
    o = torch.empty_like(x)
    lib.gdn_scale_dev(_p(x), _p(s), _p(o), x.numel(), stream())
    return o


def tanh_bwd(dout, out):
    r = torch.empty_like(out)
    lib.gdn_tanh_bwd(_p(dout), _p(out), _p(r), out.numel(), stream())
    return r


def fill_(t, value):
    lib.gdn_fill(_p(t), float(value), t.numel(), stream())
    return t


def zeros(shape, device):
    return fill_(torch.empty(shape, dtype=torch.float32, device=device), 0.0)


def _loss_ws(npix, device):
    nb = int(lib.gdn_loss_workspace_bytes(npix))
    return workspace(nb, device, "loss"), nb


def absdiff_max(a, b):
    """max|a-b| as a device scalar (the BerHu threshold's maximum; all-reduce it for --global_berhu)."""
    m = torch.empty((), dtype=torch.float32, device=a.device)
    lib.gdn_absdiff_max(_p(a), _p(b), a.numel(), _p(m), stream())
    return m


def berhu_masked(out, gt, sparse, box, dout, loss, ext_max=None):
    """out/gt [B,1,H,W]; sparse [B,Cs,H,W] or None; adds the gradient into dout; writes loss[()]."""
    B, _, H, W = out.shape
    ws, nb = _loss_ws(B * H * W, out.device)
    cbox = (ctypes.c_int32 * 4)(*box) if box is not None else None
    lib.gdn_berhu_masked(_p(out), _p(gt), _p(sparse), 0 if sparse is None else sparse.shape[1], B, H, W, cbox,
                         _p(ext_max), _p(loss), _p(dout), _p(ws), nb, stream())


def sobel_l1(pred, gt, weight, dpred, loss, plus=None, total=None):
    """total (0-dim device tensor) = loss + plus: the step's loss sum comes out of the same kernel."""
    B, _, H, W = pred.shape
    ws, nb = _loss_ws(B * H * W, pred.device)
    lib.gdn_sobel_l1(_p(pred), _p(gt), B, H, W, float(weight), _p(loss), _p(dpred), _p(plus), _p(total), _p(ws), nb, stream())


def smoothness(depth, img, ddepth, loss, plus=None, plus2=None, total=None):
    B, _, H, W = depth.shape
    ws, nb = _loss_ws(B * H * W, depth.device)
    lib.gdn_smoothness(_p(depth), _p(img), img.shape[1], B, H, W, _p(loss), _p(ddepth), _p(plus), _p(plus2), _p(total),
                       _p(ws), nb, stream())


def mse_accum(a, b, weight, loss, accumulate):
    ws, nb = _loss_ws(0, a.device)
    lib.gdn_mse(_p(a), _p(b), a.numel(), float(weight), 1 if accumulate else 0, _p(loss), _p(ws), nb, _mask(a, b), stream())


def mse_grad(a, b, weight, gscale, out_dtype=None):
    """d/da of weight*mean((a-b)^2), times the device scalar gscale (or None)."""
    da = torch.empty_like(a, dtype=out_dtype or a.dtype)       # preserves a's (dense) memory order
    if da.stride() != a.stride():
        raise GdnError("mse_grad needs a dense tensor")
    lib.gdn_mse_grad(_p(a), _p(b), a.numel(), float(weight), _p(gscale), _p(da), _mask(a, b, da), stream())
    return da


def depth_metrics(gt_sparse, gt, pred, crop=True):
    B, _, H, W = pred.shape
    nb = int(lib.gdn_depth_metrics_workspace_bytes(B, H, W))
    ws = workspace(nb, pred.device, "metrics")
    err = torch.empty(8, dtype=torch.float32, device=pred.device)
    lib.gdn_depth_metrics(_p(gt_sparse), _p(gt), _p(pred), B, H, W, 1 if crop else 0, _p(err), _p(ws), nb, stream())
    return err


def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    lib.gdn_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, weight_decay, int(step),
                      float(grad_scale), stream())


def kitti_augment(src, params, train=True):
    """src [B,H,W,C] uint8/float32 as decoded; params int32 [B,5] on the device (or None when not train).
    Returns the normalised NCHW float32 batch."""
    if not src.is_cuda or src.dtype not in (torch.uint8, torch.float32) or src.dim() != 4 or not src.is_contiguous():
        raise GdnError("kitti_augment: src must be a dense [B,H,W,C] uint8/float32 tensor on the GPU")
    B, H, W, C = src.shape
    if train and (params is None or params.dtype != torch.int32 or tuple(params.shape) != (B, 5) or not params.is_cuda):
        raise GdnError("kitti_augment: params must be a device int32 [B,5] tensor")
    dst = torch.empty((B, C, H, W), dtype=torch.float32, device=src.device)
    nb = int(lib.gdn_kitti_augment_workspace_bytes(B))
    ws = workspace(nb, src.device, "augment")
    lib.gdn_kitti_augment(_p(src), 1 if src.dtype == torch.float32 else 0, B, H, W, C, _p(params), 1 if train else 0,
                          _p(dst), _p(ws), nb, stream())
    return dst


def adam_step_dev(p, g, m, v, hyper, state):
    """Capturable Adam: hyper float32[6] and state uint8[32] live on the device (see gdn_adam_step_dev)."""
    lib.gdn_adam_step_dev(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(hyper), _p(state), stream())
