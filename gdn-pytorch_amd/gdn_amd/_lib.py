"""ctypes binding of libgdn_hip.so (the C ABI declared in include/gdn_hip.h).

There is deliberately no CPU fallback: if the HIP library is missing or a call
fails, the product path raises.
"""
import ctypes
import os
import pathlib
from ctypes import POINTER, Structure, c_char_p, c_float, c_int32, c_int64, c_size_t, c_uint64, c_void_p

_PKG = pathlib.Path(__file__).resolve().parent
LIB_PATH = pathlib.Path(os.environ.get("GDN_HIP_LIB", _PKG.parent / "lib" / "libgdn_hip.so"))


class GdnError(RuntimeError):
    pass


class ConvGeom(Structure):
    """gdn_conv_geom (include/gdn_hip.h)."""
    _fields_ = [("B", c_int32), ("H", c_int32), ("W", c_int32), ("Cin", c_int32), ("Cout", c_int32),
                ("k", c_int32), ("stride", c_int32), ("pad", c_int32), ("pad_mode", c_int32),
                ("transposed", c_int32), ("hints", c_int32)]


_PG = POINTER(ConvGeom)
_P = c_void_p
_i32, _i64, _f, _sz = c_int32, c_int64, c_float, c_size_t

# name -> (restype, argtypes); status-returning functions are wrapped to raise.
_SIGS = {
    "gdn_version": (c_int32, []),
    "gdn_strerror": (c_char_p, [c_int32]),
    "gdn_device_info": (c_int32, [c_char_p, c_int32]),
    "gdn_conv_out_dims": (c_int32, [_PG, POINTER(c_int32), POINTER(c_int32)]),
    "gdn_conv_stats_slots": (_i64, [_PG, _i32]),
    "gdn_conv_fwd_workspace_bytes": (_sz, [_PG, _i32]),
    "gdn_conv_fwd": (c_int32, [_PG, _P, _i32, _P, _i32, _i32, _P, _P, _i32, _P, _i32, _P, _P, _P, _i32, _i32, _P, _sz, _P]),
    "gdn_conv_dgrad_workspace_bytes": (_sz, [_PG, _i32]),
    "gdn_conv_dgrad": (c_int32, [_PG, _P, _i32, _P, _P, _i32, _P, _i32, _P, _i32, _P, _i32, _P, _i32, _P, _sz, _i32, _P]),
    "gdn_conv_dgrad_bnb_slots": (_i64, [_PG, _i32]),
    "gdn_conv_wgrad_workspace_bytes": (_sz, [_PG, _i32]),
    "gdn_conv_wgrad": (c_int32, [_PG, _P, _i32, _i32, _P, _i32, _P, _i32, _i32, _P, _sz, _i32, _P]),
    "gdn_conv_wgrad_bf16_workspace_bytes": (_sz, [_PG, _i32, _i32]),
    "gdn_conv_wgrad_bf16": (c_int32, [_PG, _P, _i32, _i32, _P, _i32, _P, _i32, _i32, _P, _sz, _i32, _P]),
    "gdn_fftconv_fwd_workspace_bytes": (_sz, [_PG]),
    "gdn_fftconv_spectrum_bytes": (_sz, [_PG]),
    "gdn_fftconv_stats_slots": (_i64, [_PG]),
    "gdn_fftconv_fwd": (c_int32, [_PG, _P, _i32, _P, _P, _i32, _P, _i32, _P, _P, _P, _i32, _P, _P, _i32, _i32, _P, _P, _sz, _P]),
    "gdn_fftconv_bwd_workspace_bytes": (_sz, [_PG]),
    "gdn_fftconv_bwd": (c_int32, [_PG, _P, _i32, _P, _P, _P, _i32, _P, _i32, _P, _P, _i32, _P, _P, _i32, _P, _i32, _P, _i32, _P,
                                  _i32, _i32, _P, _sz, _P]),
    "gdn_fftconv_bnb_slots": (_i64, [_PG]),
    "gdn_fftconv_cgemm_workspace_bytes": (_sz, [_PG]),
    "gdn_fftconv_cgemm": (c_int32, [_PG, _i32, _P, _sz, _P]),
    "gdn_fftconv_cgemm_shape": (c_int32, [_PG, _P, _P, _P]),
    "gdn_winoconv_fwd_workspace_bytes": (_sz, [_PG]),
    "gdn_winoconv_state_bytes": (_sz, [_PG]),
    "gdn_winoconv_stats_slots": (_i64, [_PG]),
    "gdn_winoconv_fwd": (c_int32, [_PG, _P, _i32, _P, _P, _i32, _P, _i32, _P, _P, _P, _i32, _P, _P, _i32, _i32, _P, _P, _sz, _P]),
    "gdn_winoconv_bwd_workspace_bytes": (_sz, [_PG]),
    "gdn_winoconv_bnb_slots": (_i64, [_PG]),
    "gdn_winoconv_gemm": (c_int32, [_PG, _P, _P, _P, _P]),
    "gdn_winoconv_bwd": (c_int32, [_PG, _P, _i32, _P, _P, _P, _i32, _P, _i32, _P, _P, _i32, _P, _i32, _P, _i32, _P, _sz, _P]),
    "gdn_wino2conv_fwd_workspace_bytes": (_sz, [_PG]),
    "gdn_wino2conv_state_bytes": (_sz, [_PG]),
    "gdn_wino2conv_stats_slots": (_i64, [_PG]),
    "gdn_wino2conv_fwd": (c_int32, [_PG, _P, _i32, _P, _P, _i32, _P, _i32, _P, _P, _P, _i32, _P, _P, _sz, _P]),
    "gdn_wino2conv_bwd_workspace_bytes": (_sz, [_PG]),
    "gdn_wino2conv_bwd": (c_int32, [_PG, _P, _i32, _P, _P, _i32, _P, _P, _i32, _P, _i32, _P, _P, _sz, _P]),
    "gdn_conv_c1_stats_slots": (_i64, [_i32, _i32, _i32]),
    "gdn_conv_c1_fwd": (c_int32, [_P, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _P, _P, _i32, _P, _i32, _P, _P, _P, _i32,
                                  _i32, _P]),
    "gdn_conv_c1_wgrad_workspace_bytes": (_sz, []),
    "gdn_conv_c1_wgrad": (c_int32, [_P, _P, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _P, _P, _sz, _P]),
    "gdn_gemm_x3_packed_bytes": (_sz, [_i32, _i32, _i32]),
    "gdn_gemm_x3_pack": (c_int32, [_P, _P, _i32, _i32, _i32, _P]),
    "gdn_gemm_x3_nt": (c_int32, [_P, _P, _P, _i32, _i32, _i32, _i32, _P]),
    "gdn_gemm_x3_tn": (c_int32, [_P, _P, _P, _i32, _i32, _i32, _i32, _i32, _P]),
    "gdn_gemm_x3_tn_splits": (_i64, [_i32, _i32, _i32, _i32]),
    "gdn_transpose_taps": (c_int32, [_P, _P, _i32, _i32, _i32, _i32, _P]),
    "gdn_cast": (c_int32, [_P, _P, _i64, _i32, _P]),
    "gdn_weight_to_tapmajor": (c_int32, [_P, _P, _i32, _i32, _i32, _i32, _P]),
    "gdn_weight_from_tapmajor": (c_int32, [_P, _P, _i32, _i32, _i32, _i32, _P]),
    "gdn_bn_finalize_train": (c_int32, [_P, _i64, _i32, _i64, _P, _P, _P, _P, _f, _f, _P, _P, _P, _P, _P, _P]),
    "gdn_bn_eval_coeffs": (c_int32, [_P, _P, _P, _P, _f, _i32, _P, _P, _P]),
    "gdn_bn_apply": (c_int32, [_P, _i32, _P, _P, _P, _i32, _P, _i32, _i64, _i32, _i32, _i32, _P]),
    "gdn_bn_bwd_workspace_bytes": (_sz, [_i64, _i32]),
    "gdn_bn_bwd": (c_int32, [_P, _i32, _P, _i32, _P, _P, _P, _P, _P, _P, _i32, _P, _P, _i64, _i32, _i32, _P, _i64, _P, _sz, _i32, _P]),
    "gdn_bn_bwd_coeffs": (c_int32, [_P, _i32, _P, _i32, _P, _P, _P, _P, _P, _P, _P, _i64, _i32, _i32, _P, _i64, _P, _sz, _i32, _P]),
    "gdn_bn_eval_bwd": (c_int32, [_P, _i32, _P, _i32, _P, _P, _P, _i32, _i64, _i32, _i32, _i32, _P]),
    "gdn_bn_apply_up2x": (c_int32, [_P, _P, _P, _P, _P, _P, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _P]),
    "gdn_upsample2x_fwd": (c_int32, [_P, _P, _i32, _i32, _i32, _i32, _i32, _i32, _P]),
    "gdn_upsample2x_bwd": (c_int32, [_P, _P, _i32, _i32, _i32, _i32, _i32, _i32, _P]),
    "gdn_nchw_to_nhwc": (c_int32, [_P, _P, _i32, _i32, _i32, _i32, _i32, _P]),
    "gdn_nhwc_to_nchw": (c_int32, [_P, _P, _i32, _i32, _i32, _i32, _i32, _P]),
    "gdn_add": (c_int32, [_P, _P, _P, _i64, _i32, _P]),
    "gdn_add_pitched": (c_int32, [_P, _i32, _P, _i32, _P, _i32, _i64, _i32, _i32, _P]),
    "gdn_scale_dev": (c_int32, [_P, _P, _P, _i64, _P]),
    "gdn_tanh_bwd": (c_int32, [_P, _P, _P, _i64, _P]),
    "gdn_fill": (c_int32, [_P, _f, _i64, _P]),
    "gdn_loss_workspace_bytes": (_sz, [_i64]),
    "gdn_absdiff_max": (c_int32, [_P, _P, _i64, _P, _P]),
    "gdn_berhu_masked": (c_int32, [_P, _P, _P, _i32, _i32, _i32, _i32, POINTER(c_int32), _P, _P, _P, _P, _sz, _P]),
    "gdn_sobel_l1": (c_int32, [_P, _P, _i32, _i32, _i32, _f, _P, _P, _P, _P, _P, _sz, _P]),
    "gdn_smoothness": (c_int32, [_P, _P, _i32, _i32, _i32, _i32, _P, _P, _P, _P, _P, _P, _sz, _P]),
    "gdn_mse": (c_int32, [_P, _P, _i64, _f, _i32, _P, _P, _sz, _i32, _P]),
    "gdn_kitti_augment_workspace_bytes": (_sz, [_i32]),
    "gdn_kitti_augment": (c_int32, [_P, _i32, _i32, _i32, _i32, _i32, _P, _i32, _P, _P, _sz, _P]),
    "gdn_mse_grad": (c_int32, [_P, _P, _i64, _f, _P, _P, _i32, _P]),
    "gdn_depth_metrics_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "gdn_depth_metrics": (c_int32, [_P, _P, _P, _i32, _i32, _i32, _i32, _P, _P, _sz, _P]),
    "gdn_adam_step": (c_int32, [_P, _P, _P, _P, _i64, _f, _f, _f, _f, _f, _i32, _f, _P]),
    "gdn_adam_step_dev": (c_int32, [_P, _P, _P, _P, _i64, _P, _P, _P]),
    "gdn_clock_probe_arm": (c_int32, [_P, _P]),
    "gdn_clock_probe_watch": (c_int32, [_P, c_uint64, _P]),
    "gdn_clock_probe_stop": (c_int32, [_P, _P]),
}
_STATUS_FUNCS = {n for n, (r, _) in _SIGS.items() if r is c_int32} - {"gdn_version", "gdn_device_info"}

EXPORTS = tuple(_SIGS)
# The C ABI revision these signatures (and ConvGeom's layout) describe: gdn_version() of the library must match exactly --
# a stale build would take the arguments apart differently.
ABI_VERSION = 222


class _Lib:
    def __init__(self):
        self._dll = None

    def _load(self):
        if self._dll is not None:
            return self._dll
        if not LIB_PATH.exists():
            raise GdnError(
                "libgdn_hip.so not found at %s -- build it with `python gdn-pytorch_amd/build.py` "
                "(there is no CPU fallback)" % LIB_PATH)
        # torch's wheel bundles its own libamdhip64 (same SONAME as /opt/rocm's).  It must be in the
        # process BEFORE our library is opened so both bind to ONE HIP runtime; loaded the other way
        # round the two runtimes coexist and every launch on a torch stream fails.
        import torch  # noqa: F401
        dll = ctypes.CDLL(str(LIB_PATH))
        for name, (res, args) in _SIGS.items():
            fn = getattr(dll, name)      # AttributeError if the symbol is missing: fail loudly
            fn.restype = res
            fn.argtypes = args
        have = int(dll.gdn_version())
        if have != ABI_VERSION:
            raise GdnError("%s implements C ABI revision %d, this package binds revision %d -- rebuild it with "
                           "`python gdn-pytorch_amd/build.py --force`" % (LIB_PATH, have, ABI_VERSION))
        self._dll = dll
        return dll

    def raw(self, name):
        return getattr(self._load(), name)

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        fn = getattr(self._load(), name)
        if name in _STATUS_FUNCS:
            def checked(*a, _fn=fn, _name=name):
                rc = _fn(*a)
                if rc != 0:
                    msg = self._dll.gdn_strerror(rc)
                    raise GdnError("%s failed: %s (%d)" % (_name, msg.decode() if msg else "?", rc))
                return rc
            setattr(self, name, checked)
            return checked
        setattr(self, name, fn)
        return fn


lib = _Lib()
