"""Run under LD_PRELOAD=<libclang_rt.asan> with GDN_HIP_LIB=<lib/asan/libgdn_hip_asan.so> by
tests/test_abi_cpu.py::test_host_planning_code_under_asan: drives every host-side query of the C ABI (geometry, plan,
workspace and slot sizes, error strings) over the networks' layer shapes, ragged shapes and invalid geometries.  Any
heap / stack / global out-of-bounds access or use-after-free in the host planning code aborts the process (ASan)."""
import ctypes
import itertools
import sys

from gdn_amd._lib import ConvGeom, lib

n = 0
ho, wo = ctypes.c_int32(), ctypes.c_int32()
layers = [(1, 64, 9, 1, 4, 1, 0), (64, 64, 9, 1, 4, 0, 0), (64, 128, 4, 2, 1, 1, 0), (128, 128, 7, 1, 3, 0, 0),
          (128, 256, 5, 2, 2, 1, 0), (256, 256, 5, 1, 2, 0, 0), (256, 512, 3, 2, 1, 1, 0), (512, 512, 3, 1, 1, 0, 0),
          (512, 512, 3, 1, 1, 1, 0), (512, 256, 4, 2, 1, 0, 1), (64, 1, 9, 1, 4, 0, 1), (64, 1, 9, 1, 4, 0, 0),
          (1024, 512, 1, 1, 0, 0, 0), (3, 64, 9, 1, 4, 1, 0), (256, 128, 5, 1, 2, 0, 1), (96, 80, 3, 1, 1, 0, 0)]
sizes = [(20, 128, 416), (20, 8, 26), (2, 16, 52), (1, 1, 2), (3, 33, 47), (64, 256, 832), (1, 4, 4), (5, 2, 3)]
for (ci, co, k, s, p, pm, tr), (B, H, W) in itertools.product(layers, sizes):
    g = ConvGeom(B, H, W, ci, co, k, s, p, pm, tr)
    r = ctypes.byref(g)
    try:
        lib.gdn_conv_out_dims(r, ctypes.byref(ho), ctypes.byref(wo))
    except Exception:
        pass
    for cfg in (0, 1, 2, 3, 0x800, 0x200, 0x10000, 0x10000 | 1, 0x10000 | 0x800):
        lib.gdn_conv_stats_slots(r, cfg)
        lib.gdn_conv_fwd_workspace_bytes(r, cfg)
        lib.gdn_conv_dgrad_workspace_bytes(r, cfg)
    for cx in (ci, max(1, ci // 2), 64, 1):
        lib.gdn_conv_wgrad_workspace_bytes(r, cx)
        for cfg in (0, 1, 2):
            lib.gdn_conv_wgrad_bf16_workspace_bytes(r, cx, cfg)
    for f in ("gdn_fftconv_fwd_workspace_bytes", "gdn_fftconv_spectrum_bytes", "gdn_fftconv_stats_slots",
              "gdn_fftconv_bwd_workspace_bytes", "gdn_fftconv_bnb_slots", "gdn_winoconv_fwd_workspace_bytes",
              "gdn_winoconv_state_bytes", "gdn_winoconv_stats_slots", "gdn_winoconv_bwd_workspace_bytes",
              "gdn_winoconv_bnb_slots"):
        try:
            getattr(lib, f)(r)
        except Exception:
            pass
    n += 1
# degenerate / hostile geometries: must be rejected, not crash
for vals in [(0, 0, 0, 0, 0, 0, 0, 0, 0, 0), (1, 1, 1, 1, 1, 99, 1, 0, 0, 0), (1, 8, 8, 64, 64, 3, 0, 1, 0, 0),
             (-1, 8, 8, 64, 64, 3, 1, 1, 0, 0), (2, 8, 8, 64, 64, 3, 1, 9, 1, 0), (2, 2**15, 2**15, 64, 64, 9, 1, 4, 0, 0)]:
    g = ConvGeom(*vals)
    r = ctypes.byref(g)
    for f in ("gdn_conv_fwd_workspace_bytes", "gdn_conv_dgrad_workspace_bytes"):
        getattr(lib, f)(r, 0)
    for f in ("gdn_fftconv_fwd_workspace_bytes", "gdn_fftconv_bwd_workspace_bytes", "gdn_winoconv_fwd_workspace_bytes",
              "gdn_winoconv_bwd_workspace_bytes", "gdn_winoconv_state_bytes", "gdn_fftconv_spectrum_bytes"):
        getattr(lib, f)(r)
    lib.gdn_conv_wgrad_workspace_bytes(r, 64)
    try:
        lib.gdn_conv_out_dims(r, ctypes.byref(ho), ctypes.byref(wo))
    except Exception:
        pass
for bnp in ((20 * 128 * 416, 64), (1, 4), (0, 0), (10**9, 512)):
    lib.gdn_bn_bwd_workspace_bytes(*bnp)
for npx in (0, 1, 20 * 128 * 416, 10**9):
    lib.gdn_loss_workspace_bytes(npx)
for b, h, w in ((20, 128, 416), (1, 1, 1), (0, 0, 0)):
    lib.gdn_depth_metrics_workspace_bytes(b, h, w)
    lib.gdn_kitti_augment_workspace_bytes(b)
for code in range(-8, 3):
    lib.gdn_strerror(code)
assert lib.gdn_version() >= 210
print("asan host queries ok: %d geometries" % n)
sys.exit(0)
