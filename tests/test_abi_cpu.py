"""CPU-side checks: the C-ABI library builds/loads and exports every symbol that
include/gdn_hip.h declares; host logic that needs no GPU; the product path refuses
to run without a GPU instead of falling back."""
import pathlib
import re

import pytest
import torch

REPO = pathlib.Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def built_lib():
    import importlib.util
    spec = importlib.util.spec_from_file_location("gdn_build", REPO / "gdn-pytorch_amd" / "build.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build()


def test_header_symbols_exported(built_lib):
    import ctypes
    hdr = (REPO / "include" / "gdn_hip.h").read_text()
    names = sorted(set(re.findall(r"\b(gdn_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 30
    dll = ctypes.CDLL(str(built_lib))
    missing = [n for n in names if not hasattr(dll, n)]
    assert not missing, missing
    from gdn_amd._lib import EXPORTS
    assert sorted(EXPORTS) == names      # the ctypes binding covers exactly the header


def test_host_queries_without_gpu(built_lib):
    import ctypes
    from gdn_amd._lib import ConvGeom, lib
    assert lib.gdn_version() >= 100
    assert lib.gdn_strerror(-3) == b"workspace missing or too small"
    g = ConvGeom(20, 16, 52, 512, 512, 3, 1, 1, 0, 0)
    g4 = ConvGeom(20, 8, 26, 512, 512, 3, 1, 1, 0, 0)
    ho, wo = ctypes.c_int32(), ctypes.c_int32()
    lib.gdn_conv_out_dims(ctypes.byref(g), ctypes.byref(ho), ctypes.byref(wo))
    assert (ho.value, wo.value) == (16, 52)
    gt = ConvGeom(20, 8, 26, 512, 512, 4, 2, 1, 0, 1)          # ConvTranspose2d k4 s2 p1
    lib.gdn_conv_out_dims(ctypes.byref(gt), ctypes.byref(ho), ctypes.byref(wo))
    assert (ho.value, wo.value) == (16, 52)
    assert lib.gdn_conv_stats_slots(ctypes.byref(g), 0x800 | 1) == (20 * 16 * 52 + 127) // 128
    assert lib.gdn_conv_stats_slots(ctypes.byref(g), 0x800) == (20 * 16 * 52 + 63) // 64  # 64x64 tiles, single stage
    assert lib.gdn_conv_stats_slots(ctypes.byref(g4), 0) == (20 * 8 * 26 + 15) // 16       # split-K: 16 pixels per combine block
    assert lib.gdn_conv_stats_slots(ctypes.byref(gt), 0x800 | 3) == 4 * ((20 * 8 * 26 + 63) // 64)
    gr = ConvGeom(2, 16, 24, 64, 128, 7, 2, 3, 1, 0)            # reflect-padded strided conv
    assert lib.gdn_conv_dgrad_workspace_bytes(ctypes.byref(gr), 0x800) == 2 * 22 * 30 * 64 * 4   # 0x800: no split-K
    assert lib.gdn_conv_dgrad_workspace_bytes(ctypes.byref(g), 0x800) == 0
    # small launches are split over the filter taps: partial slabs [ksplit][pixels][Cout]
    assert lib.gdn_conv_fwd_workspace_bytes(ctypes.byref(g4), 0) == 3 * 20 * 8 * 26 * 512 * 4      # 9 taps split 3,3,3
    assert lib.gdn_conv_fwd_workspace_bytes(ctypes.byref(g), 0) == 0
    assert lib.gdn_conv_fwd_workspace_bytes(ctypes.byref(g), 0x800) == 0
    assert lib.gdn_conv_wgrad_workspace_bytes(ctypes.byref(g), 512) > 0
    bad = ConvGeom(1, 4, 4, 8, 8, 11, 1, 5, 0, 0)               # 11x11 > 81 taps
    assert lib.gdn_conv_wgrad_workspace_bytes(ctypes.byref(bad), 8) == 0
    with pytest.raises(Exception):
        lib.gdn_conv_out_dims(ctypes.byref(bad), ctypes.byref(ho), ctypes.byref(wo))
    # round-4 host plans: Winograd F(4x4,3x3) where the image rounds up to whole 4x4 tiles with <= 15 % more pixels, the
    # GDN_HINT_NO_WINO_F4 switch, the reduction GEMM's split model, the data gradient's BatchNorm-backward slots
    st4, st2 = lib.gdn_winoconv_state_bytes(ctypes.byref(g)), lib.gdn_winoconv_state_bytes(ctypes.byref(ConvGeom(20, 16, 52, 512, 512, 3, 1, 1, 0, 0, 4)))
    v = lambda bins, tiles: bins * tiles * 512 * 4
    u = 36 * 512 * 512 * 6                                      # the data gradient's weight set: 36 bf16 x 3 panel sets
    assert st4 == v(36, 20 * 4 * 13) + u and st2 == v(16, 20 * 8 * 26) + 16 * 512 * 512 * 6
    assert lib.gdn_winoconv_stats_slots(ctypes.byref(g)) == 20 * 4 * 13 // 4
    g9 = ConvGeom(2, 9, 13, 128, 256, 3, 1, 1, 0, 0)            # 12 x 16 tiles for 9 x 13 pixels: +64 % -> F(2x2,3x3)
    assert lib.gdn_winoconv_stats_slots(ctypes.byref(g9)) == (2 * 5 * 7 + 3) // 4
    assert lib.gdn_gemm_x3_tn_splits(36, 1040, 512, 512) == 2    # 576 tiles on 512 slots: two rounds whole, 1.5 in halves
    assert lib.gdn_gemm_x3_tn_splits(16, 4160, 512, 512) == 2    # 256 tiles: half the chip idle unless split
    assert lib.gdn_gemm_x3_tn_splits(36, 280, 512, 512) == 1     # chunks of at least 256 rows
    assert lib.gdn_gemm_x3_tn_splits(16, 30800, 128, 256) == 16  # 32 tiles over 30800 rows
    assert lib.gdn_gemm_x3_tn_splits(16, 1040, 96, 512) == 0     # not a multiple of 128: not eligible
    assert lib.gdn_conv_dgrad_bnb_slots(ctypes.byref(ConvGeom(20, 16, 52, 512, 512, 3, 1, 1, 0, 0)), 0x10000) > 0   # bf16 ring layer
    assert lib.gdn_conv_dgrad_bnb_slots(ctypes.byref(g), 0) == 0                                                    # fp32
    assert lib.gdn_conv_dgrad_bnb_slots(ctypes.byref(ConvGeom(2, 16, 24, 64, 128, 4, 2, 1, 0, 0)), 0x10000) == 0   # stride 2
    # frequency-domain backward: the data gradient's inverse kernels index channels by shifts (64 / 128 / 256 input channels);
    # any other multiple of 64 is forward-only and the workspace query says so (ops.fft_ok then picks another path)
    f64, f192 = ConvGeom(2, 32, 64, 64, 64, 5, 1, 2, 0, 0), ConvGeom(2, 32, 64, 192, 192, 5, 1, 2, 0, 0)
    assert lib.gdn_fftconv_spectrum_bytes(ctypes.byref(f64)) > 0 and lib.gdn_fftconv_bwd_workspace_bytes(ctypes.byref(f64)) > 0
    assert lib.gdn_fftconv_spectrum_bytes(ctypes.byref(f192)) > 0 and lib.gdn_fftconv_bwd_workspace_bytes(ctypes.byref(f192)) == 0


def test_models_match_oracle_init_and_keys():
    """Host mirror: same state_dict keys, shapes and seed-exact values as the (reference-pinned) oracle."""
    import gdn_amd.AE_model_unet as M
    from oracle import gdn_oracle as O
    for name in ("AutoEncoder_DtoD", "AutoEncoder_2", "AutoEncoder"):
        torch.manual_seed(0)
        m = getattr(M, name)()
        sd, ref = m.state_dict(), O.init_state_dict(name, seed=0)
        assert list(sd) == list(ref)
        assert all(torch.equal(sd[k], ref[k]) for k in ref)


def test_no_cpu_fallback():
    import gdn_amd.AE_model_unet as M
    from gdn_amd import utils as U
    from gdn_amd._lib import GdnError
    m = M.AutoEncoder_DtoD(height=32, width=64)
    with pytest.raises((GdnError, RuntimeError)):
        m(torch.zeros(1, 1, 32, 64))
    with pytest.raises((GdnError, RuntimeError)):
        U.imgrad_loss(torch.zeros(1, 1, 8, 8, requires_grad=True), torch.zeros(1, 1, 8, 8))
    # the reference's R and G only print the norm (AE_model_unet.py:266-269): 'Instance' still builds BatchNorm blocks
    r = M.AutoEncoder_2(norm='Instance', height=32, width=64)
    assert isinstance(r.downconv1.main[2], torch.nn.BatchNorm2d)
    # the legacy net and standalone blocks do instantiate InstanceNorm layers (:73, :91, :136-155); same state_dict keys
    leg = M.AutoEncoder(norm='Instance', height=32, width=64)
    assert isinstance(leg.N64_down, torch.nn.InstanceNorm2d) and "N64_down.running_var" in leg.state_dict()
    assert isinstance(M.ConvBlock(3, 8, 3, 1, norm='Instance').main[2], torch.nn.InstanceNorm2d)


def test_tap_major_arena_layout():
    """Parameters keep their logical torch shape while living tap-major in one flat arena."""
    import gdn_amd.AE_model_unet as M
    from gdn_amd import engine as E
    torch.manual_seed(0)
    blk = M.ConvTBlock(8, 4, kernel_size=4, stride=2, padding=1)
    ref = {k: v.clone() for k, v in blk.state_dict().items()}
    ar = E.ParamArena(blk, torch.device("cpu"))
    assert ar.intact()
    for k, v in blk.state_dict().items():
        assert torch.equal(v, ref[k])                       # logical values unchanged
    w = blk.main[0].weight
    tv = E.tap_view(w.data, True)                           # [16, Cout=4, Cin=8], contiguous view of the arena
    assert tv is not None and tv.shape == (16, 4, 8) and tv.data_ptr() == ar.data.data_ptr()
    assert torch.equal(tv[5, 2, 3], ref["main.0.weight"][3, 2, 1, 1])
    ar.bind_grads()
    assert w.grad.shape == w.shape and E.tap_view(w.grad, True) is not None
    # a standard-layout optimizer still works on the strided views
    opt = torch.optim.Adam(blk.parameters(), 1e-3)
    for p in blk.parameters():
        p.grad.fill_(1.0)
    opt.step()
    assert ar.intact()


def test_save_path_and_options():
    from gdn_amd import option
    from gdn_amd.utils import crop_box_kitti, save_path_formatter
    a = option.parse_args(["/data/kitti", "--mode", "RtoD", "--batch_size", "20", "--gpu_num", "0,1"])
    assert a.mode == "RtoD" and a.height == 128 and a.width == 416 and a.lr == 2e-5 and a.beta == 0.999
    sp = str(save_path_formatter(a, option.parser))
    assert sp.startswith("kitti,b20/")
    assert crop_box_kitti(128, 416) == (52, 126, 14, 401)      # SURVEY 3.1 [probed]


def test_host_planning_code_under_asan():
    """SURVEY 5 (sanitizers): the host side of the C ABI -- geometry, tile / split-K / segment plans, workspace and slot
    sizes, argument checks -- built with AddressSanitizer (build_asan.py, `build.py --asan`: host objects instrumented; GPU ASan is not
    available on this pool) and driven over every layer shape of the three networks, ragged shapes and invalid
    geometries.  ASan aborts the child on any out-of-bounds access or use-after-free."""
    import importlib.util
    import os
    import subprocess
    import sys
    recipe = REPO / "gdn-pytorch_amd" / "build_asan.py"
    if not recipe.exists():
        pytest.skip("build_asan.py is a CPU-side tool (not shipped to GPU boxes)")
    spec = importlib.util.spec_from_file_location("gdn_build_asan", recipe)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lib = mod.build_asan()
    rt = mod.asan_runtime()
    assert rt and os.path.exists(rt), "clang ASan runtime not found"
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", GDN_HIP_LIB=str(lib),
               PYTHONPATH=os.pathsep.join([str(REPO / "gdn-pytorch_amd"), str(REPO)]))
    r = subprocess.run([sys.executable, str(REPO / "tests" / "asan_host_queries.py")], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "asan host queries ok" in r.stdout, r.stdout[-2000:] + r.stderr[-6000:]
    assert "AddressSanitizer" not in r.stderr


def test_loader_rejects_a_library_of_another_abi_revision(built_lib, monkeypatch):
    """The ctypes binding describes one revision of the C ABI (argument lists, gdn_conv_geom's layout); a stale build must be
    refused at load time instead of taking the arguments apart differently."""
    import gdn_amd._lib as L
    assert L.lib.gdn_version() == L.ABI_VERSION
    fresh = L._Lib()
    monkeypatch.setattr(L, "ABI_VERSION", L.ABI_VERSION + 1)
    with pytest.raises(L.GdnError, match="revision"):
        fresh.gdn_version()


def test_plan_override_hints_refuse_out_of_range(monkeypatch):
    """GDN_PLAN_BATCH / GDN_RING_CUS are 8-bit fields of gdn_conv_geom.hints; a value that does not fit must be an
    error, not a silently different plan (256 -> no override, 300 -> 44)."""
    import pytest
    from gdn_amd import ops
    for k in ("GDN_PLAN_BATCH", "GDN_RING_CUS", "GDN_FFT_NP"):
        monkeypatch.delenv(k, raising=False)
    assert ops.plan_override_hints() == 0
    monkeypatch.setenv("GDN_PLAN_BATCH", "64")
    monkeypatch.setenv("GDN_RING_CUS", "16")
    assert ops.plan_override_hints() == (64 << 8) | (2 << 16)
    for bad in ("0", "256", "300"):
        monkeypatch.setenv("GDN_PLAN_BATCH", bad)
        with pytest.raises(ops.GdnError):
            ops.plan_override_hints()
    monkeypatch.setenv("GDN_PLAN_BATCH", "255")
    for bad in ("4", "12", "2048", "4096"):
        monkeypatch.setenv("GDN_RING_CUS", bad)
        with pytest.raises(ops.GdnError):
            ops.plan_override_hints()
