"""GPU parity tests, edge cases: odd batch sizes, other resolutions, the unmasked (NYU) loss path,
RtoD_single, eval-mode forward with grad enabled, the trainer entry points and validate()."""
import argparse

import numpy as np
import pytest
import torch

from oracle import gdn_oracle as O
from test_hip_kernels import close, close_abs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,B,H,W", [
    ("AutoEncoder_DtoD", 1, 16, 16),       # smallest legal input: level 4 is 1x1
    ("AutoEncoder_DtoD", 3, 48, 80),       # H, W multiples of 16 but not of 32; odd batch
    ("AutoEncoder_2", 1, 32, 48),
    ("AutoEncoder_2", 5, 16, 32),
    ("AutoEncoder", 2, 48, 64),
])
def test_forward_other_shapes(gpu, name, B, H, W):
    import gdn_amd.AE_model_unet as M
    if name == "AutoEncoder_DtoD" and H == 16:
        pytest.skip("reflection pad needs pad < size at every level: 16x16 is rejected by torch too")
    depth, rgb, _ = O.synthetic_batch(B, H, W, seed=21)
    x = depth if name == "AutoEncoder_DtoD" else rgb
    sd = O.init_state_dict(name, seed=5)
    with torch.no_grad():
        ref = O.FORWARD[name]({k: v.clone() for k, v in sd.items()}, x, istrain=True, training=True, height=H, width=W)
    m = getattr(M, name)(height=H, width=W)
    m.load_state_dict(sd)
    m = m.to(gpu).train()
    with torch.no_grad():
        got = m(x.to(gpu), istrain=True)
    for i in range(8):
        close(got[i], ref[i], rtol=2e-3, atol_scale=2e-3, what="%s %dx%dx%d f%d" % (name, B, H, W, i))


def test_dtod_step_unmasked_and_odd_batch(gpu):
    """NYU-style step: no sparse tensor -> unmasked BerHu (trainer.py:420-421,445), batch 3."""
    import gdn_amd.AE_model_unet as M
    from gdn_amd import utils as U
    depth, rgb, _ = O.synthetic_batch(3, 32, 64, seed=8)
    sd = O.init_state_dict("AutoEncoder_DtoD", seed=2)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.endswith(("weight", "bias"))}
    work = {k: v.clone() for k, v in sd.items()}
    work.update(leaves)
    out_ref = O.forward_dtod(work, depth, istrain=False, training=True)
    lo = O.berhu_masked(out_ref, depth, None) + 3 * O.imgrad_loss(out_ref, depth)
    lo.backward()
    model = M.AutoEncoder_DtoD(height=32, width=64)
    model.load_state_dict(sd)
    model = model.to(gpu).train()
    out = model(depth.to(gpu), istrain=False)
    loss, _, _ = U.dtod_loss(out, depth.to(gpu), None)
    loss.backward()
    assert loss.item() == pytest.approx(lo.item(), rel=1e-3)
    typical = float(np.median([v.grad.double().norm().item() for v in leaves.values()]))
    for k, p in model.named_parameters():
        gr, rr = p.grad.detach().cpu().double(), leaves[k].grad.double()
        # (3 images at 32x64 with train-mode BN: the most rounding-sensitive configuration in the suite)
        assert float((gr - rr).norm()) <= 3e-2 * float(rr.norm()) + 3e-3 * typical, k


def test_rtod_single_step_vs_oracle(gpu):
    """mode RtoD_single: BerHu + smoothness, no guide (trainer.py:698,727)."""
    import gdn_amd.AE_model_unet as M
    from gdn_amd import utils as U
    batch = O.synthetic_batch(2, 32, 64, seed=9)
    depth, rgb, sparse = batch
    sd = O.init_state_dict("AutoEncoder_2", seed=4)
    ref = O.train_step("RtoD_single", {k: v.clone() for k, v in sd.items()}, batch, {})
    model = M.AutoEncoder_2(height=32, width=64)
    model.load_state_dict(sd)
    model = model.to(gpu).train()
    out = model(rgb.to(gpu), istrain=False)
    pix, ol, sm = U.rtod_pixel_loss(out, depth.to(gpu), rgb.to(gpu), sparse.to(gpu))
    pix.backward()
    assert pix.item() == pytest.approx(ref["loss"], rel=1e-3)
    assert ol.item() == pytest.approx(ref["output_loss"], rel=1e-3)
    assert sm.item() == pytest.approx(ref["smoothness_loss"], rel=1e-3)
    typical = float(np.median([g.double().norm().item() for g in ref["grads"].values()]))
    for k, p in model.named_parameters():
        gr, rr = p.grad.detach().cpu().double(), ref["grads"][k].double()
        # (3 images at 32x64 with train-mode BN: the most rounding-sensitive configuration in the suite)
        assert float((gr - rr).norm()) <= 3e-2 * float(rr.norm()) + 3e-3 * typical, k


def test_eval_forward_with_grad_enabled_and_second_backward(gpu):
    import gdn_amd.AE_model_unet as M
    from gdn_amd._lib import GdnError
    torch.manual_seed(0)
    m = M.AutoEncoder_DtoD(height=32, width=64).to(gpu)
    x = torch.rand(1, 1, 32, 64, device=gpu) * 2 - 1
    m.eval()
    y = m(x)                                  # eval-mode BN, grad enabled: forward must work
    with torch.no_grad():
        assert torch.equal(y, m(x))
    with pytest.raises(GdnError):
        y.sum().backward()                    # ... only the backward is unsupported
    m.train()
    y = m(x)
    y.sum().backward(retain_graph=True)
    with pytest.raises(GdnError):
        y.sum().backward()                    # the tape was consumed


def test_trainer_entry_points_and_validate(gpu, tmp_path, monkeypatch):
    """train_AE_DtoD / train_AE_RtoD / validate with the reference's signatures on synthetic loaders."""
    import gdn_amd.AE_model_unet as M
    from gdn_amd import trainer as T
    from gdn_amd.optim import Adam
    from gdn_amd.synthetic import SyntheticLoader
    monkeypatch.chdir(tmp_path)
    H, W = 32, 64
    args = argparse.Namespace(dataset="KITTI", epoch_size=2, batch_size=2, mode="DtoD", print_freq=10)
    loader = SyntheticLoader(2, 2, H, W, seed=0, device=gpu, distinct=2)
    torch.manual_seed(0)
    G = M.AutoEncoder_DtoD(height=H, width=W).to(gpu)
    opt = Adam(G.parameters(), 2e-5, [0.9, 0.999], eps=1e-08, weight_decay=5e-4)
    w0 = G.downconv1.main[1].weight.detach().clone()
    loss = T.train_AE_DtoD(args, G, None, None, opt, loader, loader, 2, 1, 2e-5, None, None)
    assert torch.isfinite(loss) and not torch.equal(w0, G.downconv1.main[1].weight)
    ck = list(tmp_path.glob("KITTI_AE_DtoD_trained_model*/epoch_1_AE_depth_loss_*.pkl"))
    assert len(ck) == 1                       # same directory / file naming as trainer.py:344,555
    errs, min_errs, names = T.validate(args, loader, G.eval(), 0, None, "DtoD")
    assert names == ['abs_diff', 'abs_rel', 'sq_rel', 'a1', 'a2', 'a3', 'rmse', 'rmse_log'] and len(errs) == 8
    assert all(np.isfinite(errs)) and all(np.isfinite(min_errs))
    args.mode = "RtoD"
    R = M.AutoEncoder_2(height=H, width=W).to(gpu)
    optR = Adam(R.parameters(), 2e-5, [0.9, 0.999], eps=1e-08, weight_decay=5e-4)
    out = T.train_AE_RtoD(args, R, G, None, None, optR, loader, None, 2, 1, 2e-5, None, None)
    assert len(out) == 3 and all(torch.isfinite(t) for t in out)     # (loss, output_loss, latent_loss), :922
    args.mode = "RtoD_single"
    out = T.train_AE_RtoD(args, R, None, None, None, optR, loader, None, 2, 1, 2e-5, None, None)
    assert float(out[2]) == 0.0


def test_cli_synthetic_smoke(gpu, tmp_path, monkeypatch):
    """python -m gdn_amd.GDN_main DATA --mode DtoD --synthetic ... with the reference's flags."""
    from gdn_amd import GDN_main, option
    monkeypatch.chdir(tmp_path)
    a = option.parse_args(["synthetic", "--mode", "DtoD", "--synthetic", "--batch_size", "2", "--epochs", "1",
                           "--epoch_size", "2", "--height", "32", "--width", "64", "--gpu_num", "0"])
    loss = GDN_main.run(a)
    assert torch.isfinite(loss)


def test_latent_grad_vs_oracle(gpu):
    """--latent_grad: d(latent)/d(outputs) through the frozen eval-mode guide, and from there
    into the trained network's parameters, against the oracle's autograd through the same graph."""
    import gdn_amd.AE_model_unet as M
    from gdn_amd import trainer as T
    from gdn_amd import utils as U
    from gdn_amd._lib import GdnError
    H, W = 32, 64
    depth, rgb, sparse = O.synthetic_batch(2, H, W, seed=6)
    sd = O.init_state_dict("AutoEncoder_2", seed=5)
    g_sd = O.init_state_dict("AutoEncoder_DtoD", seed=7)
    ref = O.train_step("RtoD", {k: v.clone() for k, v in sd.items()}, (depth, rgb, sparse), {},
                       g_sd={k: v.clone() for k, v in g_sd.items()}, latent_grad=True)
    ref0 = O.train_step("RtoD", {k: v.clone() for k, v in sd.items()}, (depth, rgb, sparse), {},
                        g_sd={k: v.clone() for k, v in g_sd.items()}, latent_grad=False)
    assert (ref["dout"] - ref0["dout"]).abs().max() > 1e-6          # the extension changes the gradient
    R = M.AutoEncoder_2(input_dim=3, height=H, width=W)
    R.load_state_dict(sd)
    G = M.AutoEncoder_DtoD(input_dim=1, height=H, width=W)
    G.load_state_dict(g_sd)
    R, G = R.to(gpu).train(), G.to(gpu).eval()
    d, r, s = depth.to(gpu), rgb.to(gpu), sparse.to(gpu)
    out = R(r, istrain=False)
    with pytest.raises(GdnError):                                    # the guide must be frozen
        T.guide_latent_loss(G, d, out, latent_grad=True)
    G.requires_grad_(False)
    # (1) the guide alone, at the oracle's own estimate: d(latent)/d(outputs) == dout(latent_grad) - dout(value only)
    for faithful in (False, True):
        xo = ref["outputs"].to(gpu).requires_grad_(True)
        lat = T.guide_latent_loss(G, d, xo, faithful=faithful, latent_grad=True)
        lat.backward()
        assert lat.item() == pytest.approx(ref["latent_loss"], rel=1e-3)
        close(xo.grad, ref["dout"] - ref0["dout"], rtol=5e-3, atol_scale=5e-3, what="d latent / d outputs")
    # (2) end to end: the trained network's estimate differs from the oracle's by its own rounding (amplified by
    # train-mode BN at this tiny size), so the total gradient is compared at the looser bar of the train-step tests
    R.zero_grad()
    out = R(r, istrain=False)
    out.retain_grad()
    lat = T.guide_latent_loss(G, d, out, latent_grad=True)
    pix, ol, sm = U.rtod_pixel_loss(out, d, r, s)
    (pix + lat).backward()
    assert lat.item() == pytest.approx(ref["latent_loss"], rel=5e-3)
    close(out.grad, ref["dout"], rtol=5e-2, atol_scale=3e-2, what="dL/dout with latent grad", outliers=2e-3)
    typical = float(np.median([ref["grads"][k].double().norm().item() for k, _ in R.named_parameters()]))
    worst = 0.0
    for k, p in R.named_parameters():
        gr, rr = p.grad.detach().cpu().double(), ref["grads"][k].double()
        rel = float((gr - rr).norm() / (rr.norm() + 1e-2 * typical))
        worst = max(worst, rel)
        assert rel < 5e-2, "%s: relative gradient error %.3e" % (k, rel)
    print("latent_grad end to end: worst relative parameter-gradient error %.3e" % worst)
    assert all(p.grad is None for p in G.parameters())


@pytest.mark.parametrize("name", ["AutoEncoder_DtoD", "AutoEncoder_2"])
def test_input_gradient_vs_oracle(gpu, name):
    """d(out)/d(input) of the whole network (1- and 3-channel inputs: N<4 data gradient + scalar reflect fold)."""
    import gdn_amd.AE_model_unet as M
    H, W = 32, 64
    depth, rgb, _ = O.synthetic_batch(1, H, W, seed=8)
    x = depth if name == "AutoEncoder_DtoD" else rgb
    sd = O.init_state_dict(name, seed=9)
    xr = x.clone().requires_grad_(True)
    out_ref = O.FORWARD[name]({k: v.clone() for k, v in sd.items()}, xr, istrain=False, training=False)
    gy = torch.randn(out_ref.shape, generator=torch.Generator().manual_seed(1))
    out_ref.backward(gy)
    model = getattr(M, name)(input_dim=x.shape[1], height=H, width=W)
    model.load_state_dict(sd)
    model = model.to(gpu).eval().requires_grad_(False)
    xg = x.to(gpu).requires_grad_(True)
    out = model(xg, istrain=False)
    close_abs(out, out_ref, 1e-3, what=name + " eval forward")
    out.backward(gy.to(gpu))
    # outliers: a first-layer pre-activation within rounding of zero takes the other side of its ReLU under a different (equally
    # valid) summation order -- one such flip moves the 9 x 9 x C input gradients under that output by ~1 % of the largest one
    # (seen with the patch-staged 3-channel first-layer kernel: 62 of 6144 elements in one 9 x 9 block, outputs equal to 1.3e-7)
    close(xg.grad, xr.grad, rtol=5e-3, atol_scale=5e-3, what=name + " d out / d input", outliers=2e-2)
    close(xg.grad, xr.grad, rtol=5e-2, atol_scale=5e-2, what=name + " d out / d input, every element")


def test_cli_augment_latent_grad_bf16_smoke(gpu, tmp_path, monkeypatch):
    """RtoD from the CLI with every opt-in of this build at once: raw samples through the GPU augmentation,
    --latent_grad through the frozen guide, bf16 compute."""
    from gdn_amd import GDN_main, option
    monkeypatch.chdir(tmp_path)
    a = option.parse_args(["synthetic", "--mode", "RtoD", "--synthetic", "--augment", "--latent_grad", "--dtype", "bf16",
                           "--batch_size", "2", "--epochs", "1", "--epoch_size", "2", "--height", "64", "--width", "128",
                           "--gpu_num", "0", "--model_dir", "none.pkl"])
    out = GDN_main.run(a)
    assert len(out) == 3 and all(torch.isfinite(t) for t in out) and float(out[2].detach()) > 0.0


def test_cli_kitti_directory(gpu, tmp_path, monkeypatch):
    """DtoD from a directory laid out like the reference's KITTI root (datasets_list.py:61-76)."""
    from PIL import Image
    from gdn_amd import GDN_main, option
    r = np.random.RandomState(0)
    root = tmp_path / "kitti"
    for scene in ("s1", "s2"):
        (root / scene / "color_gt2").mkdir(parents=True)
        (root / scene / "gt").mkdir()
        for i in range(2):
            Image.fromarray(r.randint(0, 256, (32, 64, 3)).astype(np.uint8)).save(root / scene / ("%07d.jpg" % i))
            Image.fromarray(r.randint(0, 256, (32, 64)).astype(np.uint8)).save(root / scene / "color_gt2" / ("%07d.png" % i))
            Image.fromarray(r.randint(0, 256, (32, 64)).astype(np.uint8)).save(root / scene / "gt" / ("%07d.png" % i))
    (root / "train.txt").write_text("s1\ns2\n")
    (root / "val.txt").write_text("s2\n")
    monkeypatch.chdir(tmp_path)
    a = option.parse_args([str(root), "--mode", "DtoD", "--batch_size", "2", "--epochs", "1", "--height", "32",
                           "--width", "64", "--gpu_num", "0", "-e"])
    loss = GDN_main.run(a)
    assert torch.isfinite(loss)


def test_legacy_inference_at_config4_resolution(gpu):
    """BASELINE configs[4]'s path at its own resolution (256x832, eval-mode BN folded into the conv epilogues), one
    image: fp32 against the oracle at the north-star bar, bf16 against the oracle's bf16 emulation (eval mode has no
    batch statistics to amplify rounding, so the comparison stays meaningful end to end)."""
    import gdn_amd.AE_model_unet as M
    H, W = 256, 832
    _, rgb, _ = O.synthetic_batch(1, H, W, seed=31)
    sd = O.init_state_dict("AutoEncoder", seed=6)
    with torch.no_grad():
        ref = O.forward_legacy({k: v.clone() for k, v in sd.items()}, rgb, istrain=True, training=False, height=H, width=W)
        with O.bf16_emulation():
            emu = O.forward_legacy({k: v.clone() for k, v in sd.items()}, rgb, istrain=True, training=False, height=H, width=W)
    m = M.AutoEncoder(height=H, width=W)
    m.load_state_dict(sd)
    m = m.to(gpu).eval()
    got = m(rgb.to(gpu), istrain=True)
    for i in range(8):
        close(got[i], ref[i], rtol=1e-3, atol_scale=1e-3, what="legacy 256x832 fp32 f%d" % i)
    got16 = m.compute_dtype("bf16")(rgb.to(gpu), istrain=True)
    assert got16[0].dtype == torch.bfloat16 and got16[7].dtype == torch.float32
    for i in range(8):
        e = float((got16[i].float().cpu() - emu[i]).norm() / (emu[i].norm() + 1e-30))
        assert e < 2e-2, "legacy 256x832 bf16 f%d: rel L2 %.3e vs the bf16 emulation" % (i, e)


def test_legacy_inference_b64_graph_matches_single_image(gpu, monkeypatch):
    """BASELINE configs[4] at its OWN batch (depth_extract.py:113-147's network, 256x832, B = 64, one hipGraph replay): the
    9x9 layers' activations are 3.5 GB and each of their spectra ~6.9 GB, so every kernel of the eval path (frequency-domain
    transforms and per-bin GEMMs, Winograd, the direct kernels) indexes past 4 GiB.  Eval-mode BatchNorm has no cross-sample
    coupling, so (a) two copies of one image in slots 0 and 63 must come out BITWISE equal inside the B = 64 replay, and
    (b) images 0 and 63 must be BITWISE equal to the same images run alone at B = 1 -- with GDN_PLAN_BATCH=64 for that run,
    because the library picks tile sizes / split factors from the batch size and a different plan is a different summation
    order (without the override the B = 1 run agrees to rounding, also checked)."""
    import gdn_amd.AE_model_unet as M
    H, W, B = 256, 832, 64
    sd = O.init_state_dict("AutoEncoder", seed=6)
    gen = torch.Generator().manual_seed(77)
    x = torch.rand(B, 3, H, W, generator=gen) * 2 - 1
    x[B - 2] = x[1]                                        # a duplicate pair (slots 1 and 62) far apart in memory
    x = x.to(gpu)

    def build():
        m = M.AutoEncoder(height=H, width=W)
        m.load_state_dict(sd)
        return m.to(gpu).eval()

    m = build()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        m(x, istrain=False)                                # warm-up: workspaces allocated outside the capture
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = m(x, istrain=False)
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    assert torch.equal(out[1], out[B - 2]), "the same image in slots 1 and 62 of one B=64 replay differs"
    assert not torch.equal(out[0], out[1])
    big = {b: out[b].clone() for b in (0, 1, B - 1)}
    del g, out, m
    torch.cuda.empty_cache()
    # alone, default plans: equal to rounding
    m1 = build()
    for b in (0, B - 1):
        o1 = m1(x[b:b + 1].contiguous(), istrain=False)
        close_abs(o1[0], big[b], 1e-4, what="B=64 image %d vs alone (default B=1 plans)" % b)
    del m1
    # alone, with the B = 64 plans: bitwise
    monkeypatch.setenv("GDN_PLAN_BATCH", str(B))
    m2 = build()
    for b in (0, B - 1):
        o2 = m2(x[b:b + 1].contiguous(), istrain=False)
        assert torch.equal(o2[0], big[b]), "image %d: B=64 graph replay vs B=1 (same plans) max diff %.3e" % (
            b, float((o2[0] - big[b]).abs().max()))


def test_legacy_instance_norm_eval(gpu):
    """AutoEncoder(norm='Instance') (AE_model_unet.py:146-155): InstanceNorm2d(affine, track_running_stats) in eval()
    normalises with the running statistics; checked against the same layers executed by torch on the CPU."""
    import gdn_amd.AE_model_unet as M
    import torch.nn as nn
    H, W = 32, 64
    torch.manual_seed(3)
    m = M.AutoEncoder(norm='Instance', height=H, width=W)
    for mod in m.modules():                       # non-trivial running statistics and affine parameters
        if isinstance(mod, (nn.InstanceNorm2d, nn.BatchNorm2d)):
            mod.running_mean.normal_(0, 0.2); mod.running_var.uniform_(0.5, 1.5)
            mod.weight.data.uniform_(0.5, 1.5); mod.bias.data.normal_(0, 0.2)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    _, rgb, _ = O.synthetic_batch(2, H, W, seed=41)
    with torch.no_grad():       # the oracle's legacy graph: F.batch_norm(training=False) == eval InstanceNorm with tracked stats
        ref = O.forward_legacy(sd, rgb, istrain=False, training=False, height=H, width=W)
        cb = nn.Sequential(nn.Conv2d(3, 64, 9, 1, 4, bias=False), nn.InstanceNorm2d(64, affine=True, track_running_stats=True)).eval()
        cb[0].weight.copy_(sd["downconv0.weight"])
        for k in ("weight", "bias", "running_mean", "running_var"):
            getattr(cb[1], k).copy_(sd["N64_down." + k])
        x3_torch = torch.relu(cb(rgb))
    m = m.to(gpu).eval()
    out = m(rgb.to(gpu), istrain=False)
    close_abs(out, ref, 1e-3, what="legacy InstanceNorm eval output")
    blk = M.ConvBlock(3, 64, kernel_size=9, stride=1, padding=4, norm='Instance')
    with torch.no_grad():
        blk.main[1].weight.copy_(sd["downconv0.weight"])
        for k in ("weight", "bias", "running_mean", "running_var"):
            getattr(blk.main[2], k).copy_(sd["N64_down." + k])
        refb = torch.relu(cb[1](torch.nn.functional.conv2d(torch.nn.functional.pad(rgb, (4, 4, 4, 4), mode="reflect"), sd["downconv0.weight"])))
    blk = blk.to(gpu).eval()
    with torch.no_grad():
        close(blk(rgb.to(gpu)), refb, what="ConvBlock(norm='Instance') eval")
    assert x3_torch.shape == (2, 64, H, W)


@pytest.mark.parametrize("kind", ["ConvBlock", "ConvTBlock"])
def test_standalone_instance_norm_blocks_train(gpu, kind):
    """ConvBlock / ConvTBlock(norm='Instance') in train mode (AE_model_unet.py:70-75, :88-92): per-instance statistics,
    affine, tracked running statistics (batch mean of the per-instance mean / unbiased variance), backward -- against the same
    layers executed by torch on the CPU."""
    import copy
    import gdn_amd.AE_model_unet as M
    import torch.nn as nn
    torch.manual_seed(11)
    if kind == "ConvBlock":
        blk = M.ConvBlock(64, 128, kernel_size=3, stride=2, padding=1, norm='Instance')
        ref = nn.Sequential(nn.ReflectionPad2d(1), nn.Conv2d(64, 128, 3, 2, padding=0, bias=False),
                            nn.InstanceNorm2d(128, affine=True, track_running_stats=True), nn.ReLU())
        conv_i, norm_i = 1, 2
    else:
        blk = M.ConvTBlock(64, 64, kernel_size=4, stride=2, padding=1, norm='Instance')
        ref = nn.Sequential(nn.ConvTranspose2d(64, 64, 4, 2, 1, bias=False),
                            nn.InstanceNorm2d(64, affine=True, track_running_stats=True), nn.ReLU())
        conv_i, norm_i = 0, 1
    nb, nr = blk.main[norm_i], ref[norm_i]
    with torch.no_grad():
        nb.weight.uniform_(0.5, 1.5); nb.bias.normal_(0, 0.2)
        nb.running_mean.normal_(0, 0.2); nb.running_var.uniform_(0.5, 1.5)
        ref[conv_i].weight.copy_(blk.main[conv_i].weight)
        for k in ("weight", "bias", "running_mean", "running_var"):
            getattr(nr, k).copy_(getattr(nb, k))
    x = torch.randn(3, 64, 12, 20, generator=torch.Generator().manual_seed(5))
    gy_seed = torch.Generator().manual_seed(6)
    xr = x.clone().requires_grad_(True)
    ref.train()
    yr = ref(xr)
    gy = torch.randn(yr.shape, generator=gy_seed)
    yr.backward(gy)
    blk = blk.to(gpu).train()
    xg = x.to(gpu).requires_grad_(True)
    yg = blk(xg)
    close(yg, yr, what="train-mode output")
    yg.backward(gy.to(gpu))
    close(xg.grad, xr.grad, what="input gradient")
    close(blk.main[conv_i].weight.grad, ref[conv_i].weight.grad, what="conv weight gradient")
    close(nb.weight.grad, nr.weight.grad, what="gamma gradient")
    close(nb.bias.grad, nr.bias.grad, what="beta gradient")
    close(nb.running_mean, nr.running_mean, what="running mean")
    close(nb.running_var, nr.running_var, what="running var")
    assert int(nb.num_batches_tracked) == int(nr.num_batches_tracked)
    # and eval() afterwards uses the updated running statistics
    blk.eval(); ref.eval()
    with torch.no_grad():
        close(blk(x.to(gpu)), ref(x), what="eval after the update")


def test_depth_extract_cli(gpu, tmp_path):
    """depth_extract.py counterpart: reference-format checkpoint (module.-prefixed legacy AutoEncoder) + image folder ->
    one depth PNG per image at the source resolution; the batched run equals the one-by-one run."""
    from PIL import Image
    import gdn_amd.AE_model_unet as M
    from gdn_amd import depth_extract, trainer
    r = np.random.RandomState(0)
    imgs = tmp_path / "imgs"
    imgs.mkdir()
    for i, (h, w) in enumerate(((90, 300), (128, 416), (60, 200))):
        Image.fromarray(r.randint(0, 256, (h, w, 3)).astype(np.uint8)).save(imgs / ("im%d.png" % i))
    torch.manual_seed(0)
    m = M.AutoEncoder(height=64, width=128).to(gpu)
    ck = tmp_path / "legacy.pkl"
    trainer._save_checkpoint(m, str(ck))
    assert all(k.startswith("module.") for k in torch.load(ck))
    depth_extract.main(["--model_dir", str(ck), "--img_dir", str(imgs), "--out_dir", str(tmp_path / "o1"),
                        "--height", "64", "--width", "128", "--batch", "1"])
    depth_extract.main(["--model_dir", str(ck), "--img_dir", str(imgs), "--out_dir", str(tmp_path / "o3"),
                        "--height", "64", "--width", "128", "--batch", "3"])
    for i, (h, w) in enumerate(((90, 300), (128, 416), (60, 200))):
        a = np.asarray(Image.open(tmp_path / "o1" / ("im%d_depth.png" % i)))
        b = np.asarray(Image.open(tmp_path / "o3" / ("im%d_depth.png" % i)))
        assert a.shape == (h, w) and a.dtype == np.uint16 and a.std() > 0
        assert np.array_equal(a, b)          # eval-mode BN: no cross-sample coupling, batching changes nothing


def test_bench_contract_two_ranks(gpu):
    """bench.py under torch.distributed.run with two ranks (test hooks: gloo + both ranks on cuda:0, since RCCL refuses
    two ranks per device): barrier + max-over-ranks timing, one JSON line from rank 0 with the contract's fields."""
    import json
    import os
    import pathlib
    import subprocess
    import sys
    root = pathlib.Path(__file__).resolve().parent.parent
    env = dict(os.environ, GDN_DIST_BACKEND="gloo", GDN_SINGLE_DEVICE="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29513", str(root / "bench.py"), "--gpus", "2",
                        "--steps", "2", "--warmup", "1", "--batch", "4", "--no-roofline"],
                       capture_output=True, text=True, env=env, timeout=600, cwd=str(root))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config"):
        assert k in rec, k
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["scaling"] == "weak" and rec["value"] > 0
    assert rec["config"]["global_batch"] == 8 and "cpu_baseline" not in rec


def test_bench_json_line_is_last_on_stdout_with_rccl(gpu):
    """With RCCL initialised (a forced 1-rank group: the same code path as N > 1) the library prints its version banner to C
    stdout, block-buffered when stdout is a pipe; bench.py flushes it before the result line, so the LAST line of stdout is
    the JSON record the driver reads."""
    import json
    import os
    import pathlib
    import subprocess
    import sys
    root = pathlib.Path(__file__).resolve().parent.parent
    env = dict(os.environ, GDN_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", RANK="0", LOCAL_RANK="0",
               WORLD_SIZE="1")
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "4",
                        "--no-roofline", "--no-cpu-baseline", "--no-other-configs"],
                       capture_output=True, text=True, env=env, timeout=600, cwd=str(root))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    rec = json.loads(lines[-1])
    assert rec["n_gpus"] == 1 and rec["value"] > 0
    assert sum(ln.startswith("{") for ln in lines) == 1


@pytest.mark.parametrize("mode,dtype", [("DtoD", "fp32"), ("RtoD", "bf16")])
def test_bench_gpus_2_self_launch_trains_two_ranks(gpu, mode, dtype):
    """`python bench.py --gpus 2` as ONE plain command on the GPU: the parent starts two fresh rank processes (it never
    touches the GPU itself), they train data-parallel with the bucketed gradient all-reduce overlapped with backward, and
    rank 0's JSON record -- n_gpus 2, rccl_ranks 2, global batch 2 x B -- is the last line of the parent's stdout.  On a
    1-GPU box both ranks share cuda:0 over gloo (GDN_SINGLE_DEVICE / GDN_DIST_BACKEND test hooks: RCCL refuses two ranks on
    one device); on a multi-GPU node the same command runs RCCL over xGMI.  RtoD bf16 is BASELINE configs[3]'s per-rank
    workload; without the hooks a 1-GPU box must refuse --gpus 2 with the device count in the message."""
    import json
    import os
    import pathlib
    import subprocess
    import sys
    root = pathlib.Path(__file__).resolve().parent.parent
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2",
           "--mode", mode, "--dtype", dtype, "--no-roofline", "--no-cpu-baseline", "--no-other-configs"]
    if torch.cuda.device_count() < 2:
        r = subprocess.run(cmd, capture_output=True, text=True, env=base, timeout=300, cwd=str(root))
        assert r.returncode == 2 and "1 GPU(s) are visible" in r.stderr and r.stdout.strip() == ""
        env = dict(base, GDN_SINGLE_DEVICE="1", GDN_DIST_BACKEND="gloo")
        backend = "gloo"
    else:
        env, backend = base, "nccl"
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=str(root))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    rec = json.loads(lines[-1])
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["dist_backend"] == backend and rec["value"] > 0
    assert rec["config"]["global_batch"] == 4 and rec["config"]["parallelism"] == "dp2" and rec["scaling"] == "weak"
    assert rec["dtype"] == ("f32" if dtype == "fp32" else "bf16") and mode in rec["config"]["workload"]
    assert sum(ln.startswith("{") for ln in lines) == 1
    # the multi-rank record explains itself: per-rank step spread, exposed all-reduce time, bucketing, switches
    dp = rec["data_parallel"]
    assert dp["overlap"] is True and dp["syncs"] == 2 and dp["buckets"] >= 2 and dp["bytes_reduced"] > 4 * 60e6
    assert len(dp["step_ms_per_rank"]["all"]) == 2 and dp["step_ms_per_rank"]["min"] <= dp["step_ms_per_rank"]["max"]
    assert dp["allreduce_exposed_ms"] is not None and dp["allreduce_exposed_ms"] >= 0.0 and len(dp["allreduce_exposed_ms_per_rank"]) == 2
    shared = torch.cuda.device_count() < 2
    assert rec["config"]["shared_gpu_ranks"] == (2 if shared else 1) and rec["config"]["x3"] is (not shared)
    # ... and --no-overlap gives the curve to read it against: one whole-arena all-reduce after backward
    r2 = subprocess.run(cmd + ["--no-overlap"], capture_output=True, text=True, env=env, timeout=900, cwd=str(root))
    assert r2.returncode == 0, r2.stderr[-3000:]
    dp2 = json.loads([ln for ln in r2.stdout.splitlines() if ln.strip()][-1])["data_parallel"]
    assert dp2["overlap"] is False and dp2["buckets_in_flight_before_backward_returned"] == 0 and abs(dp2["bytes_reduced"] - dp["bytes_reduced"]) < 0.01 * dp["bytes_reduced"]


def test_shader_clock_probe(gpu):
    """gdn_clock_probe_*: one sleeping wave on a second stream brackets a window of launches on the current stream and returns
    shader cycles / 100 MHz ticks: a plausible engine clock (0.3-2.6 GHz on MI355X), the window at least as long as the event
    time of the launches; the side stream is one that really runs beside the current stream (chosen by trial: late in a long
    pytest process a fresh stream may share the current stream's hardware queue, and a watcher there blocks what it should
    bracket); a window whose stop never comes ends by the tick limit (reported, no hang)."""
    from gdn_amd import ops
    x = torch.randn(8, 128, 416, 64, device=gpu)
    sc, sh = torch.ones(64, device=gpu), torch.zeros(64, device=gpu)
    y = torch.empty_like(x)
    for _ in range(3):
        ops.bn_apply(x, sc, sh, True, None, out=y)
    torch.cuda.synchronize()
    junk = [torch.cuda.Stream() for _ in range(13)]          # (push the round-robin of hardware queues along)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    clk = ops.ShaderClock(gpu)
    assert clk.side is not None, "no stream runs beside the current one"
    with clk:
        e0.record()
        for _ in range(50):
            ops.bn_apply(x, sc, sh, True, None, out=y)
        e1.record()
    torch.cuda.synchronize()
    cyc, ticks, ended = clk.read()
    ghz = clk.ghz()
    print("clock probe: %d cycles / %d ticks -> %.3f GHz over %.3f ms of launches" % (cyc, ticks, ghz or -1, e0.elapsed_time(e1)))
    assert ended and ghz is not None and 0.3 < ghz < 2.6
    assert ticks * 1e-5 >= 0.9 * e0.elapsed_time(e1)                 # 100 MHz ticks -> ms
    # no stop: the watcher gives up at its tick limit (20 ms here) and says so
    clk2 = ops.ShaderClock(gpu, max_s=0.02)
    clk2.buf.zero_()
    ops.lib.gdn_clock_probe_arm(clk2.buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
    clk2.side.wait_stream(torch.cuda.current_stream())
    ops.lib.gdn_clock_probe_watch(clk2.buf.data_ptr(), clk2.max_ticks, clk2.side.cuda_stream)
    torch.cuda.synchronize()
    cyc, ticks, ended = clk2.read()
    assert not ended and 2_000_000 <= ticks < 4_000_000 and clk2.ghz() is None
    del junk


def _no_error_keys(o, path=""):
    """Every 'error' / '*_error' key anywhere in a bench record (there must be none)."""
    bad = []
    if isinstance(o, dict):
        for k, v in o.items():
            if k == "error" or k.endswith("_error"):
                bad.append(path + "/" + k)
            bad += _no_error_keys(v, path + "/" + k)
    elif isinstance(o, list):
        for i, v in enumerate(o):
            bad += _no_error_keys(v, "%s[%d]" % (path, i))
    return bad


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_bench_gpus_2_default_flags_rooflines_on(gpu, launcher):
    """The command the driver's scaling run takes: `bench.py --gpus 2` WITHOUT --no-roofline.  After the
    closing barrier the other rank leaves the process group; rank 0 must not run another training step (it holds the gradient
    all-reduce -- under RCCL a collective issued alone never returns, under gloo it raises): one JSON line, no 'error' anywhere
    in it, the breakdown marked as an N = 1 property, a live `roofline`, and the tail after the timed region short.  1-GPU box:
    gloo hooks, both ranks on cuda:0; >= 2 GPUs: RCCL, one rank per device."""
    import json
    import os
    import pathlib
    import subprocess
    import sys
    import time
    root = pathlib.Path(__file__).resolve().parent.parent
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    multi = torch.cuda.device_count() >= 2
    env = base if multi else dict(base, GDN_SINGLE_DEVICE="1", GDN_DIST_BACKEND="gloo")
    tail = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2"]
    if launcher == "self":
        cmd = [sys.executable, str(root / "bench.py"), *tail]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29523", str(root / "bench.py"), *tail]
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=str(root))
    wall = time.time() - t0
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert sum(ln.startswith("{") for ln in lines) == 1
    rec = json.loads(lines[-1] if launcher == "self" else [ln for ln in lines if ln.startswith("{")][0])
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["dist_backend"] == ("nccl" if multi else "gloo") and rec["value"] > 0
    assert _no_error_keys(rec) == [], _no_error_keys(rec)
    assert "skipped" in rec["step_kernel_breakdown"] and "cpu_baseline" not in rec and "other_configs" not in rec
    rf = rec["roofline"]
    assert rf["bound"] == "mfma" and rf["achieved"] > 0 and 0 < rf["frac"] < 1 and "cgemm_bins_kernel" in rf["kernel"]
    assert rec["tail_s"] < 60.0 and wall < 600.0, (rec["tail_s"], wall)
    assert "parity_bar" in rec["config"]


def test_bench_forced_rccl_one_rank_rooflines_on(gpu):
    """A forced 1-rank RCCL group (GDN_FORCE_DIST=1: the same reducer / stream-ordering path as N > 1) with the rooflines ON:
    at world size 1 the profiler pass over two more steps runs (its all-reduce is this rank's own), and the record carries the
    family shares over ALL symbols plus the step's largest symbol; no 'error' anywhere."""
    import json
    import os
    import pathlib
    import subprocess
    import sys
    root = pathlib.Path(__file__).resolve().parent.parent
    env = dict(os.environ, GDN_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29527", RANK="0", LOCAL_RANK="0",
               WORLD_SIZE="1")
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "4",
                        "--no-cpu-baseline", "--no-other-configs"],
                       capture_output=True, text=True, env=env, timeout=900, cwd=str(root))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    rec = json.loads(lines[-1])
    assert rec["n_gpus"] == 1 and rec["dist_backend"] == "nccl" and rec["value"] > 0
    assert _no_error_keys(rec) == [], _no_error_keys(rec)
    brk = rec["step_kernel_breakdown"]
    assert brk["symbols"] > len(brk["top"]) and brk["top_symbol"] == brk["top"][0]["symbol"]
    assert abs(sum(brk["families"].values()) - 1.0) < 1e-2 and "cgemm" in brk["families"]
    for k in ("roofline", "roofline_direct3x3", "roofline_fftconv"):
        assert rec[k]["achieved"] > 0, k
    assert "largest single symbol" in rec["roofline"]["chosen_by"]


def _diag_child(script, args, timeout=600):
    """Run tests/diag/<script> child <args> in a fresh process; returns its `R <hash> <losses>` line."""
    import pathlib
    import subprocess
    import sys
    root = pathlib.Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "tests" / "diag" / script), "child", *[str(a) for a in args]],
                       capture_output=True, text=True, timeout=timeout, cwd=str(root))
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("R ")]
    assert r.returncode == 0 and lines, r.stderr[-2000:]
    return lines[-1]


@pytest.mark.gpu
def test_training_does_not_read_unwritten_global_memory(gpu):
    """Four tiny DtoD training steps (frequency-domain, both Winograd forms and their bf16 x 3 GEMMs, direct kernels,
    BatchNorm, losses, fused Adam) give the SAME BITS when every torch.empty / empty_like / new_empty device allocation
    of the process is pre-filled with NaNs (0xFF) or finite junk (0x42): no kernel consumes memory nobody wrote -- which
    is what 'results that change from run to run' would otherwise have to be checked against first."""
    clean = _diag_child("poison_alloc.py", [-1, 4, 32, 64, 2, "DtoD"])
    assert "nan" not in clean.lower()
    for byte in (0xFF, 0x42):
        assert _diag_child("poison_alloc.py", [byte, 4, 32, 64, 2, "DtoD"]) == clean, "fill 0x%02x" % byte


@pytest.mark.gpu
def test_training_does_not_read_lds_left_by_other_workgroups(gpu):
    """The same steps with the whole LDS of every CU filled with NaNs before every library call
    (tests/diag/lds_fill.hip, built here with hipcc): same bits as the clean run."""
    import pathlib
    import shutil
    import subprocess
    root = pathlib.Path(__file__).resolve().parent.parent
    so = root / "tests" / "diag" / "_build" / "liblds_fill.so"
    if not so.exists():
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        so.parent.mkdir(exist_ok=True)
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", str(so),
                            str(root / "tests" / "diag" / "lds_fill.hip")], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
    clean = _diag_child("lds_poison.py", [-1, 2, 32, 64, 2])
    assert _diag_child("lds_poison.py", [0xFFFFFFFF, 2, 32, 64, 2]) == clean


def _lds_fill_lib():
    """tests/diag/lds_fill.hip as a shared object (built here with hipcc when the snapshot does not carry it)."""
    import ctypes
    import pathlib
    import shutil
    import subprocess
    root = pathlib.Path(__file__).resolve().parent.parent
    so = root / "tests" / "diag" / "_build" / "liblds_fill.so"
    if not so.exists():
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        so.parent.mkdir(exist_ok=True)
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", str(so),
                            str(root / "tests" / "diag" / "lds_fill.hip")], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
    lf = ctypes.CDLL(str(so))
    lf.lds_fill.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
    return lf


@pytest.mark.gpu
def test_bf16_lds_dma_kernels_do_not_read_stale_lds(gpu):
    """The LDS-DMA kernels of the bf16 path (conv_ring_bf16, conv_ring2_bf16, wgrad_ring_bf16) with the whole LDS of every CU
    filled with NaN bit patterns before each launch: same bits as the clean launch, on the shapes that walk every staging path
    (round 5: a ring row whose width is 8 mod 16 was read 8 positions past what the launch had written -- stale values times a
    zero gradient, harmless until the stale value is a NaN)."""
    from gdn_amd import ops
    from test_hip_bf16 import RING_WGRAD_CASES
    lf = _lds_fill_lib()
    sink = torch.zeros(4, dtype=torch.int32, device=gpu)
    g = torch.Generator(device=gpu).manual_seed(0)

    def poisoned(fn):
        ref = fn()
        for pat in (0xFFFFFFFF, 0x7F007F00):
            assert lf.lds_fill(pat, sink.data_ptr(), ops.stream()) == 0
            out = fn()
            for a, b in zip(ref if isinstance(ref, tuple) else (ref,), out if isinstance(out, tuple) else (out,)):
                assert torch.equal(a, b), "LDS pattern %#x changes the result" % pat
    for name, ci, co, k, p, refl, B, H, W in RING_WGRAD_CASES + [("k9_b3_128x416", 64, 64, 9, 4, False, 3, 128, 416)]:
        op = ops.Conv(ci, co, k, 1, p, reflect=refl)
        Ho, Wo = H + 2 * p - k + 1, W + 2 * p - k + 1
        x = torch.randn(B, H, W, ci, device=gpu, generator=g).bfloat16()
        gy = torch.randn(B, Ho, Wo, co, device=gpu, generator=g).bfloat16()

        def wgrad():
            dw = torch.zeros(k * k, co, ci, device=gpu)
            op.wgrad(x, gy, dw, cfg=4)
            return dw
        poisoned(wgrad)
        if p == k // 2:
            w = (torch.randn(k * k, co, ci, device=gpu, generator=g) * 0.05).bfloat16()
            wt = ops.transpose_taps(w)
            for cfg in (10, 11, 12):
                poisoned(lambda: op.fwd(x, w, stats=True, tile_cfg=cfg))
                if not refl:
                    poisoned(lambda: op.dgrad(gy, wt, (H, W), addsrc=x, tile_cfg=cfg))
