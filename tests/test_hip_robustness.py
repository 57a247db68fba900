"""GPU parity tests, edge cases: odd batch sizes, other resolutions, the unmasked (NYU) loss path,
RtoD_single, eval-mode forward with grad enabled, the trainer entry points and validate()."""
import argparse

import numpy as np
import pytest
import torch

from oracle import gdn_oracle as O
from test_hip_kernels import close

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
        assert float((gr - rr).norm()) <= 2e-2 * float(rr.norm()) + 2e-3 * typical, k


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
        assert float((gr - rr).norm()) <= 2e-2 * float(rr.norm()) + 2e-3 * typical, k


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
