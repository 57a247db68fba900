"""Pin the CPU oracle against fixtures produced by the real reference.

The fixtures in tests/golden/*.npz were written by tests/golden/gen_golden.py,
which imports and runs /root/reference/src on torch CPU.  Every oracle function
is checked here; the GPU tests then use the oracle as the live checker.
"""
import hashlib
import json

import numpy as np
import pytest
import torch

from oracle import gdn_oracle as O

torch.set_num_threads(8)


def _digest(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(v.detach().contiguous().numpy().tobytes())
    return h.hexdigest()


@pytest.mark.parametrize("model", ["AutoEncoder_DtoD", "AutoEncoder_2", "AutoEncoder"])
def test_init_seed_exact(golden, model):
    g = golden["init"]
    sd = O.init_state_dict(model, seed=0)
    assert list(sd.keys()) == json.loads(str(g[model + ".keys"]))
    assert [list(v.shape) for v in sd.values()] == json.loads(str(g[model + ".shapes"]))
    nparams = sum(v.numel() for k, v in sd.items() if k.endswith(("weight", "bias")))
    assert nparams == int(g[model + ".nparams"])
    assert _digest(sd) == str(g[model + ".sha256"])


def _feat_check(g, prefix, feats, rtol=2e-4):
    for i in range(7):
        st = g[prefix + ".f%d.stats" % i]
        f = feats[i].double()
        assert list(f.shape) == list(g[prefix + ".f%d.shape" % i])
        got = np.array([f.sum().item(), f.abs().sum().item(), (f * f).sum().item()])
        np.testing.assert_allclose(got[1:], st[1:], rtol=rtol)
        fl = feats[i].reshape(-1)
        idx = torch.linspace(0, fl.numel() - 1, 64).long()
        np.testing.assert_allclose(fl[idx].numpy(), g[prefix + ".f%d.sample" % i], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("model", ["AutoEncoder_DtoD", "AutoEncoder_2", "AutoEncoder"])
def test_forward_full_size(golden, model):
    g = golden["forward"]
    depth, rgb, sparse = O.synthetic_batch(2, 128, 416, seed=0)
    x = depth if model == "AutoEncoder_DtoD" else rgb
    sd = O.init_state_dict(model, seed=0)
    with torch.no_grad():
        feats = O.FORWARD[model](sd, x, istrain=True, training=True)
        np.testing.assert_allclose(feats[7].numpy(), g[model + ".train.out"], rtol=1e-3, atol=1e-5)
        _feat_check(g, model + ".train", feats)
        out = O.FORWARD[model](sd, x, istrain=False, training=False)
    np.testing.assert_allclose(out.numpy(), g[model + ".eval.out"], rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("mode", ["DtoD", "RtoD"])
def test_train_step_vs_real_trainer(golden, mode):
    g = golden["train_dtod" if mode == "DtoD" else "train_rtod"]
    batch = O.synthetic_batch(2, 128, 416, seed=0)
    g_sd = None
    if mode == "DtoD":
        sd = O.init_state_dict("AutoEncoder_DtoD", seed=0)
    else:
        sd = O.init_state_dict("AutoEncoder_2", seed=0)
        g_sd = O.init_state_dict("AutoEncoder_DtoD", seed=1)
    st = {}
    res = O.train_step(mode, sd, batch, st, g_sd=g_sd)
    assert res["loss"] == pytest.approx(float(g["loss"]), rel=1e-5)
    np.testing.assert_allclose(res["outputs"].numpy(), g["out"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(res["dout"].numpy(), g["dout"], rtol=1e-3, atol=1e-9)
    keys = json.loads(str(g["keys"]))
    assert keys == O.trainable_keys(sd)
    gn = np.array([res["grads"][k].double().norm().item() for k in keys])
    np.testing.assert_allclose(gn, g["grad_norm"], rtol=2e-3, atol=1e-7)
    pn = np.array([sd[k].double().norm().item() for k in keys])
    np.testing.assert_allclose(pn, g["param_norm_after"], rtol=1e-6)
    ps = np.array([sd[k].double().sum().item() for k in keys])
    np.testing.assert_allclose(ps, g["param_sum_after"], rtol=1e-4, atol=1e-4)
    bn_keys = json.loads(str(g["bn_keys"]))
    for i, k in enumerate(bn_keys):
        t = sd[k].double()
        np.testing.assert_allclose([t.sum().item(), t.abs().sum().item()], g["bn_stats_after"][i][:2], rtol=1e-4, atol=1e-5)
    if "returned" in g:
        ret = g["returned"]
        assert res["loss"] == pytest.approx(ret[0], rel=1e-5)
        assert res["output_loss"] == pytest.approx(ret[1], rel=1e-5)
        assert res["latent_loss"] == pytest.approx(ret[2], rel=1e-4)


def test_loss_helpers(golden):
    g = golden["losses"]
    pred = torch.from_numpy(g["pred"]).requires_grad_(True)
    gt, img = torch.from_numpy(g["gt"]), torch.from_numpy(g["img"])
    l = O.imgrad_loss(pred, gt)
    l.backward()
    assert l.item() == pytest.approx(float(g["imgrad_loss"]), rel=1e-6)
    np.testing.assert_allclose(pred.grad.numpy(), g["imgrad_loss.dpred"], rtol=1e-5, atol=1e-8)
    pred.grad = None
    sm = O.depth_smoothness(pred, img)
    np.testing.assert_allclose(sm.detach().numpy(), g["smooth_map"], rtol=1e-6, atol=1e-7)
    ls = O.smoothness_loss(pred, img)
    ls.backward()
    assert ls.item() == pytest.approx(float(g["smooth_loss"]), rel=1e-6)
    np.testing.assert_allclose(pred.grad.numpy(), g["smooth_loss.dpred"], rtol=1e-5, atol=1e-9)


def test_metrics(golden):
    g = golden["losses"]
    depth, _, _ = O.synthetic_batch(3, 128, 416, seed=int(g["metrics.seed_depth"]))
    pred, sp = torch.from_numpy(g["metrics.pred"]), torch.from_numpy(g["metrics.sparse"])
    np.testing.assert_allclose(O.compute_errors(sp, depth, pred, crop=True), g["metrics.errors"], rtol=1e-5)
    np.testing.assert_allclose(O.compute_errors(sp, depth, pred, crop=False), g["metrics.errors_nocrop"], rtol=1e-5)


_BLOCKS = {
    "rb_k9": ("rb", 9, 1, 4), "rb_k3": ("rb", 3, 1, 1), "cb_k7s2": ("cb", 7, 2, 3), "cb_k4s2": ("cb", 4, 2, 1),
    "cb_k5s1": ("cb", 5, 1, 2), "cb_k9c3": ("cb", 9, 1, 4), "cb_k1": ("cb", 1, 1, 0), "ctb_k4s2": ("ctb", 4, 2, 1),
}


def block_state(g, nm):
    sd = {}
    for k in g.files:
        if k.startswith(nm + ".p."):
            sd["blk." + k[len(nm) + 3:]] = torch.from_numpy(g[k]).clone()
    for k in list(sd):
        if k.endswith(".bias"):
            c = sd[k].numel()
            base = k[:-4]
            sd[base + "running_mean"] = torch.zeros(c)
            sd[base + "running_var"] = torch.ones(c)
    return sd


def run_block(sd, kind, k, s, p, x):
    if kind == "rb":
        return O.residual_block(x, sd, "blk", k, p, True)
    if kind == "cb":
        return O.conv_block(x, sd, "blk", k, s, p, True)
    return O.convt_block(x, sd, "blk", k, s, p, True)


@pytest.mark.parametrize("nm", sorted(_BLOCKS))
def test_blocks(golden, nm):
    g = golden["blocks"]
    kind, k, s, p = _BLOCKS[nm]
    sd = block_state(g, nm)
    leaves = {kk: v.requires_grad_(True) for kk, v in sd.items() if kk.endswith(("weight", "bias"))}
    x = torch.from_numpy(g[nm + ".x"]).requires_grad_(True)
    y = run_block(sd, kind, k, s, p, x)
    np.testing.assert_allclose(y.detach().numpy(), g[nm + ".y"], rtol=1e-4, atol=1e-5)
    y.backward(torch.from_numpy(g[nm + ".dy"]))
    np.testing.assert_allclose(x.grad.numpy(), g[nm + ".dx"], rtol=1e-3, atol=1e-4)
    for kk, v in leaves.items():
        np.testing.assert_allclose(v.grad.numpy(), g[nm + ".g." + kk[4:]], rtol=1e-3, atol=2e-4)
    for kk in g.files:
        if kk.startswith(nm + ".b."):
            np.testing.assert_allclose(sd["blk." + kk[len(nm) + 3:]].numpy(), g[kk], rtol=1e-5, atol=1e-6)


def test_upsample_conventions(golden):
    g = golden["blocks"]
    x = torch.from_numpy(g["up.x"])
    np.testing.assert_allclose(O._up_ac0(x).numpy(), g["up.ac0"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(O._up_ac1(x).numpy(), g["up.ac1"], rtol=1e-6, atol=1e-7)
