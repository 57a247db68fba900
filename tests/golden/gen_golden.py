#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz from the REAL reference.

Runs only in the build container (it imports /root/reference/src, which does
not exist on the GPU box and never travels there).  It executes the reference's
own Python (model classes, loss helpers, metrics, and the real
``trainer.train_AE_DtoD`` / ``train_AE_RtoD`` loops for one iteration) on torch
CPU with seeded inputs, and stores small input/output vectors.  No reference
source text is stored -- only data.

    python -B tests/golden/gen_golden.py            # writes tests/golden/*.npz

Recipe for importing the reference: stub the absent
third-party modules, never write bytecode into the read-only tree.
"""
import argparse
import hashlib
import json
import os
import pathlib
import sys
import tempfile
import types

sys.dont_write_bytecode = True
HERE = pathlib.Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))

import numpy as np  # noqa: E402
import torch  # noqa: E402

REF = "/root/reference/src"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    class _Dummy:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, n):
            return _Dummy()

        def __call__(self, *a, **k):
            return _Dummy()

    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms")
    tv.utils = _stub("torchvision.utils", save_image=lambda *a, **k: None)
    ip = _stub("IPython", get_ipython=lambda: None)
    ip.display = _stub("IPython.display", clear_output=lambda *a, **k: None, display=lambda *a, **k: None)
    _stub("cv2")
    _stub("path", Path=pathlib.Path)
    _stub("tensorboardX", SummaryWriter=_Dummy)
    _stub("blessings", Terminal=_Dummy)
    _stub("progressbar", ProgressBar=_Dummy)
    try:
        import scipy.misc  # noqa: F401
    except Exception:
        import scipy
        scipy.misc = _stub("scipy.misc")
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.path.insert(0, REF)
    import AE_model_unet
    import utils
    import calculate_error
    import trainer
    return AE_model_unet, utils, calculate_error, trainer


def sd_digest(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(v.detach().contiguous().numpy().tobytes())
    return h.hexdigest()


def tstats(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


def sample64(t):
    f = t.detach().reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, 64).long()
    return f[idx].numpy().copy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=str(HERE))
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    out = pathlib.Path(args.out)
    only = set(args.only.split(",")) if args.only else None

    def want(n):
        return only is None or n in only

    torch.set_num_threads(8)
    AE, U, CE, TR = import_reference()
    from oracle.gdn_oracle import synthetic_batch  # input recipe only

    models = {"AutoEncoder_DtoD": lambda: AE.AutoEncoder_DtoD(input_dim=1),
              "AutoEncoder_2": lambda: AE.AutoEncoder_2(input_dim=3),
              "AutoEncoder": lambda: AE.AutoEncoder()}

    # legacy AutoEncoder.forward calls x.cuda() (AE_model_unet.py:161)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self

    # ---------------- A. seed-exact init ----------------
    if want("init"):
        rec = {}
        for name, ctor in models.items():
            torch.manual_seed(0)
            m = ctor()
            sd = m.state_dict()
            rec[name + ".sha256"] = np.array(sd_digest(sd))
            rec[name + ".keys"] = np.array(json.dumps(list(sd.keys())))
            rec[name + ".shapes"] = np.array(json.dumps([list(v.shape) for v in sd.values()]))
            rec[name + ".stats"] = np.stack([tstats(v.float()) for v in sd.values()])
            rec[name + ".nparams"] = np.array(sum(p.numel() for p in m.parameters()))
        np.savez_compressed(out / "init.npz", **rec)
        print("init done")

    # ---------------- B. full-size forward, B=2 ----------------
    if want("forward"):
        depth, rgb, sparse = synthetic_batch(2, 128, 416, seed=0)
        rec = {}
        for name, ctor in models.items():
            torch.manual_seed(0)
            m = ctor()
            x = depth if name == "AutoEncoder_DtoD" else rgb
            m.train()
            with torch.no_grad():
                feats = m(x, istrain=True)
            rec[name + ".train.out"] = feats[7].numpy().astype(np.float32)
            for i in range(7):
                rec[name + ".train.f%d.stats" % i] = tstats(feats[i])
                rec[name + ".train.f%d.sample" % i] = sample64(feats[i])
                rec[name + ".train.f%d.shape" % i] = np.array(feats[i].shape)
            m.eval()
            with torch.no_grad():
                o = m(x, istrain=False)
            rec[name + ".eval.out"] = o.numpy().astype(np.float32)
            print("forward", name, "done")
        np.savez_compressed(out / "forward.npz", **rec)

    # ---------------- C. real trainer, one iteration, B=2 ----------------
    def run_trainer(mode):
        depth, rgb, sparse = synthetic_batch(2, 128, 416, seed=0)
        loader = [(depth, rgb, sparse)]
        cap = {}
        orig_backward = torch.Tensor.backward

        def rec_backward(self, *a, **k):
            cap["loss"] = self.item()
            return orig_backward(self, *a, **k)

        torch.manual_seed(0)
        if mode == "DtoD":
            model = AE.AutoEncoder_DtoD(input_dim=1)
            G = None
        else:
            model = AE.AutoEncoder_2(input_dim=3)
            torch.manual_seed(1)
            G = AE.AutoEncoder_DtoD(input_dim=1)
            G.eval()
        keys = [k for k, _ in model.named_parameters()]

        def fwd_hook(mod, inp, outp):
            if isinstance(outp, torch.Tensor) and outp.requires_grad and "out" not in cap:
                cap["out"] = outp.detach().clone()
                outp.register_hook(lambda g: cap.__setitem__("dout", g.detach().clone()))
        model.register_forward_hook(fwd_hook)
        opt = torch.optim.Adam(model.parameters(), 2e-5, [0.9, 0.999], eps=1e-08, weight_decay=5e-4)
        a = argparse.Namespace(dataset="KITTI", local_rank=0, save_path=pathlib.Path(tempfile.mkdtemp()),
                               epoch_size=1, batch_size=2, print_freq=10, mode=mode,
                               log_full="full.csv", log_summary="summary.csv")
        TR.save_image_tensor = lambda *a, **k: None
        TR.save_image_batch = lambda *a, **k: None
        TR.ftmap_extract = lambda *a, **k: None
        orig_save = torch.save
        torch.save = lambda *a, **k: None
        torch.Tensor.backward = rec_backward
        cwd = os.getcwd()
        os.chdir(tempfile.mkdtemp())
        ret = None
        try:
            L2, L1 = torch.nn.MSELoss(), torch.nn.L1Loss()
            if mode == "DtoD":
                try:
                    TR.train_AE_DtoD(a, model, L2, L1, opt, loader, None, 2, 1, 2e-5, None, None)
                except NameError as e:   # reference defect at trainer.py:565
                    print("expected reference NameError:", e)
            else:
                ret = TR.train_AE_RtoD(a, model, G, L2, L1, opt, loader, None, 2, 1, 2e-5, None, None)
        finally:
            os.chdir(cwd)
            torch.save = orig_save
            torch.Tensor.backward = orig_backward
        P = dict(model.named_parameters())
        sd = model.state_dict()
        rec = {
            "loss": np.array(cap["loss"], dtype=np.float64),
            "out": cap["out"].numpy().astype(np.float32),
            "dout": cap["dout"].numpy().astype(np.float32),
            "keys": np.array(json.dumps(keys)),
            "grad_norm": np.array([P[k].grad.double().norm().item() for k in keys]),
            "grad_sum": np.array([P[k].grad.double().sum().item() for k in keys]),
            "param_norm_after": np.array([P[k].detach().double().norm().item() for k in keys]),
            "param_sum_after": np.array([P[k].detach().double().sum().item() for k in keys]),
            "bn_keys": np.array(json.dumps([k for k in sd if "running_" in k])),
            "bn_stats_after": np.stack([tstats(sd[k]) for k in sd if "running_" in k]),
        }
        if ret is not None:
            rec["returned"] = np.array([float(r) for r in ret], dtype=np.float64)
        return rec

    if want("train_dtod"):
        np.savez_compressed(out / "train_dtod.npz", **run_trainer("DtoD"))
        print("train_dtod done")
    if want("train_rtod"):
        np.savez_compressed(out / "train_rtod.npz", **run_trainer("RtoD"))
        print("train_rtod done")

    # ---------------- D. loss helpers + metrics ----------------
    if want("losses"):
        g = torch.Generator().manual_seed(7)
        pred = (torch.rand(2, 1, 24, 40, generator=g) * 2 - 1).requires_grad_(True)
        gt = torch.rand(2, 1, 24, 40, generator=g) * 2 - 1
        img = torch.rand(2, 3, 24, 40, generator=g) * 2 - 1
        rec = {"pred": pred.detach().numpy(), "gt": gt.numpy(), "img": img.numpy()}
        l = U.imgrad_loss(pred, gt)
        l.backward()
        rec["imgrad_loss"] = np.array(l.item(), dtype=np.float64)
        rec["imgrad_loss.dpred"] = pred.grad.numpy().copy()
        pred.grad = None
        sm = U.depth_smoothness(pred, img)
        rec["smooth_map"] = sm.detach().numpy()
        ls = torch.mean(torch.abs(0.1 * sm))
        ls.backward()
        rec["smooth_loss"] = np.array(ls.item(), dtype=np.float64)
        rec["smooth_loss.dpred"] = pred.grad.numpy().copy()
        # metrics at full size
        depth, rgb, sparse = synthetic_batch(3, 128, 416, seed=3)
        g2 = torch.Generator().manual_seed(11)
        predm = (depth + 0.3 * (torch.rand(3, 1, 128, 416, generator=g2) - 0.5)).clamp(-1, 1)
        # denser lidar so the valid set is not tiny
        sp = torch.where(torch.rand(3, 1, 128, 416, generator=g2) < 0.3, depth, torch.full_like(depth, -1.0))
        rec["metrics.pred"] = predm.numpy().astype(np.float32)
        rec["metrics.sparse"] = sp.numpy().astype(np.float32)
        rec["metrics.seed_depth"] = np.array(3)
        rec["metrics.errors"] = np.array(CE.compute_errors(sp, depth, predm, crop=True), dtype=np.float64)
        rec["metrics.errors_nocrop"] = np.array(CE.compute_errors(sp, depth, predm, crop=False), dtype=np.float64)
        np.savez_compressed(out / "losses.npz", **rec)
        print("losses done")

    # ---------------- E. block micro-fixtures (full tensors, fwd + bwd) ----------------
    if want("blocks"):
        rec = {}
        cases = [
            ("rb_k9", lambda: AE.ResidualBlock(16, 16, 9, 4), (2, 16, 12, 20)),
            ("rb_k3", lambda: AE.ResidualBlock(32, 32, 3, 1), (2, 32, 8, 26)),
            ("cb_k7s2", lambda: AE.ConvBlock(16, 32, kernel_size=7, stride=2, padding=3), (2, 16, 12, 20)),
            ("cb_k4s2", lambda: AE.ConvBlock(16, 32, kernel_size=4, stride=2, padding=1), (2, 16, 12, 20)),
            ("cb_k5s1", lambda: AE.ConvBlock(32, 16, kernel_size=5, stride=1, padding=2), (2, 32, 8, 12)),
            ("cb_k9c3", lambda: AE.ConvBlock(3, 16, kernel_size=9, stride=1, padding=4), (2, 3, 12, 20)),
            ("cb_k1", lambda: AE.ConvBlock(64, 32, kernel_size=1, stride=1, padding=0), (2, 64, 6, 10)),
            ("ctb_k4s2", lambda: AE.ConvTBlock(32, 16, kernel_size=4, stride=2, padding=1), (2, 32, 6, 10)),
        ]
        for i, (nm, ctor, shp) in enumerate(cases):
            torch.manual_seed(100 + i)
            blk = ctor()
            blk.train()
            for p_ in blk.parameters():          # non-trivial BN affine
                if p_.dim() == 1:
                    p_.data.uniform_(0.5, 1.5)
            x = torch.randn(shp, requires_grad=True)
            y = blk(x)
            dy = torch.randn(y.shape)
            y.backward(dy)
            rec[nm + ".x"] = x.detach().numpy()
            rec[nm + ".y"] = y.detach().numpy()
            rec[nm + ".dy"] = dy.numpy()
            rec[nm + ".dx"] = x.grad.numpy()
            for k, p_ in blk.named_parameters():
                rec[nm + ".p." + k] = p_.detach().numpy()
                rec[nm + ".g." + k] = p_.grad.numpy()
            for k, b_ in blk.named_buffers():
                if "running" in k:
                    rec[nm + ".b." + k] = b_.numpy().copy()
        # bilinear x2 upsample, both conventions (F7)
        xx = torch.randn(1, 2, 5, 7)
        rec["up.x"] = xx.numpy()
        rec["up.ac0"] = torch.nn.functional.interpolate(xx, scale_factor=2, mode="bilinear", align_corners=False).numpy()
        rec["up.ac1"] = torch.nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)(xx).numpy()
        np.savez_compressed(out / "blocks.npz", **rec)
        print("blocks done")


if __name__ == "__main__":
    main()
