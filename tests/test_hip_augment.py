"""GPU parity of the device-side KITTI augmentation (gdn_kitti_augment) against the oracle's restatement of the
reference's host pipeline (oracle/kitti_augment.py, itself pinned bit-exact to Pillow on the CPU): byte work,
so the bar is BIT-EXACT output tensors."""
import numpy as np
import pytest
import torch

from oracle import kitti_augment as K

pytestmark = pytest.mark.gpu


def _run(gpu, arrays, params, train=True):
    from gdn_amd import ops
    src = torch.from_numpy(np.stack(arrays)).to(gpu)
    p = None if params is None else torch.tensor(params, dtype=torch.int32, device=gpu)
    return ops.kitti_augment(src, p, train).cpu().numpy()


@pytest.mark.parametrize("shape", [(128, 416, 3), (128, 416, 1), (37, 53, 3), (16, 24, 1)])
@pytest.mark.parametrize("dtype", ["u8", "f32"])
def test_augment_bit_exact(gpu, shape, dtype):
    H, W, C = shape
    r = np.random.RandomState(H + C)
    B = 6
    if dtype == "u8":
        imgs = [r.randint(0, 256, shape).astype(np.uint8) for _ in range(B)]
    else:       # float data as imread(...).astype(float32) of a 16-bit / arbitrary-range file: exercises bytescale
        imgs = [(r.rand(*shape) * r.uniform(50, 3000) + r.uniform(0, 40)).astype(np.float32) for _ in range(B)]
    py, npr = K.make_rngs(7)
    params = [K.draw_params(H, W, py, npr) for _ in range(B)]
    params[0] = (1, H, W, 0, 0)                 # flip only (no resampling: Pillow returns a copy)
    params[1] = (0, H, int(W * 1.1), 0, min(3, int(W * 1.1) - W))      # horizontal pass only
    params[2] = (1, int(H * 1.15), W, min(2, int(H * 1.15) - H), 0)     # vertical pass only
    got = _run(gpu, imgs, params)
    for b in range(B):
        ref = K.augment_sample([imgs[b]], params[b])[0]
        assert got[b].shape == ref.shape
        assert np.array_equal(got[b], ref), "sample %d params %s: %d elements differ, max %.3e" % (
            b, params[b], int((got[b] != ref).sum()), float(np.abs(got[b] - ref).max()))
    val = _run(gpu, imgs, None, train=False)
    for b in range(B):
        assert np.array_equal(val[b], K.augment_sample([imgs[b]], None, train=False)[0])


def test_augment_many_random_scales(gpu):
    """200 random (scale, offset, flip) draws at one size: the double-precision filter weights computed on the device
    must round exactly like Pillow's on every one of them."""
    H, W = 32, 104
    r = np.random.RandomState(3)
    py, npr = K.make_rngs(11)
    imgs = [r.randint(0, 256, (H, W, 3)).astype(np.uint8) for _ in range(200)]
    params = [K.draw_params(H, W, py, npr) for _ in imgs]
    got = _run(gpu, imgs, params)
    bad = [b for b in range(len(imgs)) if not np.array_equal(got[b], K.augment_sample([imgs[b]], params[b])[0])]
    assert not bad, "samples %s differ" % bad[:10]


def test_gpu_loader_matches_host_pipeline(gpu):
    """GpuAugmentLoader (product) against the oracle run sample by sample with the same seed: same random draws in the
    same order, same tensors; the three tensors of a sample share one set of draws (transform_list.py Compose on a list)."""
    from gdn_amd.datasets import GpuAugmentLoader, SyntheticRawKitti
    ds = SyntheticRawKitti(10, 32, 64, seed=5)
    loader = GpuAugmentLoader(ds, 4, gpu, train=True, seed=9, shuffle=False)
    py, npr = K.make_rngs(9)
    i = 0
    for gt, rgb, sp in loader:
        assert gt.shape[1:] == (1, 32, 64) and rgb.shape[1:] == (3, 32, 64) and gt.dtype == torch.float32 and gt.is_cuda
        for b in range(gt.shape[0]):
            prm = K.draw_params(32, 64, py, npr)
            ref = K.augment_sample(list(ds[i]), prm)
            for t, rf in zip((gt, rgb, sp), ref):
                assert np.array_equal(t[b].cpu().numpy(), rf)
            i += 1
    assert i == 10 and len(loader) == 3
    val = GpuAugmentLoader(ds, 5, gpu, train=False)
    gt, rgb, sp = next(iter(val))
    assert np.array_equal(sp[0].cpu().numpy(), K.augment_sample([ds[0][2]], None, train=False)[0])
    assert float(sp.min()) == -1.0           # empty sparse pixels normalise to exactly -1 (valid_mask = sparse > -1)


def test_sequence_folder_reads_reference_layout(gpu, tmp_path):
    """File layout of datasets_list.py:61-76 (scene/*.jpg, scene/color_gt2/*.png, scene/gt/*.png, train.txt)."""
    from PIL import Image
    from gdn_amd.datasets import GpuAugmentLoader, SequenceFolder
    r = np.random.RandomState(0)
    for scene in ("s1", "s2"):
        (tmp_path / scene / "color_gt2").mkdir(parents=True)
        (tmp_path / scene / "gt").mkdir()
        for i in range(3):
            Image.fromarray(r.randint(0, 256, (16, 24, 3)).astype(np.uint8)).save(tmp_path / scene / ("%07d.jpg" % i))
            Image.fromarray(r.randint(0, 256, (16, 24)).astype(np.uint8)).save(tmp_path / scene / "color_gt2" / ("%07d.png" % i))
            Image.fromarray(r.randint(0, 256, (16, 24)).astype(np.uint8)).save(tmp_path / scene / "gt" / ("%07d.png" % i))
    (tmp_path / "train.txt").write_text("s1\ns2\n")
    (tmp_path / "val.txt").write_text("s2\n")
    import argparse
    ds = SequenceFolder(tmp_path, argparse.Namespace(img_test=False), seed=1, train=True)
    assert len(ds) == 6 and ds[0][0].shape == (16, 24, 1) and ds[0][1].shape == (16, 24, 3)
    batches = list(GpuAugmentLoader(ds, 4, gpu, train=True, seed=2))
    assert [b[0].shape[0] for b in batches] == [4, 2] and batches[0][1].shape == (4, 3, 16, 24)
    assert len(SequenceFolder(tmp_path, argparse.Namespace(img_test=False), train=False)) == 3
