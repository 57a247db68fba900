"""CPU: the oracle's restatement of scipy.misc.imresize (bytescale + Pillow's bilinear resampler) is bit-exact
against the Pillow of this image, over the scale range RandomScaleCrop draws (transform_list.py:185-199)."""
import numpy as np
import pytest

from oracle import kitti_augment as K

PIL = pytest.importorskip("PIL")
from PIL import Image  # noqa: E402


@pytest.mark.parametrize("seed", range(6))
def test_resize_matches_pillow_bit_exact(seed):
    r = np.random.RandomState(seed)
    H, W = (128, 416) if seed < 3 else (37, 53)
    C = (3, 1, 3)[seed % 3]
    img = r.randint(0, 256, (H, W, C) if C > 1 else (H, W)).astype(np.uint8)
    for _ in range(4):
        sy, sx = r.uniform(1, 1.15, 2)
        oh, ow = int(H * sy), int(W * sx)
        ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
        got = K.resize_bilinear_u8(img, oh, ow)
        assert got.shape == ref.shape and np.array_equal(got, ref)
    # one axis unchanged, and the identity
    assert np.array_equal(K.resize_bilinear_u8(img, H, W + 7), np.asarray(Image.fromarray(img).resize((W + 7, H), Image.BILINEAR)))
    assert np.array_equal(K.resize_bilinear_u8(img, H, W), img)


def test_bytescale_and_pipeline_semantics():
    r = np.random.RandomState(1)
    f = (r.rand(16, 24, 3) * 200 + 20).astype(np.float32)
    b = K.bytescale(f)
    assert b.dtype == np.uint8 and b.min() == 0 and b.max() == 255
    u = r.randint(0, 256, (16, 24)).astype(np.uint8)
    assert K.bytescale(u) is u                                   # uint8 passes through
    const = np.full((4, 4), 7.0, np.float32)
    assert np.array_equal(K.bytescale(const), np.zeros((4, 4), np.uint8))
    # draw order and ranges (transform_list.py:158-166, :185-199)
    py, npr = K.make_rngs(3)
    flip, sh, sw, oy, ox = K.draw_params(128, 416, py, npr)
    assert flip in (0, 1) and 128 <= sh <= 147 and 416 <= sw <= 478 and 0 <= oy <= sh - 128 and 0 <= ox <= sw - 416
    out = K.augment_sample([f, u.astype(np.float32)[:, :, None].repeat(1, 2)], K.draw_params(16, 24, py, npr))
    assert out[0].shape == (3, 16, 24) and out[1].shape == (1, 16, 24)
    assert out[0].dtype == np.float32 and -1.0 <= out[0].min() and out[0].max() <= 1.0
    val = K.augment_sample([u], None, train=False)[0]
    assert np.array_equal(val, ((u[None].astype(np.float32) / np.float32(255)) - np.float32(0.5)) / np.float32(0.5))
