"""CPU restatement (numpy) of the reference's KITTI training-time augmentation -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this; the product path is
gdn_amd.datasets on the HIP kernel `gdn_kitti_augment`.

Reference path (file:line in /root/reference/src):
  GDN_main.py:57-62      Compose([RandomHorizontalFlip(), RandomScaleCrop(), ArrayToTensor(H, W), normalize])
  transform_list.py:158-166  RandomHorizontalFlip   random.random() < 0.5 -> np.fliplr on every image of the sample
  transform_list.py:185-199  RandomScaleCrop        x_scaling, y_scaling = np.random.uniform(1, 1.15, 2);
                                                    scaled = int(in * scaling); imresize(im, (scaled_h, scaled_w));
                                                    offset_y/x = np.random.randint(scaled - in + 1); crop back to in
  transform_list.py:100-118  ArrayToTensor          HWC -> CHW, float()/255
  transform_list.py:84-92    Normalize              (t - 0.5) / 0.5 per channel
  GDN_main.py:49-52      validation: ArrayToTensor + normalize only

Third-party arithmetic not vendored in the reference: ``scipy.misc.imresize`` (scipy < 1.3, removed since; the
reference pins no version).  Its published algorithm is restated here:
  imresize(arr, size) = fromimage(toimage(arr).resize((w, h), PIL.Image.BILINEAR))
  toimage(arr)        = bytescale(arr) for non-uint8 data: byte = uint8(clip((v - min) * (255 / (max - min)), 0, 255) + 0.5),
                        min/max over the WHOLE array, arithmetic in the array's float32 (NumPy 1.x scalar casting)
  PIL bilinear resize = Pillow's two-pass convolution resampler (src/libImaging/Resample.c): triangle filter of
                        support max(scale, 1), double-precision weights normalised to 1, converted to 22-bit fixed
                        point, horizontal pass rounded to uint8, then the vertical pass.
``resize_bilinear_u8`` is pinned against the Pillow in this image by tests/test_augment_cpu.py (bit-exact).
"""
import random

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def bytescale(data):
    """scipy.misc.bytescale(data) with the defaults imresize uses (low 0, high 255)."""
    data = np.asarray(data)
    if data.dtype == np.uint8:
        return data
    data = data.astype(np.float32, copy=False)
    cmin, cmax = data.min(), data.max()
    cscale = np.float32(cmax - cmin)
    if cscale == 0:
        cscale = np.float32(1)
    scale = np.float32(np.float64(255.0) / np.float64(cscale))
    b = (data - cmin) * scale                      # float32 throughout
    return (np.clip(b, np.float32(0), np.float32(255)) + np.float32(0.5)).astype(np.uint8)


def _coeffs(in_size, out_size):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle) filter, whole axis."""
    scale = float(in_size) / float(out_size)
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    bounds = np.zeros((out_size, 2), np.int64)
    kk = np.zeros((out_size, ksize), np.int64)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.zeros(ksize)
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - a if a < 1.0 else 0.0
        ww = w[:xmax].sum() if xmax else 0.0
        # Pillow sums sequentially in double; three terms at most here, np.sum of <= 8 doubles is sequential too
        ww = 0.0
        for x in range(xmax):
            ww += w[x]
        for x in range(xmax):
            if ww != 0.0:
                w[x] /= ww
        for x in range(ksize):
            kk[xx, x] = int(-0.5 + w[x] * (1 << PRECISION_BITS)) if w[x] < 0 else int(0.5 + w[x] * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bilinear_u8(img, out_h, out_w):
    """PIL.Image.fromarray(img).resize((out_w, out_h), BILINEAR) for uint8 [H, W] or [H, W, C] arrays."""
    img = np.asarray(img)
    assert img.dtype == np.uint8
    squeeze = img.ndim == 2
    a = img[:, :, None] if squeeze else img
    H, W, C = a.shape
    if W != out_w:
        bounds, kk = _coeffs(W, out_w)
        tmp = np.empty((H, out_w, C), np.uint8)
        src = a.astype(np.int64)
        for xx in range(out_w):
            x0, n = bounds[xx]
            acc = np.full((H, C), 1 << (PRECISION_BITS - 1), np.int64)
            for j in range(n):
                acc += src[:, x0 + j, :] * kk[xx, j]
            tmp[:, xx, :] = _clip8(acc)
        a = tmp
    if H != out_h:
        bounds, kk = _coeffs(H, out_h)
        out = np.empty((out_h, a.shape[1], C), np.uint8)
        src = a.astype(np.int64)
        for yy in range(out_h):
            y0, n = bounds[yy]
            acc = np.full((a.shape[1], C), 1 << (PRECISION_BITS - 1), np.int64)
            for j in range(n):
                acc += src[y0 + j] * kk[yy, j]
            out[yy] = _clip8(acc)
        a = out
    return a[:, :, 0] if squeeze else a


def imresize(arr, size):
    """scipy.misc.imresize(arr, (h, w)) with its defaults (interp='bilinear', mode=None)."""
    return resize_bilinear_u8(bytescale(arr), int(size[0]), int(size[1]))


def draw_params(in_h, in_w, py_rng, np_rng, train=True):
    """The reference's random draws for one sample, in its call order.  Returns (flip, scaled_h, scaled_w, off_y, off_x)."""
    if not train:
        return (0, in_h, in_w, 0, 0)
    flip = 1 if py_rng.random() < 0.5 else 0
    x_scaling, y_scaling = np_rng.uniform(1, 1.15, 2)
    scaled_h, scaled_w = int(in_h * y_scaling), int(in_w * x_scaling)
    off_y = int(np_rng.randint(scaled_h - in_h + 1))
    off_x = int(np_rng.randint(scaled_w - in_w + 1))
    return (flip, scaled_h, scaled_w, off_y, off_x)


def to_tensor_normalized(img):
    """ArrayToTensor + Normalize(0.5, 0.5): HWC (or HW) -> CHW float32 in [-1, 1]."""
    a = np.asarray(img)
    if a.ndim == 2:
        a = a[:, :, None]
    t = a.transpose(2, 0, 1).astype(np.float32) / np.float32(255)
    return (t - np.float32(0.5)) / np.float32(0.5)


def augment_sample(images, params, train=True):
    """images: list of HWC/HW arrays (uint8 or float32 as decoded); params from draw_params.  Returns CHW float32 list."""
    if not train:
        return [to_tensor_normalized(im) for im in images]
    flip, sh, sw, oy, ox = params
    in_h, in_w = images[0].shape[:2]
    out = []
    for im in images:
        if flip:
            im = np.copy(np.fliplr(im))
        im = imresize(im, (sh, sw))
        out.append(to_tensor_normalized(im[oy:oy + in_h, ox:ox + in_w]))
    return out


def make_rngs(seed):
    return random.Random(seed), np.random.RandomState(seed)
