"""Demo inference on image files with the HIP path (counterpart of the reference's depth_extract.py:60-150).

    python -m gdn_amd.depth_extract --model_dir X.pkl --img_dir ./imgs --out_dir ./depth [--batch 8]

Loads a reference-format checkpoint (``module.``-prefixed keys of the legacy ``AutoEncoder``), resizes
each image to 128x416 (PIL bilinear; the reference used the since-removed scipy.misc.imresize), maps
to [-1,1] exactly like ArrayToTensor + Normalize (transform_list.py:89-113), runs the forward on the
GPU and writes the depth map resized back to the source resolution as 16-bit PNG.  Unlike the
reference's timing loop (sync before but not after the forward, :117-125) the reported time brackets
the forward with synchronisation on both sides.
"""
import argparse
import pathlib
import time

import numpy as np
import torch

from .AE_model_unet import AutoEncoder
from .trainer import load_checkpoint

EXTS = (".png", ".jpg", ".jpeg", ".bmp")


def load_image(path, H, W):
    from PIL import Image
    im = Image.open(path).convert("RGB")
    size = im.size
    arr = np.asarray(im.resize((W, H), Image.BILINEAR), dtype=np.float32) / 255.0
    t = torch.from_numpy(arr).permute(2, 0, 1)
    return (t - 0.5) / 0.5, size


def main(argv=None):
    ap = argparse.ArgumentParser(description="GDN depth extraction on MI355X")
    ap.add_argument("--model_dir", type=str, default="")
    ap.add_argument("--img_dir", type=str, required=True)
    ap.add_argument("--out_dir", type=str, default="./depth_out")
    ap.add_argument("--height", type=int, default=128)
    ap.add_argument("--width", type=int, default=416)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
    a = ap.parse_args(argv)
    from PIL import Image
    dev = torch.device("cuda", 0)
    model = AutoEncoder(height=a.height, width=a.width)
    if a.model_dir:
        load_checkpoint(model, a.model_dir)
    else:
        print("=> no --model_dir: running with randomly initialised weights")
    model = model.to(dev).eval().compute_dtype(a.dtype)
    files = sorted(p for p in pathlib.Path(a.img_dir).iterdir() if p.suffix.lower() in EXTS)
    out = pathlib.Path(a.out_dir)
    out.mkdir(parents=True, exist_ok=True)
    total, n = 0.0, 0
    for i in range(0, len(files), a.batch):
        chunk = files[i:i + a.batch]
        loaded = [load_image(p, a.height, a.width) for p in chunk]
        x = torch.stack([t for t, _ in loaded]).to(dev)
        torch.cuda.synchronize()
        t0 = time.time()
        d = model(x, istrain=False)
        torch.cuda.synchronize()
        total += time.time() - t0
        n += len(chunk)
        d = ((d.float().cpu() + 1) / 2).clamp(0, 1)
        for (_, size), p, dm in zip(loaded, chunk, d):
            im = Image.fromarray((dm[0].numpy() * 65535).astype(np.uint16))
            im.resize(size, Image.BILINEAR).save(out / (p.stem + "_depth.png"))
    if n:
        print("Avg time: %.4f s/image over %d images" % (total / n, n))


if __name__ == "__main__":
    main()
