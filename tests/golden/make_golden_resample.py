"""Generates tests/golden/resample_pil.npz with PILLOW ITSELF (the dependency the reference's transforms
run on): CLIP resize+crop outputs and 3x3 'simple' chunks of small seeded images.
Run from the repo root:  python tests/golden/make_golden_resample.py"""
import os

import numpy as np
from PIL import Image

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "resample_pil.npz")


def clip_resize_crop(img, n_px=224):
    im = Image.fromarray(img)
    w, h = im.size
    short, long = (w, h) if w <= h else (h, w)
    if short != n_px:
        new_short, new_long = n_px, int(n_px * long / short)
        nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
        im = im.resize((nw, nh), Image.BICUBIC)
    w, h = im.size
    top, left = int(round((h - n_px) / 2.0)), int(round((w - n_px) / 2.0))
    return np.asarray(im.crop((left, top, left + n_px, top + n_px)).convert("RGB"))


def main():
    rng = np.random.default_rng(20240924)
    sizes = [(97, 131), (260, 180), (64, 300)]
    d = {"n": np.int64(len(sizes))}
    for i, (h, w) in enumerate(sizes):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        d[f"img{i}"] = img
        d[f"clip{i}"] = clip_resize_crop(img)
    im = Image.fromarray(d["img0"])
    resized = im.resize((240, 240))  # PatchifySimple.infer: default resample (BICUBIC)
    boxes = [(0, 0, 240, 240)] + [(j, i, j + 80, i + 80) for i in range(0, 240, 80) for j in range(0, 240, 80)]
    for k, bb in enumerate(boxes):
        d[f"chunk{k}"] = np.asarray(resized.crop(bb))
    fx, fy = im.size[0] / 240, im.size[1] / 240
    d["chunk_boxes"] = np.asarray([[b[0] * fx, b[1] * fy, b[2] * fx, b[3] * fy] for b in boxes])
    np.savez_compressed(OUT, **d)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
