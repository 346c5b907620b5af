#!/usr/bin/env python3
"""Golden vectors for the input pre-processing (Pillow's BILINEAR resize as torchvision.transforms.Resize calls it on a PIL
image, utils/dataloader.py:266-293), produced with Pillow itself (12.2.0 here): random uint8 images -> resized uint8.
    python tools/make_golden_preprocess.py   -> tests/golden/preprocess_resize.npz"""
import os
import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rng = np.random.default_rng(20260703)
cases = {}
for name, (h, w, c, oh, ow) in {"rgb_down": (97, 131, 3, 24, 24), "rgb_up": (19, 23, 3, 48, 48), "rgb_mixed": (40, 200, 3, 64, 64),
                                "mask_down": (150, 111, 1, 32, 32), "mask_up": (9, 7, 1, 16, 16), "rgb_big_down": (211, 307, 3, 56, 56),
                                "rgb_same_w": (70, 48, 3, 48, 48)}.items():
    a = rng.integers(0, 256, size=(h, w, c), dtype=np.uint8)
    if c == 1:
        a = (a > 128).astype(np.uint8) * 255 if "mask" in name else a
        im = Image.fromarray(a[:, :, 0], mode="L")
    else:
        im = Image.fromarray(a, mode="RGB")
    out = np.asarray(im.resize((ow, oh), Image.BILINEAR))
    cases[name + "_in"] = a if c == 3 else a[:, :, 0]
    cases[name + "_out"] = out
import PIL
cases["pillow_version"] = np.array(PIL.__version__)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "preprocess_resize.npz"), **cases)
print({k: v.shape for k, v in cases.items() if hasattr(v, "shape") and v.ndim})
