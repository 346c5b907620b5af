"""Oracle (test infrastructure, CPU only): the reference's image pre-processing, restated.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this package; the product path
(cor_amd.preprocess -> libcor_amd.so) never does.

ref: utils/dataloader.py:266-293 (Resize((S,S)) -> ToTensor() -> Normalize(mean, std) for the query (S=1024) and support
(S=384) images, Resize -> ToTensor for the masks), :349-350. torchvision's Resize on a PIL image is `img.resize((S,S),
Image.BILINEAR)`; that algorithm lives in a third-party dependency that is not part of /root/reference:
Pillow (12.2.0 in this image), src/libImaging/Resample.c, restated here from its published source:
  * precompute_coeffs(): per output index xx: center = (xx+0.5)*scale, scale = in/out, filterscale = max(scale, 1),
    support = 1.0*filterscale (bilinear), xmin = max(0, int(center-support+0.5)), xmax = min(in, int(center+support+0.5)),
    weights w = max(0, 1-|(x+xmin-center+0.5)/filterscale|) normalised to sum 1 (double precision);
  * normalize_coeffs_8bpc(): k = int(0.5 + w * 2^22) (PRECISION_BITS = 32-8-2);
  * ImagingResampleHorizontal/Vertical_8bpc(): acc = 2^21 + sum(pixel * k); out = clip(acc >> 22, 0, 255); the
    horizontal pass runs first and its uint8 result feeds the vertical pass.
ToTensor = uint8 -> float32 / 255; Normalize = (x - mean) / std in float32 (torchvision functional_tensor.normalize).
Pinned by tests/golden/preprocess_*.npz, produced with Pillow itself by tools/make_golden_preprocess.py.
"""
from __future__ import annotations

import numpy as np

PRECISION_BITS = 32 - 8 - 2
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def precompute_coeffs(in_size: int, out_size: int):
    """Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR filter over the whole axis.
    -> (bounds int32[out,2] = (xmin, count), kk int32[out,ksize], ksize)"""
    scale = float(in_size) / out_size            # C: (double)(in1 - in0) / outSize with float box ends
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)       # C cast: truncation toward zero (arguments are > -1 here)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        x = np.arange(xmax, dtype=np.float64)
        w = np.abs((x + xmin - center + 0.5) * ss)
        w = np.where(w < 1.0, 1.0 - w, 0.0)
        ww = w.sum()
        if ww != 0.0:
            w = w / ww
        k = np.where(w < 0, -0.5 + w * (1 << PRECISION_BITS), 0.5 + w * (1 << PRECISION_BITS)).astype(np.int64)   # (int) truncates
        bounds[xx] = (xmin, xmax)
        kk[xx, :xmax] = k
    return bounds, kk, ksize


def _clip8(acc):
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bilinear_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """img uint8 [H,W] or [H,W,C] -> uint8 [out_h,out_w(,C)], bit-for-bit Pillow `resize((out_w,out_h), BILINEAR)`."""
    squeeze = img.ndim == 2
    a = img[:, :, None] if squeeze else img
    H, W, C = a.shape
    if W != out_w:                               # horizontal pass (Resample.c: need_horizontal)
        b, kk, ks = precompute_coeffs(W, out_w)
        tmp = np.empty((H, out_w, C), np.uint8)
        a64 = a.astype(np.int64)
        for xx in range(out_w):
            x0, n = b[xx]
            acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(a64[:, x0:x0 + n, :], kk[xx, :n].astype(np.int64), axes=([1], [0]))
            tmp[:, xx, :] = _clip8(acc)
        a = tmp
    if H != out_h:                               # vertical pass
        b, kk, ks = precompute_coeffs(H, out_h)
        out = np.empty((out_h, a.shape[1], C), np.uint8)
        a64 = a.astype(np.int64)
        for yy in range(out_h):
            y0, n = b[yy]
            acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kk[yy, :n].astype(np.int64), a64[y0:y0 + n], axes=([0], [0]))
            out[yy] = _clip8(acc)
        a = out
    return a[:, :, 0] if squeeze else a


def to_tensor_normalize(img_u8: np.ndarray, mean=None, std=None) -> np.ndarray:
    """ToTensor (+ Normalize): uint8 [H,W(,C)] -> float32 [C,H,W]; x/255 then (x - mean)/std, all in float32."""
    a = img_u8[:, :, None] if img_u8.ndim == 2 else img_u8
    x = a.transpose(2, 0, 1).astype(np.float32) / np.float32(255)
    if mean is not None:
        m = np.asarray(mean, np.float32)[:, None, None]
        s = np.asarray(std, np.float32)[:, None, None]
        x = (x - m) / s
    return x


def preprocess_image(img_u8: np.ndarray, size: int, normalize: bool = True) -> np.ndarray:
    """ref: utils/dataloader.py:266-293: Resize((size,size)) -> ToTensor -> Normalize (images) / Resize -> ToTensor (masks)."""
    r = resize_bilinear_u8(img_u8, size, size)
    return to_tensor_normalize(r, IMAGENET_MEAN if normalize else None, IMAGENET_STD if normalize else None)
