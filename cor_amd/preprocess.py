"""Input pre-processing on the GPU: the device side of the reference's data-loader transforms.

ref: utils/dataloader.py:266-293 — `Resize((S,S)) -> ToTensor() -> Normalize(IMAGENET mean/std)` for the query (S = 1024) and
support (S = 384) images, `Resize -> ToTensor` for the masks; :349-350 applies them. torchvision's Resize on a PIL image is
Pillow's antialiased BILINEAR `Image.resize` on uint8 data (Pillow src/libImaging/Resample.c): an integer algorithm, so the
HIP kernels (csrc/preproc.hip) are bit-exact with it; the fixed-point coefficient tables are computed here on the host in
double precision exactly as Pillow does (precompute_coeffs + normalize_coeffs_8bpc) and cached per (in, out) size.

    x = preprocess.QueryImageTransform()(img_u8)        # uint8 [H,W,3] on the GPU -> float32 [3,1024,1024]
    m = preprocess.MaskTransform(384)(mask_u8)          # uint8 [H,W] -> float32 [1,384,384]

Decoding (PNG/JPEG -> uint8) and the SigLIP tokenizer stay on the host (no vocabulary file in this image)."""
from __future__ import annotations

import functools

import numpy as np
import torch

from . import _native as nat

PRECISION_BITS = 32 - 8 - 2
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


@functools.lru_cache(maxsize=256)
def _tables_host(in_size: int, out_size: int):
    """Pillow Resample.c: precompute_coeffs() for the BILINEAR filter (support 1.0) over the whole axis, then
    normalize_coeffs_8bpc(). -> (bounds int32[out,2], kk int32[out,ksize], ksize)"""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    xx = np.arange(out_size, dtype=np.float64)
    center = (xx + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)            # C (int) cast: values > -1, truncation
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    x = np.arange(ksize, dtype=np.float64)[None, :]
    w = np.abs((x + xmin[:, None] - center[:, None] + 0.5) * (1.0 / filterscale))
    w = np.where(w < 1.0, 1.0 - w, 0.0)
    w = np.where(x < xmax[:, None], w, 0.0)
    ww = w.sum(axis=1, keepdims=True)
    w = np.where(ww != 0.0, w / np.where(ww != 0.0, ww, 1.0), w)
    kk = (0.5 + w * (1 << PRECISION_BITS)).astype(np.int64)                    # weights are >= 0 for this filter
    kk = np.where(x < xmax[:, None], kk, 0).astype(np.int32)
    bounds = np.stack([xmin, xmax], axis=1).astype(np.int32)
    return bounds, kk, ksize


_dev_tables: dict = {}


def resample_tables(in_size: int, out_size: int, device):
    key = (in_size, out_size, str(device))
    if key not in _dev_tables:
        b, k, ks = _tables_host(in_size, out_size)
        _dev_tables[key] = (torch.from_numpy(b).to(device), torch.from_numpy(k).to(device), ks)
    return _dev_tables[key]


def _s():
    return torch.cuda.current_stream().cuda_stream


def resize_to_tensor(img: torch.Tensor, out_h: int, out_w: int, mean=None, std=None, return_u8: bool = False):
    """img uint8 [H,W] or [H,W,C] (C in {1,3}) on the GPU -> float32 [C,out_h,out_w] (ToTensor, + Normalize when mean/std are
    given); with return_u8 also the resized uint8 image [out_h,out_w,C] (what PIL's resize returns)."""
    if not img.is_cuda:
        raise RuntimeError("cor_amd.preprocess runs on the GPU (no CPU path)")
    if img.dtype != torch.uint8 or img.dim() not in (2, 3):
        raise ValueError("expected a uint8 [H,W] or [H,W,C] image")
    a = img.unsqueeze(-1) if img.dim() == 2 else img
    a = a.contiguous()
    H, W, C = a.shape
    if C not in (1, 3):
        raise ValueError("C must be 1 or 3")
    with torch.cuda.device(a.device):                # raw-pointer launches go to the CURRENT device: make it the image's
        return _resize_to_tensor(a, H, W, C, out_h, out_w, mean, std, return_u8)


def _resize_to_tensor(a, H, W, C, out_h, out_w, mean, std, return_u8):
    lib = nat.load()
    dev = a.device
    if W != out_w:                                   # Pillow: the horizontal pass runs first when the width changes
        b, k, ks = resample_tables(W, out_w, dev)
        tmp = torch.empty((H, out_w, C), dtype=torch.uint8, device=dev)
        nat.check(lib.cor_resample_rows_u8(a.data_ptr(), tmp.data_ptr(), b.data_ptr(), k.data_ptr(), ks, H, W, C, out_w, _s()),
                  "cor_resample_rows_u8")
        a = tmp
    b, k, ks = resample_tables(H, out_h, dev)        # H == out_h gives the identity tables (one tap of 2^22)
    out = torch.empty((C, out_h, out_w), dtype=torch.float32, device=dev)
    out_u8 = torch.empty((out_h, out_w, C), dtype=torch.uint8, device=dev) if return_u8 else None
    m = torch.tensor(mean, dtype=torch.float32, device=dev) if mean is not None else None
    s = torch.tensor(std, dtype=torch.float32, device=dev) if std is not None else None
    nat.check(lib.cor_resample_cols_u8(a.data_ptr(), out.data_ptr(), out_u8.data_ptr() if return_u8 else None, b.data_ptr(), k.data_ptr(),
                                       ks, H, out_w, C, out_h, m.data_ptr() if m is not None else None,
                                       s.data_ptr() if s is not None else None, _s()), "cor_resample_cols_u8")
    return (out, out_u8) if return_u8 else out


class ImageTransform:
    """Resize((size,size)) -> ToTensor() -> Normalize(IMAGENET). ref: utils/dataloader.py:266-272, 280-286."""

    def __init__(self, size: int):
        self.size = int(size)

    def __call__(self, img_u8: torch.Tensor) -> torch.Tensor:
        return resize_to_tensor(img_u8, self.size, self.size, IMAGENET_MEAN, IMAGENET_STD)


class QueryImageTransform(ImageTransform):
    def __init__(self):
        super().__init__(1024)                       # "use sam pretrain can't change" (dataloader.py:262)


class MaskTransform:
    """Resize((size,size)) -> ToTensor(). ref: utils/dataloader.py:274-279, 288-293."""

    def __init__(self, size: int):
        self.size = int(size)

    def __call__(self, mask_u8: torch.Tensor) -> torch.Tensor:
        return resize_to_tensor(mask_u8, self.size, self.size)
