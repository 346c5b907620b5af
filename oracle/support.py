"""Oracle (test infrastructure): support branch = mask pooling (RRE), gated fusion (AVTI), projection head.

Functional fp32 CPU restatement; each function cites the reference lines it follows.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .sam import _linear, _ln, gelu_erf
from . import siglip as _siglip


def bilinear_resize(x, oh, ow):
    """F.interpolate(mode='bilinear', align_corners=False, antialias=False) written out as a 4-tap gather:
    src = max((dst+0.5)*in/out - 0.5, 0); i0 = floor(src); i1 = min(i0+1, in-1); w1 = src - i0.
    (used by ref: mask_adapter.py:20,58,158,62-67 ; utils/loss_func.py:47). x [B,C,H,W]"""
    B, C, H, W = x.shape
    if (H, W) == (oh, ow):
        return x

    def taps(n_in, n_out):
        s = (torch.arange(n_out, dtype=x.dtype) + 0.5) * (n_in / n_out) - 0.5      # x's dtype: an fp64 evaluation stays fp64
        s = s.clamp(min=0)
        i0 = s.floor().long().clamp(max=n_in - 1)
        i1 = (i0 + 1).clamp(max=n_in - 1)
        return i0, i1, s - i0.to(x.dtype)

    y0, y1, wy = taps(H, oh)
    x0, x1, wx = taps(W, ow)
    top = x[:, :, y0][:, :, :, x0] * (1 - wx) + x[:, :, y0][:, :, :, x1] * wx
    bot = x[:, :, y1][:, :, :, x0] * (1 - wx) + x[:, :, y1][:, :, :, x1] * wx
    return top * (1 - wy)[:, None] + bot * wy[:, None]


def _ln_cf(sd, p, x, eps=1e-6):
    """channels_first LayerNorm over dim 1 of NCHW. ref: lib/support_model/mask_adapter.py:246-251."""
    return _ln(sd, p, x.permute(0, 2, 3, 1), eps).permute(0, 3, 1, 2)


def _conv1x1(sd, p, x):
    w = sd[p + "weight"]
    return torch.einsum("bchw,oc->bohw", x, w[:, :, 0, 0]) + sd[p + "bias"][None, :, None, None]


def masked_pooling(feat, mask):
    """ref: lib/support_model/mask_adapter.py:13-25. feat [B,C,H,W], mask [B,1,h,w] -> [B,C]"""
    mask = bilinear_resize(mask, *feat.shape[2:])
    return ((feat * mask).sum((2, 3)) / (mask.sum((2, 3)) + 1e-8)).squeeze(1)


def convnext_block(sd, p, x):
    """ref: lib/support_model/mask_adapter.py:210-223. x [B,C,H,W]"""
    C = x.shape[1]
    y = F.conv2d(x, sd[p + "dwconv.weight"], sd[p + "dwconv.bias"], padding=3, groups=C).permute(0, 2, 3, 1)
    y = _ln(sd, p + "norm.", y, 1e-6)
    y = _linear(sd, p + "pwconv2.", gelu_erf(_linear(sd, p + "pwconv1.", y))) * sd[p + "gamma"]
    return x + y.permute(0, 3, 1, 2)


def mask_adapter_maps(sd, p, dense, mask):
    """GenerateMaskAdapterMap.forward with one mask per sample.
    ref: lib/support_model/mask_adapter.py:144-179 (mask_downscaling :128-142). dense [B,512,H,W], mask [B,1,H,W]"""
    H, W = dense.shape[-2:]
    m = bilinear_resize(mask.to(dense.dtype), 4 * H, 4 * W)
    md = p + "mask_downscaling."
    m = F.conv2d(m, sd[md + "0.weight"], sd[md + "0.bias"], stride=2, padding=1)
    m = gelu_erf(_ln_cf(sd, md + "1.", m))
    m = F.conv2d(m, sd[md + "3.weight"], sd[md + "3.bias"], stride=2, padding=1)
    m = gelu_erf(_ln_cf(sd, md + "4.", m))
    m = _conv1x1(sd, md + "6.", m)
    y = _conv1x1(sd, p + "fuse.", dense + m)
    for i in (1, 2, 3):
        y = convnext_block(sd, f"{p}cnext{i}.", y)
    y = _ln(sd, p + "norm.", y.permute(0, 2, 3, 1), 1e-6).permute(0, 3, 1, 2)
    return _conv1x1(sd, p + "final.", y)


def mask_adapter_pooling(sd, feat, mask, p="support_branch.mask_pooling.", num_maps=8, return_maps=False):
    """ref: lib/support_model/mask_adapter.py:52-80 (ChannelReduction :83-94).
    feat [B,D,H,W], mask [B,1,h,w] -> [B,1,D]"""
    B, D, H, W = feat.shape
    mask = bilinear_resize(mask, H, W)
    c = p + "channel_clip_to_maskadapter."
    dense = gelu_erf(_ln_cf(sd, c + "norm.", _conv1x1(sd, c + "conv.", feat)))
    maps = mask_adapter_maps(sd, p + "get_mask_map.", dense, mask)       # [B,8,H,W]
    maps = bilinear_resize(maps, H, W)                                   # same size: identity
    a = torch.softmax(F.logsigmoid(maps).reshape(B, num_maps, H * W), dim=-1)
    pooled = a @ feat.reshape(B, D, H * W).transpose(1, 2)               # [B,8,D]
    out = pooled.reshape(B, 1, num_maps, D).mean(dim=2)
    return (out, maps) if return_maps else out


def cir_fuse(sd, img, txt, p="support_branch.cir_fuse."):
    """ref: lib/support_model/cir_feature_fuse.py:44-64 (Dropout = identity in eval). [N,D],[N,D] -> [N,D]"""
    def gate(q, x):
        return torch.sigmoid(_linear(sd, q + "3.", torch.relu(_linear(sd, q + "0.", x))))
    raw = torch.cat([img, txt], -1)
    img2 = gate(p + "atten_Image.", raw) * img
    txt2 = gate(p + "atten_Text.", raw) * txt
    dyn = gate(p + "dynamic_scalar.", torch.cat([img2, txt2], -1))       # [N,1]
    return F.normalize(dyn * img2 + (1 - dyn) * txt2)


def support_head(sd, tokens_nchw, text_feat, mask, mask_pooling, p="support_branch."):
    """Everything in SupportBranch.forward after the SigLIP towers.
    ref: lib/support_branch.py:58-62,65-66,85-86 (dim_proj :47-54; Dropout(0.8) = identity in eval)."""
    x = _ln_cf(sd, p + "ln_channel_first.", tokens_nchw)
    if mask_pooling == "MaskAdapterPooling":
        pooled = mask_adapter_pooling(sd, x, mask, p + "mask_pooling.")  # [N,1,D]
    elif mask_pooling == "MaskedPooling":
        pooled = masked_pooling(x, mask)                                 # [N,D]
    else:
        raise ValueError(f"Invalid mask pooling method: {mask_pooling}")
    pooled = _ln(sd, p + "ln_channel_last.", pooled, 1e-6)
    if pooled.dim() == 3:
        pooled = pooled.squeeze(1)
    fused = cir_fuse(sd, pooled, text_feat, p + "cir_fuse.")
    h = gelu_erf(_linear(sd, p + "dim_proj.0.", fused))
    h = gelu_erf(_linear(sd, p + "dim_proj.3.", h))
    return F.normalize(h, p=2, dim=-1).unsqueeze(1)                      # [N,1,256]


def support_branch(sd, s_img, text, mask, gcfg, mask_pooling, p="support_branch."):
    """ref: lib/support_branch.py:56-87."""
    tok = _siglip.vision_tokens(sd, s_img, gcfg, p + "siglip.model.visual.trunk.")
    txt = _siglip.text_features(sd, text, gcfg, p + "siglip.model.text.")
    return support_head(sd, _siglip.tokens_to_nchw(tok), txt, mask, mask_pooling, p)
