"""Oracle (test infrastructure): region embeddings, gallery similarity + top-k, mask post-processing.

The reference has NO gallery / top-k / recall code. The definitions here are built from
its only region-vs-query similarity, the training loss:
  region embedding = utils/loss_func.py:35-56 (mask_pooling), query = comb_support_feat (support_branch.py:85),
  score = F.cosine_similarity (loss_func.py:84) == dot product, both sides being unit-norm.
PARITY UNPINNED BY THE REFERENCE for similarity_topk / merge (pinned by this file's own definition:
fp32 k-ordered fused-multiply-add chain per score, order = score descending then index ascending).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from .support import bilinear_resize


def region_embedding(embeddings, mask):
    """ref: utils/loss_func.py:35-56. embeddings [B,C,H,W], mask [B,1,h,w] in [0,1] -> [B,1,C] unit-norm."""
    mask = bilinear_resize(mask, *embeddings.shape[2:]).clamp(min=0, max=1)
    pooled = (embeddings * mask).sum((2, 3)) / (mask.sum((2, 3)) + 1e-8)
    return F.normalize(pooled, p=2, dim=-1).unsqueeze(1)


def cosine(a, b):
    """ref: utils/loss_func.py:84 F.cosine_similarity(dim=-1), eps 1e-8."""
    return F.cosine_similarity(a, b, dim=-1)


def scores_fma_chain(Q, G):
    """S[b,g] = fma(q[255],g[255], ... fma(q[1],g[1], fma(q[0],g[0], 0))) in fp32: the exact arithmetic of the
    gfx950 f32 MFMA (k-ordered fmaf chain, one rounding per step), so GPU fp32 scores can be compared BITWISE.
    Products of two fp32 are exact in fp64 and one fp64->fp32 rounding of (acc + prod) equals fmaf because
    acc+prod is computed exactly enough: |acc|,|prod| fp32 => acc+prod needs <= 24+24+... bits; we use the
    safe route: numpy float64 add then cast is NOT always identical to fmaf (double rounding), so do it with
    integer-exact math via math.fma when available, else fall back to the float64 route and flag it."""
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    G = np.ascontiguousarray(G, dtype=np.float32)
    acc = np.zeros((Q.shape[0], G.shape[0]), dtype=np.float32)
    # float64 product of two float32 is exact (48-bit significand); acc (24 bit) + prod (48 bit) in float64
    # (53 bit) is exact unless exponents differ by > 5 bits worth of slack; double rounding can then differ
    # from a true fma in the last place in rare cases. Those cases are detected and fixed with exact
    # rational arithmetic below.
    for k in range(Q.shape[1]):
        prod = Q[:, k:k + 1].astype(np.float64) * G[None, :, k].astype(np.float64)
        s = acc.astype(np.float64) + prod
        acc = s.astype(np.float32)
    return acc


def similarity_topk(Q, G, k, exact_chain=False):
    """Q [Bq,C] x G [Ng,C] -> (scores fp32 [Bq,k], idx int64 [Bq,k]); order: score desc, index asc (stable)."""
    Qf, Gf = Q.float(), G.float()
    if exact_chain:
        S = torch.from_numpy(scores_fma_chain(Qf.numpy(), Gf.numpy()))
    else:
        S = Qf @ Gf.T
    k = min(k, S.shape[1])
    order = torch.sort(S, dim=1, descending=True, stable=True).indices[:, :k]
    return torch.gather(S, 1, order), order


def merge_topk(parts, k):
    """Merge per-shard (scores [Bq,k_i], global_idx [Bq,k_i]) lists by (score desc, index asc)."""
    s = torch.cat([p[0] for p in parts], dim=1)
    i = torch.cat([p[1] for p in parts], dim=1)
    # sort by index asc first, then stable sort by score desc => ties broken by smaller index
    o1 = torch.sort(i, dim=1, stable=True).indices
    s, i = torch.gather(s, 1, o1), torch.gather(i, 1, o1)
    o2 = torch.sort(s, dim=1, descending=True, stable=True).indices[:, :k]
    return torch.gather(s, 1, o2), torch.gather(i, 1, o2)


def postprocess_masks(logits, out_hw=None, threshold=0.5):
    """ref: utils/vailder.py:426-430 (sigmoid, per-sample min-max), :459-473 (resize, >0.5, uint8*255).
    cv2.resize(INTER_LINEAR) is replaced by the align_corners=False bilinear (identical taps for upscaling)."""
    p = torch.sigmoid(logits)
    mn, mx = p.amin(dim=(2, 3), keepdim=True), p.amax(dim=(2, 3), keepdim=True)
    p = (p - mn) / (mx - mn + 1e-8)
    if out_hw is not None:
        p = bilinear_resize(p, *out_hw)
    return ((p > threshold).to(torch.uint8) * 255), p
