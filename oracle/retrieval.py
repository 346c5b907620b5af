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
    """fp32 scores by the exact k-ordered fmaf chain of the GPU's f32 MFMA (oracle/c/sim_chain.c, built by
    __graft_entry__.build() into oracle/_build/libsimchain.so): bitwise comparable with cor_similarity_topk on an
    fp32 gallery. Q [Bq,C], G [Ng,C] float32 numpy arrays, C % 8 == 0."""
    import ctypes
    import os
    so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libsimchain.so")
    if not os.path.exists(so):
        raise RuntimeError("oracle/_build/libsimchain.so missing: run python -c 'import __graft_entry__ as g; g.build()'")
    lib = ctypes.CDLL(so)
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    G = np.ascontiguousarray(G, dtype=np.float32)
    assert Q.shape[1] == G.shape[1] and Q.shape[1] % 8 == 0
    out = np.empty((Q.shape[0], G.shape[0]), dtype=np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    lib.sim_chain_scores.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp]
    lib.sim_chain_scores.restype = None
    lib.sim_chain_scores(Q.ctypes.data_as(fp), G.ctypes.data_as(fp), Q.shape[0], G.shape[0], Q.shape[1], out.ctypes.data_as(fp))
    return out


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


def mask_metrics(pred, gt, smooth=1e-5):
    """ref: utils/trainer_v3_g.py:381-443 -> [B,5] = dice, mae, iou, mdice, miou."""
    p, g = pred.reshape(pred.shape[0], -1).float(), gt.reshape(gt.shape[0], -1).float()

    def dice(a, b):
        return (2.0 * (a * b).sum(1) + smooth) / (a.sum(1) + b.sum(1) + smooth)

    def iou(a, b):
        inter = (a * b).sum(1)
        return (inter + smooth) / (a.sum(1) + b.sum(1) - inter + smooth)

    return torch.stack([dice(p, g), (p - g).abs().mean(1), iou(p, g), 0.5 * (dice(p, g) + dice(1 - p, 1 - g)),
                        0.5 * (iou(p, g) + iou(1 - p, 1 - g))], dim=1)
