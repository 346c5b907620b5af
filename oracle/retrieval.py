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
    lib = _chain_lib()
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    G = np.ascontiguousarray(G, dtype=np.float32)
    assert Q.shape[1] == G.shape[1] and Q.shape[1] % 8 == 0
    out = np.empty((Q.shape[0], G.shape[0]), dtype=np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    lib.sim_chain_scores.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp]
    lib.sim_chain_scores.restype = None
    lib.sim_chain_scores(Q.ctypes.data_as(fp), G.ctypes.data_as(fp), Q.shape[0], G.shape[0], Q.shape[1], out.ctypes.data_as(fp))
    return out


def _chain_lib():
    import ctypes
    import os
    so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libsimchain.so")
    if not os.path.exists(so):
        raise RuntimeError("oracle/_build/libsimchain.so missing: run python -c 'import __graft_entry__ as g; g.build()'")
    return ctypes.CDLL(so)


def similarity_topk_chain(Q, G, k, chunk=64, margin=2e-4):
    """EXACT top-k by the fmaf-chain score, (chain score desc, index asc), for shards too large to chain in full.
    Q, G: float32 tensors holding the values the GPU multiplies (16-bit galleries: Q rounded to the gallery dtype, G widened).
    Per query chunk: plain fp32 scores S = Q @ G.T, T = k-th best; every row with S >= T - margin is chained
    (oracle/c/sim_chain.c: sim_chain_pairs) and ranked. A row of the chain top-k cannot be missed: |chain - S| < 3.1e-5 for
    unit vectors (two fp32 summation orders of the same exact products), margin = 2e-4."""
    import ctypes
    lib = _chain_lib()
    fp, ip = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_longlong)
    lib.sim_chain_pairs.argtypes = [fp, fp, ip, ip, ctypes.c_longlong, ctypes.c_int, fp]
    lib.sim_chain_pairs.restype = None
    Qn = np.ascontiguousarray(Q.float().numpy()); Gn = np.ascontiguousarray(G.float().numpy())
    Bq, Ng = Qn.shape[0], Gn.shape[0]
    kk = min(k, Ng)
    rs = torch.empty((Bq, kk)); ri = torch.empty((Bq, kk), dtype=torch.int64)
    Gt = torch.from_numpy(Gn)
    for c0 in range(0, Bq, chunk):
        S = torch.from_numpy(Qn[c0:c0 + chunk]) @ Gt.T
        T = torch.topk(S, kk, dim=1).values[:, -1:]
        qi, gi = (S >= T - margin).nonzero(as_tuple=True)
        qi = np.ascontiguousarray((qi + c0).numpy().astype(np.int64)); gi = np.ascontiguousarray(gi.numpy().astype(np.int64))
        out = np.empty(qi.shape[0], dtype=np.float32)
        lib.sim_chain_pairs(Qn.ctypes.data_as(fp), Gn.ctypes.data_as(fp), qi.ctypes.data_as(ip), gi.ctypes.data_as(ip), qi.shape[0], Qn.shape[1],
                            out.ctypes.data_as(fp))
        order = np.lexsort((gi, -out.astype(np.float64), qi))          # by query, then chain score desc, then index asc
        qi, gi, out = qi[order], gi[order], out[order]
        starts = np.searchsorted(qi, np.arange(c0, min(c0 + chunk, Bq)))
        for j, st in enumerate(starts):
            rs[c0 + j] = torch.from_numpy(out[st:st + kk]); ri[c0 + j] = torch.from_numpy(gi[st:st + kk])
    return rs, ri


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
