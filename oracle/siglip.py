"""Oracle (test infrastructure): SigLIP vision trunk (patch tokens) and text tower.

PARITY UNPINNED BY THE REFERENCE: the arithmetic lives in third-party packages that are neither vendored
in /root/reference nor installed (open_clip_torch==2.31.0, timm==1.0.15; requirements.txt:55,77).
Restated from the published architecture:
  * vision = timm.VisionTransformer (vit_*_siglip_384): conv patch-embed with bias, learned pos_embed,
    no class token, pre-LN blocks (eps 1e-6, fused qkv with bias, softmax(q k^T / sqrt(hd)) v, proj,
    MLP fc1-GELU-fc2), final norm. The reference reads trunk.{patch_embed,pos_embed,blocks,norm} directly
    (lib/support_model/siglip_openclip.py:30-35) and never uses the MAP head output on the live path.
  * text = open_clip.TextTransformer: token_embedding + positional_embedding, NO causal mask, pre-LN
    nn.MultiheadAttention blocks, ln_final, pool = last token, text_projection = Linear with bias;
    then F.normalize (lib/support_model/siglip_openclip.py:53-56).
Cross-checked against transformers.models.siglip (random init) in tests/test_oracle_siglip_hf.py.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .sam import _linear, _ln, gelu_erf, patch_tokens


def gelu_tanh(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def _act(kind):
    return gelu_erf if kind == "erf" else gelu_tanh


def _mha(q, k, v, heads):
    """softmax(q k^T / sqrt(hd)) v over [N, T, D] with D split into heads."""
    N, T, D = q.shape
    hd = D // heads
    sp = lambda t: t.reshape(N, T, heads, hd).transpose(1, 2)
    a = torch.softmax((sp(q) * hd ** -0.5) @ sp(k).transpose(2, 3), dim=-1)
    return (a @ sp(v)).transpose(1, 2).reshape(N, T, D)


def vision_tokens(sd, x, cfg, p="support_branch.siglip.model.visual.trunk."):
    """ref: lib/support_model/siglip_openclip.py:30-35 (+ timm VisionTransformer blocks).
    x [N,3,384,384] -> last hidden states [N, P, D] (after trunk.norm)."""
    N = x.shape[0]
    t = patch_tokens(x, sd[p + "patch_embed.proj.weight"], sd[p + "patch_embed.proj.bias"])
    t = t.reshape(N, -1, t.shape[-1]) + sd[p + "pos_embed"]
    act = _act(cfg.get("v_gelu", cfg.get("gelu", "erf")))        # per-tower GELU flavour (legacy key `gelu` = both)
    for i in range(cfg["depth"]):
        b = f"{p}blocks.{i}."
        h = _ln(sd, b + "norm1.", t, 1e-6)
        q, k, v = _linear(sd, b + "attn.qkv.", h).chunk(3, dim=-1)
        t = t + _linear(sd, b + "attn.proj.", _mha(q, k, v, cfg["heads"]))
        h = _ln(sd, b + "norm2.", t, 1e-6)
        t = t + _linear(sd, b + "mlp.fc2.", act(_linear(sd, b + "mlp.fc1.", h)))
    return _ln(sd, p + "norm.", t, 1e-6)


def tokens_to_nchw(t):
    """ref: lib/support_model/siglip_openclip.py:38-42. [N,P,D] -> [N,D,sqrt(P),sqrt(P)]"""
    N, P, D = t.shape
    g = int(math.isqrt(P))
    return t.permute(0, 2, 1).reshape(N, D, g, g)


def text_features(sd, tokens, cfg, p="support_branch.siglip.model.text.", normalize=True):
    """ref: lib/support_model/siglip_openclip.py:46-59 (+ open_clip TextTransformer, pool 'last').
    tokens int64 [N,64] -> [N,D]"""
    x = sd[p + "token_embedding.weight"][tokens] + sd[p + "positional_embedding"][: tokens.shape[1]]
    act = _act(cfg.get("t_gelu", cfg.get("gelu", "erf")))
    D = x.shape[-1]
    for i in range(cfg["t_depth"]):
        b = f"{p}transformer.resblocks.{i}."
        h = _ln(sd, b + "ln_1.", x, 1e-6)
        qkv = h @ sd[b + "attn.in_proj_weight"].T + sd[b + "attn.in_proj_bias"]
        q, k, v = qkv.split(D, dim=-1)
        x = x + _linear(sd, b + "attn.out_proj.", _mha(q, k, v, cfg["t_heads"]))
        h = _ln(sd, b + "ln_2.", x, 1e-6)
        x = x + _linear(sd, b + "mlp.c_proj.", act(_linear(sd, b + "mlp.c_fc.", h)))
    x = _ln(sd, p + "ln_final.", x, 1e-6)
    feat = _linear(sd, p + "text_projection.", x[:, -1])
    return F.normalize(feat, dim=-1) if normalize else feat
