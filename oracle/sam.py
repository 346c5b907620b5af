"""Oracle (test infrastructure): SAM ViTDet image encoder, dense PE, two-way transformer, mask decoder.

Functional fp32 CPU restatement over a flat state_dict ``sd``. Every function names the reference
lines it follows. Formulations are deliberately GEMM/token-major (the way the HIP path computes them)
rather than the reference's nn.Module/conv formulation, so agreement with the imported reference
(tests/golden) checks the math, not a copy.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _linear(sd, p, x):
    return x @ sd[p + "weight"].T + sd[p + "bias"]


def _ln(sd, p, x, eps):
    """LayerNorm over the last dim, biased variance (ref: torch.nn.LayerNorm; common.py:38-42 for the 2d form)."""
    mu = x.mean(-1, keepdim=True)
    var = (x - mu).pow(2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * sd[p + "weight"] + sd[p + "bias"]


def gelu_erf(x):
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


# ----------------------------------------------------------------------------------------
# image encoder
# ----------------------------------------------------------------------------------------

def patch_tokens(x, w, b):
    """Conv2d(k=s=patch) as a GEMM over non-overlapping patches.
    ref: lib/sam_model/image_encoder.py:386-394 (PatchEmbed: conv then NCHW->NHWC).
    x [B,3,H,W] -> [B, H/p, W/p, d]"""
    B, C, H, W = x.shape
    d, _, p, _ = w.shape
    gh, gw = H // p, W // p                     # a strided conv drops the ragged border (SO400M/14 at 384: 27x27, 6 px unused)
    x = x[:, :, :gh * p, :gw * p]
    cols = x.reshape(B, C, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(B * gh * gw, C * p * p)
    return (cols @ w.reshape(d, -1).T + b).reshape(B, gh, gw, d)


def rel_pos_bias(q, rel_h, rel_w, S):
    """Decomposed relative position bias, from the UNSCALED q.
    ref: lib/sam_model/image_encoder.py:293-323 (get_rel_pos, native-size branch: index = qi - ki + S-1),
         :326-362 (add_decomposed_rel_pos).
    q [N, S*S, hd]; rel_h/rel_w [2S-1, hd] -> bias [N, S*S, S*S]"""
    idx = torch.arange(S)[:, None] - torch.arange(S)[None, :] + (S - 1)  # [q, k]
    Rh, Rw = rel_h[idx], rel_w[idx]                                      # [S, S, hd]
    q4 = q.reshape(q.shape[0], S, S, -1)
    bh = torch.einsum("nhwc,hkc->nhwk", q4, Rh)                          # [N, qh, qw, kh]
    bw = torch.einsum("nhwc,wkc->nhwk", q4, Rw)                          # [N, qh, qw, kw]
    bias = bh[:, :, :, :, None] + bw[:, :, :, None, :]
    return bias.reshape(q.shape[0], S * S, S * S)


def vit_attention(sd, p, x, heads):
    """ref: lib/sam_model/image_encoder.py:225-241. x [N, S, S, d] -> [N, S, S, d]"""
    N, S, _, d = x.shape
    hd = d // heads
    qkv = _linear(sd, p + "qkv.", x.reshape(N, S * S, d))                # [N, T, 3d]
    qkv = qkv.reshape(N, S * S, 3, heads, hd).permute(2, 0, 3, 1, 4)     # [3, N, h, T, hd]
    q, k, v = (t.reshape(N * heads, S * S, hd) for t in qkv)
    logits = (q * hd ** -0.5) @ k.transpose(1, 2)
    logits = logits + rel_pos_bias(q, sd[p + "rel_pos_h"], sd[p + "rel_pos_w"], S)
    o = torch.softmax(logits, dim=-1) @ v                                # [N*h, T, hd]
    o = o.reshape(N, heads, S * S, hd).permute(0, 2, 1, 3).reshape(N, S, S, d)
    return _linear(sd, p + "proj.", o)


def to_windows(x, ws):
    """ref: lib/sam_model/image_encoder.py:244-265. zero-pad bottom/right to a multiple of ws, then tile."""
    B, H, W, C = x.shape
    Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
    xp = x.new_zeros(B, Hp, Wp, C)
    xp[:, :H, :W] = x
    xp = xp.reshape(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5)
    return xp.reshape(-1, ws, ws, C), (Hp, Wp)


def from_windows(w, ws, pad_hw, hw):
    """ref: lib/sam_model/image_encoder.py:268-290."""
    Hp, Wp = pad_hw
    H, W = hw
    B = w.shape[0] // ((Hp // ws) * (Wp // ws))
    x = w.reshape(B, Hp // ws, Wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, -1)
    return x[:, :H, :W]


def vit_block(sd, p, x, heads, window):
    """ref: lib/sam_model/image_encoder.py:169-185 (pad AFTER norm1), common.py:25-26 (exact-erf GELU)."""
    h = _ln(sd, p + "norm1.", x, 1e-6)
    if window > 0:
        H, W = h.shape[1:3]
        hw, pad_hw = to_windows(h, window)
        hw = vit_attention(sd, p + "attn.", hw, heads)
        h = from_windows(hw, window, pad_hw, (H, W))
    else:
        h = vit_attention(sd, p + "attn.", h, heads)
    x = x + h
    m = _ln(sd, p + "norm2.", x, 1e-6)
    m = _linear(sd, p + "mlp.lin2.", gelu_erf(_linear(sd, p + "mlp.lin1.", m)))
    return x + m


def neck(sd, p, x):
    """1x1 conv (no bias) -> LN2d -> 3x3 conv pad 1 (no bias) -> LN2d, computed channels-last.
    ref: lib/sam_model/image_encoder.py:86-102,117 ; common.py:31-43 (LayerNorm2d eps 1e-6).
    x [B,g,g,d] -> [B,256,g,g]"""
    y = x @ sd[p + "0.weight"][:, :, 0, 0].T
    y = _ln(sd, p + "1.", y, 1e-6)
    y = F.conv2d(y.permute(0, 3, 1, 2), sd[p + "2.weight"], padding=1).permute(0, 2, 3, 1)
    y = _ln(sd, p + "3.", y, 1e-6)
    return y.permute(0, 3, 1, 2).contiguous()


def image_encoder(sd, x, cfg, p="image_encoder.", return_tokens=False):
    """ref: lib/sam_model/image_encoder.py:109-119. x [B,3,1024,1024] -> [B,256,64,64]"""
    t = patch_tokens(x, sd[p + "patch_embed.proj.weight"], sd[p + "patch_embed.proj.bias"])
    t = t + sd[p + "pos_embed"]
    for i in range(cfg["depth"]):
        win = 0 if i in cfg["global_idx"] else cfg["window"]
        t = vit_block(sd, f"{p}blocks.{i}.", t, cfg["heads"], win)
    out = neck(sd, p + "neck.", t)
    return (out, t) if return_tokens else out


# ----------------------------------------------------------------------------------------
# prompt encoder (cut-down)
# ----------------------------------------------------------------------------------------

def dense_pe(sd, p="prompt_encoder.", size=64):
    """Random-Fourier dense positional encoding.
    ref: lib/sam_model/my_prompt_encoder.py:62-71,191-211. -> [1,256,size,size]"""
    G = sd[p + "pe_layer.positional_encoding_gaussian_matrix"]           # [2,128]
    c = (torch.arange(size, dtype=G.dtype) + 0.5) / size                 # cumsum(ones) - 0.5, / size (G's dtype: an fp64 state runs in fp64)
    xy = torch.stack([c[None, :].expand(size, size), c[:, None].expand(size, size)], dim=-1)  # (x, y)
    ang = 2 * math.pi * ((2 * xy - 1) @ G)                               # [size,size,128]
    pe = torch.cat([torch.sin(ang), torch.cos(ang)], dim=-1)
    return pe.permute(2, 0, 1).unsqueeze(0).contiguous()


def dense_no_mask(sd, B, p="prompt_encoder.", size=64):
    """ref: lib/sam_model/my_prompt_encoder.py:166-174."""
    return sd[p + "no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(B, -1, size, size)


# ----------------------------------------------------------------------------------------
# two-way transformer + mask decoder
# ----------------------------------------------------------------------------------------

def dec_attention(sd, p, q, k, v, heads=8):
    """ref: lib/sam_model/transformer.py:218-240 (scale AFTER q@k^T: /sqrt(c_per_head))."""
    q, k, v = _linear(sd, p + "q_proj.", q), _linear(sd, p + "k_proj.", k), _linear(sd, p + "v_proj.", v)
    B, Nq, C = q.shape
    c = C // heads
    sp = lambda t: t.reshape(B, t.shape[1], heads, c).transpose(1, 2)
    a = torch.softmax((sp(q) @ sp(k).transpose(2, 3)) / math.sqrt(c), dim=-1)
    o = (a @ sp(v)).transpose(1, 2).reshape(B, Nq, C)
    return _linear(sd, p + "out_proj.", o)


def two_way_block(sd, p, queries, keys, query_pe, key_pe, skip_first_layer_pe):
    """ref: lib/sam_model/transformer.py:151-182."""
    if skip_first_layer_pe:
        queries = dec_attention(sd, p + "self_attn.", queries, queries, queries)
    else:
        q = queries + query_pe
        queries = queries + dec_attention(sd, p + "self_attn.", q, q, queries)
    queries = _ln(sd, p + "norm1.", queries, 1e-5)
    q, k = queries + query_pe, keys + key_pe
    queries = _ln(sd, p + "norm2.", queries + dec_attention(sd, p + "cross_attn_token_to_image.", q, k, keys), 1e-5)
    mlp = _linear(sd, p + "mlp.lin2.", torch.relu(_linear(sd, p + "mlp.lin1.", queries)))
    queries = _ln(sd, p + "norm3.", queries + mlp, 1e-5)
    q, k = queries + query_pe, keys + key_pe
    keys = _ln(sd, p + "norm4.", keys + dec_attention(sd, p + "cross_attn_image_to_token.", k, q, queries), 1e-5)
    return queries, keys


def two_way_transformer(sd, p, src, pos, tokens, trace=None):
    """ref: lib/sam_model/transformer.py:62-106. src,pos [B,256,h,w]; tokens [B,Nt,256]"""
    keys = src.flatten(2).permute(0, 2, 1)
    key_pe = pos.flatten(2).permute(0, 2, 1)
    queries = tokens
    for i in range(2):
        queries, keys = two_way_block(sd, f"{p}layers.{i}.", queries, keys, tokens, key_pe, i == 0)
        if trace is not None:
            trace[f"tokens_l{i}"], trace[f"keys_l{i}"] = queries, keys
    q, k = queries + tokens, keys + key_pe
    queries = _ln(sd, p + "norm_final_attn.", queries + dec_attention(sd, p + "final_attn_token_to_image.", q, k, keys), 1e-5)
    return queries, keys


def _mlp3(sd, p, x):
    """ref: lib/sam_model/mask_decoder.py:147-167 (ReLU between, none after last)."""
    x = torch.relu(_linear(sd, p + "layers.0.", x))
    x = torch.relu(_linear(sd, p + "layers.1.", x))
    return _linear(sd, p + "layers.2.", x)


def conv_transpose_2x2(x, w, b):
    """ConvTranspose2d(k=2,s=2) on channels-last tokens as one GEMM + pixel shuffle.
    x [B,H,W,Cin]; w [Cin,Cout,2,2] -> [B,2H,2W,Cout]"""
    B, H, W, Cin = x.shape
    Cout = w.shape[1]
    y = x.reshape(-1, Cin) @ w.reshape(Cin, Cout * 4)                    # [.., (co,dy,dx)]
    y = y.reshape(B, H, W, Cout, 2, 2).permute(0, 1, 4, 2, 5, 3).reshape(B, 2 * H, 2 * W, Cout)
    return y + b


def mask_decoder(sd, image_embeddings, image_pe, sparse, dense, multimask_output, p="mask_decoder.", trace=None):
    """ref: lib/sam_model/mask_decoder.py:66-142. Returns (masks [B,C,256,256], iou [B,C], keys [B,4096,256])."""
    B = image_embeddings.shape[0]
    out_tok = torch.cat([sd[p + "iou_token.weight"], sd[p + "mask_tokens.weight"]], 0)   # [5,256]
    tokens = torch.cat([out_tok.unsqueeze(0).expand(B, -1, -1), sparse], dim=1)           # [B,6,256]
    src = image_embeddings + dense
    pos = image_pe.expand(B, -1, -1, -1)
    hs, keys = two_way_transformer(sd, p + "transformer.", src, pos, tokens, trace)
    iou_tok, mask_tok = hs[:, 0], hs[:, 1:5]
    # :132  src.transpose(1,2).view(B,-1,64,64): keys [B,4096,256] seen as NCHW == channels-last tokens [B,64,64,256]
    g = int(math.isqrt(keys.shape[1]))
    x = keys.reshape(B, g, g, -1)
    x = conv_transpose_2x2(x, sd[p + "output_upscaling.0.weight"], sd[p + "output_upscaling.0.bias"])
    x = gelu_erf(_ln(sd, p + "output_upscaling.1.", x, 1e-6))
    u1 = x
    x = gelu_erf(conv_transpose_2x2(x, sd[p + "output_upscaling.3.weight"], sd[p + "output_upscaling.3.bias"]))
    hyper = torch.stack([_mlp3(sd, f"{p}output_hypernetworks_mlps.{i}.", mask_tok[:, i]) for i in range(4)], 1)
    masks = torch.einsum("bkc,bhwc->bkhw", hyper, x)                     # [B,4,256,256]
    iou = _mlp3(sd, p + "iou_prediction_head.", iou_tok)
    if trace is not None:                                                # per-stage parity tables (tests)
        trace.update(hs=hs, upscaled1=u1, hyper=hyper, iou=iou, masks_all=masks)
    sl = slice(1, None) if multimask_output else slice(0, 1)             # :97-102
    return masks[:, sl], iou[:, sl], keys
