"""HIP execution of the CORE forward path: weight packing + the kernel schedule.

Every arithmetic step is a launch of a hand-written gfx950 kernel through cor_amd.ops (C ABI in
include/cor_amd.h). torch only owns memory. Data layout in HBM:
  * activations are token-major [tokens, C] (channels-last) everywhere; NCHW exists only at the API boundary;
  * the residual stream is fp32; GEMM operands are `T` (torch.float32 = exact mode, torch.bfloat16 = fast mode);
  * weights are packed once per (state_dict version, T): [N,K] row-major in T, biases / LN / rel-pos in fp32.
The schedule follows lib/sam_with_sup_branch.py:57-104 of the reference; each function cites what it replaces.
"""
from __future__ import annotations

import math

import torch

from . import _native as nat
from . import ops
from .ops import ACT_NONE, ACT_GELU_ERF, ACT_RELU, ACT_SIGMOID, ACT_GELU_TANH

F32 = torch.float32


def _round_up(x, m):
    return (x + m - 1) // m * m


# =====================================================================================================
# packing
# =====================================================================================================
class _Packer:
    """Collects kernel-ready tensors from a state_dict (reference key names)."""

    def __init__(self, sd: dict, T: torch.dtype):
        dev = next(iter(sd.values())).device
        if dev.type != "cuda":
            raise RuntimeError("cor_amd: the model must live on a GPU (model.to('cuda')); there is no CPU path")
        self.sd, self.T, self.W = sd, T, {}

    def f32(self, k, t=None):
        self.W[k] = (self.sd[k] if t is None else t).detach().to(F32).contiguous()

    def mat(self, k, t=None):
        self.W[k] = (self.sd[k] if t is None else t).detach().to(self.T).contiguous()

    def lin(self, p):
        self.mat(p + "weight")
        self.f32(p + "bias")

    def ln(self, p):
        self.f32(p + "weight")
        self.f32(p + "bias")

    def patch_embed(self, p, dim, patch):
        kp = 3 * patch * patch
        K = _round_up(kp, 16)
        self.W[p + "patch.K"] = K
        w = self.sd[p + "patch_embed.proj.weight"].detach().reshape(dim, kp)
        if K != kp:
            w = torch.nn.functional.pad(w, (0, K - kp))
        self.mat(p + "patch_embed.proj.weight", w)
        self.f32(p + "patch_embed.proj.bias")
        self.f32(p + "pos_embed", self.sd[p + "pos_embed"].detach().reshape(-1, dim))


def pack_sam_encoder(pk: _Packer, cfg: dict, p="image_encoder."):
    """ref: lib/sam_model/image_encoder.py:57-102."""
    sd, W, d = pk.sd, pk.W, cfg["dim"]
    pk.patch_embed(p, d, cfg["patch"])
    for i in range(cfg["depth"]):
        b = f"{p}blocks.{i}."
        pk.ln(b + "norm1."); pk.ln(b + "norm2.")
        # bf16 MFMA attention (head_dim 64) runs the score product directly in the log2 domain: scale * log2(e) is folded into
        # the q rows of the qkv projection HERE (one rounding of the scaled weight; nothing at run time). fp32 mode: raw q.
        qs = nat.Q_PRESCALE_HD64 if (pk.T == torch.bfloat16 and d // cfg["heads"] == 64) else 1.0
        W[b + "attn.q_prescale"] = qs
        wq, bq = sd[b + "attn.qkv.weight"].detach().to(F32).clone(), sd[b + "attn.qkv.bias"].detach().to(F32).clone()
        wq[:d] *= qs; bq[:d] *= qs
        pk.mat(b + "attn.qkv.weight", wq); pk.f32(b + "attn.qkv.bias", bq)
        pk.lin(b + "attn.proj."); pk.lin(b + "mlp.lin1."); pk.lin(b + "mlp.lin2.")
        pk.f32(b + "attn.rel_pos_h"); pk.f32(b + "attn.rel_pos_w")
        pk.mat(b + "attn.pad_row", bq)                               # qkv of a zero (padded) token
    pk.mat(p + "neck.0.weight", sd[p + "neck.0.weight"].detach().reshape(cfg["out"], d))
    pk.ln(p + "neck.1."); pk.ln(p + "neck.3.")
    w3 = sd[p + "neck.2.weight"].detach()                            # [O, C, 3, 3] -> [O, (ky,kx,c)]
    pk.mat(p + "neck.2.weight", w3.permute(0, 2, 3, 1).reshape(w3.shape[0], -1))


def pack_prompt_encoder(pk: _Packer, p="prompt_encoder.", size=64):
    """ref: lib/sam_model/my_prompt_encoder.py:62-71,166-174: input independent -> folded at pack time."""
    g = pk.sd[p + "pe_layer.positional_encoding_gaussian_matrix"].detach().to(F32).contiguous()
    pk.W["prompt.dense_pe"] = ops.dense_pe(g, size)
    pk.f32("prompt.no_mask", pk.sd[p + "no_mask_embed.weight"].detach().reshape(-1))


def pack_siglip(pk: _Packer, g: dict, p="support_branch.siglip.model."):
    """open_clip / timm key names (see lib/support_model/siglip_openclip.py)."""
    v = p + "visual.trunk."
    pk.patch_embed(v, g["dim"], g["patch"])
    for i in range(g["depth"]):
        b = f"{v}blocks.{i}."
        pk.ln(b + "norm1."); pk.ln(b + "norm2.")
        pk.lin(b + "attn.qkv."); pk.lin(b + "attn.proj."); pk.lin(b + "mlp.fc1."); pk.lin(b + "mlp.fc2.")
    pk.ln(v + "norm.")
    t = p + "text."
    pk.f32(t + "token_embedding.weight"); pk.f32(t + "positional_embedding")
    for i in range(g["t_depth"]):
        b = f"{t}transformer.resblocks.{i}."
        pk.ln(b + "ln_1."); pk.ln(b + "ln_2.")
        pk.mat(b + "attn.in_proj_weight"); pk.f32(b + "attn.in_proj_bias")
        pk.lin(b + "attn.out_proj."); pk.lin(b + "mlp.c_fc."); pk.lin(b + "mlp.c_proj.")
    pk.ln(t + "ln_final."); pk.lin(t + "text_projection.")


def pack_mask_adapter(pk: _Packer, m="support_branch.mask_pooling."):
    """ref: lib/support_model/mask_adapter.py:30-50,83-94,97-142,197-208."""
    sd = pk.sd
    c = m + "channel_clip_to_maskadapter."
    pk.mat(c + "conv.weight", sd[c + "conv.weight"].detach().flatten(1)); pk.f32(c + "conv.bias"); pk.ln(c + "norm.")
    g = m + "get_mask_map."
    pk.mat(g + "fuse.weight", sd[g + "fuse.weight"].detach().flatten(1)); pk.f32(g + "fuse.bias")
    for i in (1, 2, 3):
        b = f"{g}cnext{i}."
        pk.f32(b + "dwconv.weight", sd[b + "dwconv.weight"].detach().reshape(-1, 49).t())   # [49, C] tap-major
        pk.f32(b + "dwconv.bias"); pk.ln(b + "norm."); pk.lin(b + "pwconv1."); pk.lin(b + "pwconv2."); pk.f32(b + "gamma")
    pk.ln(g + "norm.")
    pk.mat(g + "final.weight", sd[g + "final.weight"].detach().flatten(1)); pk.f32(g + "final.bias")
    md = g + "mask_downscaling."
    pk.f32(md + "0.weight"); pk.f32(md + "0.bias"); pk.ln(md + "1.")
    pk.f32(md + "3.weight"); pk.f32(md + "3.bias"); pk.ln(md + "4.")
    pk.mat(md + "6.weight", sd[md + "6.weight"].detach().flatten(1)); pk.f32(md + "6.bias")


def pack_fuse(pk: _Packer, f="support_branch.cir_fuse."):
    """ref: lib/support_model/cir_feature_fuse.py:20-42."""
    for n in ("atten_Image", "atten_Text", "dynamic_scalar"):
        pk.lin(f"{f}{n}.0."); pk.lin(f"{f}{n}.3.")


def pack_support_head(pk: _Packer, mask_pooling: str, s="support_branch."):
    """ref: lib/support_branch.py:43-54."""
    pk.ln(s + "ln_channel_first."); pk.ln(s + "ln_channel_last.")
    pk.lin(s + "dim_proj.0."); pk.lin(s + "dim_proj.3.")
    pack_fuse(pk, s + "cir_fuse.")
    if mask_pooling == "MaskAdapterPooling":
        pack_mask_adapter(pk, s + "mask_pooling.")


def pack_mask_decoder(pk: _Packer, q="mask_decoder."):
    """ref: lib/sam_model/mask_decoder.py:44-64, transformer.py:41-59,129-147."""
    sd = pk.sd
    tr = q + "transformer."
    attn_names = []
    for i in range(2):
        L = f"{tr}layers.{i}."
        attn_names += [L + "self_attn.", L + "cross_attn_token_to_image.", L + "cross_attn_image_to_token."]
        for n in ("norm1.", "norm2.", "norm3.", "norm4."):
            pk.ln(L + n)
        pk.lin(L + "mlp.lin1."); pk.lin(L + "mlp.lin2.")
    attn_names.append(tr + "final_attn_token_to_image.")
    for a in attn_names:
        for n in ("q_proj.", "k_proj.", "v_proj.", "out_proj."):
            pk.lin(a + n)
    pk.ln(tr + "norm_final_attn.")
    pk.f32(q + "out_tokens", torch.cat([sd[q + "iou_token.weight"], sd[q + "mask_tokens.weight"]], 0).detach().reshape(-1))
    w0 = sd[q + "output_upscaling.0.weight"].detach()                # [Cin, Cout, 2, 2] -> rows n = (dy,dx,co)
    pk.mat(q + "output_upscaling.0.weight", w0.permute(2, 3, 1, 0).reshape(4 * w0.shape[1], w0.shape[0]))
    pk.f32(q + "output_upscaling.0.bias", sd[q + "output_upscaling.0.bias"].detach().repeat(4))
    pk.ln(q + "output_upscaling.1.")
    pk.f32(q + "output_upscaling.3.weight"); pk.f32(q + "output_upscaling.3.bias")
    for i in range(4):
        for j in range(3):
            pk.lin(f"{q}output_hypernetworks_mlps.{i}.layers.{j}.")
    for j in range(3):
        pk.lin(f"{q}iou_prediction_head.layers.{j}.")
    # the five output MLPs stacked for ONE launch (cor_decoder_heads); MLP 4 = the IoU head (hidden width 256 like the others: build_model.py's default)
    names = [f"{q}output_hypernetworks_mlps.{i}.layers." for i in range(4)] + [f"{q}iou_prediction_head.layers."]
    # cor_decoder_heads' shapes: hidden width 256 twice, then 4 x [32, 256] (hyper-networks) + [4, 256] (IoU head); anything else keeps
    # the 15-GEMM path. Keys carry the decoder's prefix (two decoders packed into one W would otherwise collide).
    last = [tuple(sd[n + "2.weight"].shape) for n in names]
    if (all(tuple(sd[n + f"{j}.weight"].shape) == (256, 256) for n in names for j in (0, 1))
            and last == [(32, 256)] * 4 + [(4, 256)]):
        pk.mat(q + "heads.w01", torch.stack([torch.stack([sd[n + "0.weight"].detach(), sd[n + "1.weight"].detach()]) for n in names]))
        pk.f32(q + "heads.b01", torch.stack([torch.stack([sd[n + "0.bias"].detach(), sd[n + "1.bias"].detach()]) for n in names]))
        pk.mat(q + "heads.w2", torch.cat([sd[n + "2.weight"].detach() for n in names], 0))
        pk.f32(q + "heads.b2", torch.cat([sd[n + "2.bias"].detach() for n in names], 0))


def pack(sd: dict, scfg: dict, gcfg: dict, mask_pooling: str, T: torch.dtype) -> dict:
    """Full-model state_dict -> kernel-ready tensors. One-off, at load / after load_state_dict."""
    pk = _Packer(sd, T)
    pack_sam_encoder(pk, scfg)
    pack_prompt_encoder(pk)
    pack_siglip(pk, gcfg)
    pack_support_head(pk, mask_pooling)
    pack_mask_decoder(pk)
    return pk.W


# =====================================================================================================
# building blocks
# =====================================================================================================
def _lin(W, p, a, out_dtype, act=ACT_NONE, residual=None, col_scale=None, out=None, reverse=False):
    return ops.gemm(a, W[p + "weight"], out_dtype=out_dtype, bias=W[p + "bias"], act=act, residual=residual,
                    col_scale=col_scale, out=out, reverse=reverse)


def _ln(W, p, x, eps, out_dtype, act=ACT_NONE, reverse=False):
    return ops.layernorm(x, W[p + "weight"], W[p + "bias"], eps, out_dtype=out_dtype, act=act, reverse=reverse)


def _gelu(kind):
    return ACT_GELU_ERF if kind == "erf" else ACT_GELU_TANH


def sam_encoder(W, img, cfg, T, p="image_encoder."):
    """ref: lib/sam_model/image_encoder.py:109-119. img fp32 [B,3,1024,1024] -> tokens fp32 [B*g*g, 256]."""
    B = img.shape[0]
    g, d, H = cfg["img"] // cfg["patch"], cfg["dim"], cfg["heads"]
    assert img.shape[1:] == (3, cfg["img"], cfg["img"]), f"SAM input must be [B,3,{cfg['img']},{cfg['img']}], got {tuple(img.shape)}"
    cols = ops.patchify(img, cfg["patch"], W[p + "patch.K"], T)
    x = ops.gemm(cols, W[p + "patch_embed.proj.weight"], out_dtype=F32, bias=W[p + "patch_embed.proj.bias"],
                 residual=W[p + "pos_embed"], res_row_mod=g * g)                        # :110-112
    del cols
    # Work order along the chain: the GEMMs write their outputs from the first row panel to the last, so the kernels that
    # consume a GEMM's output (LayerNorm over the 403 MB residual, attention over the 604 MB qkv) walk it from the LAST row to
    # the first - they start on the ~200 MB the producer left in the 256 MB Infinity Cache and finish on the low rows, where
    # the next GEMM starts (measured +0.5 ... +2.5 % end to end depending on the box, profiles/archive/r02_work_order_ab.txt).
    for i in range(cfg["depth"]):
        b = f"{p}blocks.{i}."
        win = 0 if i in cfg["global_idx"] else cfg["window"]
        h = _ln(W, b + "norm1.", x, 1e-6, T, reverse=True)                                # :169
        qkv = _lin(W, b + "attn.qkv.", h, T)                                              # :229
        a = ops.sam_attention(qkv, W[b + "attn.pad_row"], W[b + "attn.rel_pos_h"], W[b + "attn.rel_pos_w"], B, H, g, win,
                              q_prescale=W[b + "attn.q_prescale"], reverse=True)            # :172-180,232-238
        del qkv
        _lin(W, b + "attn.proj.", a, F32, residual=x, out=x)                              # :239,182
        h = _ln(W, b + "norm2.", x, 1e-6, T, reverse=True)
        m = _lin(W, b + "mlp.lin1.", h, T, act=ACT_GELU_ERF)                              # common.py:25-26
        _lin(W, b + "mlp.lin2.", m, F32, residual=x, out=x)                               # :183
        del h, a, m
    y = ops.gemm(ops.cast(x, T), W[p + "neck.0.weight"], out_dtype=F32)                   # :87-92 1x1 conv, no bias
    y = _ln(W, p + "neck.1.", y, 1e-6, T)
    y = ops.gemm(ops.im2col3x3(y, B, g, g), W[p + "neck.2.weight"], out_dtype=F32)        # :94-100 3x3 conv, no bias
    return _ln(W, p + "neck.3.", y, 1e-6, F32)


def _vit_mha(qkv, N, Tn, heads, D, T):
    hd = D // heads
    return ops.attention(qkv[:, 0:D], qkv[:, D:2 * D], qkv[:, 2 * D:3 * D], N, heads, Tn, Tn, hd, hd ** -0.5, out_dtype=T)


def siglip_vision(W, img, g, T, p="support_branch.siglip.model.visual.trunk."):
    """ref: lib/support_model/siglip_openclip.py:30-35 (trunk run ONCE; the MAP-head pass of :26 is dead on the
    live path). img fp32 [N,3,384,384] -> last hidden states fp32 [N*P, D]."""
    N = img.shape[0]
    D, P = g["dim"], (g["image"] // g["patch"]) ** 2
    assert img.shape[1:] == (3, g["image"], g["image"]), f"SigLIP input must be [N,3,{g['image']},{g['image']}], got {tuple(img.shape)}"
    cols = ops.patchify(img, g["patch"], W[p + "patch.K"], T)
    x = ops.gemm(cols, W[p + "patch_embed.proj.weight"], out_dtype=F32, bias=W[p + "patch_embed.proj.bias"],
                 residual=W[p + "pos_embed"], res_row_mod=P)
    act = _gelu(g.get("v_gelu", g.get("gelu", "erf")))
    for i in range(g["depth"]):
        b = f"{p}blocks.{i}."
        h = _ln(W, b + "norm1.", x, 1e-6, T)
        qkv = _lin(W, b + "attn.qkv.", h, T)
        a = _vit_mha(qkv, N, P, g["heads"], D, T)
        _lin(W, b + "attn.proj.", a, F32, residual=x, out=x)
        h = _ln(W, b + "norm2.", x, 1e-6, T)
        m = _lin(W, b + "mlp.fc1.", h, T, act=act)
        _lin(W, b + "mlp.fc2.", m, F32, residual=x, out=x)
    return _ln(W, p + "norm.", x, 1e-6, F32)


def siglip_text(W, tokens, g, T, p="support_branch.siglip.model.text."):
    """ref: lib/support_model/siglip_openclip.py:46-59 (encode_text, pool = last token, then F.normalize).
    tokens int64 [N,64] -> fp32 [N, D] unit-norm."""
    N, ctx = tokens.shape
    D = g["dim"]
    x = ops.embed_tokens(tokens.contiguous(), W[p + "token_embedding.weight"], W[p + "positional_embedding"])
    act = _gelu(g.get("t_gelu", g.get("gelu", "erf")))
    for i in range(g["t_depth"]):
        b = f"{p}transformer.resblocks.{i}."
        h = _ln(W, b + "ln_1.", x, 1e-6, T)
        qkv = ops.gemm(h, W[b + "attn.in_proj_weight"], out_dtype=T, bias=W[b + "attn.in_proj_bias"])
        a = _vit_mha(qkv, N, ctx, g["t_heads"], D, T)
        _lin(W, b + "attn.out_proj.", a, F32, residual=x, out=x)
        h = _ln(W, b + "ln_2.", x, 1e-6, T)
        m = _lin(W, b + "mlp.c_fc.", h, T, act=act)
        _lin(W, b + "mlp.c_proj.", m, F32, residual=x, out=x)
    last = torch.empty((N, D), dtype=F32, device=x.device)
    ops.copy_rows(x, ctx * D, N, D, last, src_offset=(ctx - 1) * D)                       # pool_type 'last'
    last = _ln(W, p + "ln_final.", last, 1e-6, T)
    feat = _lin(W, p + "text_projection.", last, F32)
    return ops.l2norm_rows(feat)


def mask_adapter_pooling(W, feat, mask, N, gh, D, T, p="support_branch.mask_pooling."):
    """ref: lib/support_model/mask_adapter.py:52-80 (+ :83-94, :144-179, :210-223).
    feat fp32 tokens [N*P, D] (already ln_channel_first-ed), mask fp32 [N,1,h,w] -> pooled fp32 [N, D]."""
    P = gh * gh
    m24 = ops.bilinear(mask, gh, gh) if tuple(mask.shape[-2:]) != (gh, gh) else mask     # :57-58
    c = p + "channel_clip_to_maskadapter."
    dense = ops.gemm(ops.cast(feat, T), W[c + "conv.weight"], out_dtype=F32, bias=W[c + "conv.bias"])
    dense = _ln(W, c + "norm.", dense, 1e-6, F32, act=ACT_GELU_ERF)                       # :90-93
    g = p + "get_mask_map."
    md = g + "mask_downscaling."
    m96 = ops.bilinear(m24, 4 * gh, 4 * gh)                                               # :158
    c1 = ops.conv3x3s2_small(m96, False, W[md + "0.weight"], W[md + "0.bias"], N, 1, 4 * gh, 4 * gh)
    c1 = _ln(W, md + "1.", c1.view(-1, c1.shape[-1]), 1e-6, F32, act=ACT_GELU_ERF)
    c2 = ops.conv3x3s2_small(c1, True, W[md + "3.weight"], W[md + "3.bias"], N, W[md + "3.weight"].shape[1], 2 * gh, 2 * gh)
    c2 = _ln(W, md + "4.", c2.view(-1, c2.shape[-1]), 1e-6, T, act=ACT_GELU_ERF)
    summed = ops.gemm(c2, W[md + "6.weight"], out_dtype=F32, bias=W[md + "6.bias"], residual=dense)   # :159-161
    y = ops.gemm(ops.cast(summed, T), W[g + "fuse.weight"], out_dtype=F32, bias=W[g + "fuse.bias"])   # :163
    for i in (1, 2, 3):                                                                   # :210-223
        b = f"{g}cnext{i}."
        dw = ops.dwconv7x7(y, W[b + "dwconv.weight"], W[b + "dwconv.bias"], N, gh, gh)
        h = _ln(W, b + "norm.", dw, 1e-6, T)
        h = _lin(W, b + "pwconv1.", h, T, act=ACT_GELU_ERF)
        _lin(W, b + "pwconv2.", h, F32, residual=y, col_scale=W[b + "gamma"], out=y)
    h = _ln(W, g + "norm.", y, 1e-6, T)
    maps = ops.gemm(h, W[g + "final.weight"], out_dtype=F32, bias=W[g + "final.bias"])    # [N*P, 8]
    M = maps.shape[1]
    # :62-67 interpolate to the same size is the identity (scale 1 => lambda 0)
    return ops.adapter_pool(maps, feat, N, P, M, D), maps


def support_head(W, vis_tokens, text_feat, mask, g, mask_pooling, T, p="support_branch."):
    """ref: lib/support_branch.py:58-62,65-66,85-86 ; lib/support_model/cir_feature_fuse.py:44-64.
    vis_tokens fp32 [N*P, D] (trunk.norm output); -> comb_support_feat fp32 [N, 256] unit-norm."""
    D = g["dim"]
    gh = g["image"] // g["patch"]
    N = vis_tokens.shape[0] // (gh * gh)
    feat = _ln(W, p + "ln_channel_first.", vis_tokens, 1e-6, F32)                         # LN over C of NCHW == row LN of tokens
    if mask_pooling == "MaskAdapterPooling":
        pooled, _ = mask_adapter_pooling(W, feat, mask, N, gh, D, T, p + "mask_pooling.")
    else:                                                                                  # mask_adapter.py:13-25
        m = ops.bilinear(mask, gh, gh) if tuple(mask.shape[-2:]) != (gh, gh) else mask
        pooled = ops.masked_pool(feat, m, N, gh * gh, D)
    img = _ln(W, p + "ln_channel_last.", pooled, 1e-6, F32)
    f = p + "cir_fuse."
    raw = torch.empty((N, 2 * D), dtype=T, device=img.device)
    ops.copy_rows(img, D, N, D, raw, ld_out=2 * D)
    ops.copy_rows(text_feat, D, N, D, raw[:, D:], ld_out=2 * D)

    def gate(name, x):
        h = _lin(W, f"{f}{name}.0.", x, T, act=ACT_RELU)
        return _lin(W, f"{f}{name}.3.", h, F32, act=ACT_SIGMOID)

    cat = ops.fuse_gate(img, text_feat, gate("atten_Image", raw), gate("atten_Text", raw))
    dyn = gate("dynamic_scalar", ops.cast(cat, T))
    fused = ops.fuse_mix(cat, dyn)
    h = _lin(W, p + "dim_proj.0.", ops.cast(fused, T), T, act=ACT_GELU_ERF)               # Dropout(0.8): identity in eval
    h = _lin(W, p + "dim_proj.3.", h, F32, act=ACT_GELU_ERF)
    return ops.l2norm_rows(h)


def _dec_attn(W, p, q, k, v, B, Tq, Tk, T, residual=None, out=None):
    """ref: lib/sam_model/transformer.py:218-240. q [B*Tq,256], k,v [B*Tk,256] in T -> fp32 [B*Tq,256] (+residual)."""
    qp, kp, vp = _lin(W, p + "q_proj.", q, T), _lin(W, p + "k_proj.", k, T), _lin(W, p + "v_proj.", v, T)
    internal = qp.shape[1]
    hd = internal // 8
    o = ops.attention(qp, kp, vp, B, 8, Tq, Tk, hd, 1.0 / math.sqrt(hd), out_dtype=T)
    return _lin(W, p + "out_proj.", o, F32, residual=residual, out=out)


def mask_decoder(W, emb_tokens, feat, T, multimask_output, all_masks=False, p="mask_decoder.", trace=None, fused_heads=True):
    """ref: lib/sam_model/mask_decoder.py:107-142 + transformer.py:62-106,151-182 + sam_with_sup_branch.py:96-100.
    emb_tokens fp32 [B*4096,256], feat fp32 [B,256] -> (final_masks [B,1,256,256], iou [B,4], best [B], masks_all|None)."""
    B = feat.shape[0]
    Tk, Tq, C = emb_tokens.shape[0] // B, 6, 256
    g = int(math.isqrt(Tk))
    tokens = torch.empty((B * Tq, C), dtype=F32, device=feat.device)                      # :113-115 cat(iou, mask x4, prompt)
    ops.copy_rows(W[p + "out_tokens"], 0, B, 5 * C, tokens.view(B, Tq * C), ld_out=Tq * C)
    ops.copy_rows(feat, C, B, C, tokens.view(B, Tq * C)[:, 5 * C:], ld_out=Tq * C)
    keys = ops.add(emb_tokens, W["prompt.no_mask"])                                        # :118 src = emb + dense
    key_pe = W["prompt.dense_pe"]
    tr = p + "transformer."
    queries = tokens
    tok_T = ops.cast(tokens, T)
    for i in range(2):
        L = f"{tr}layers.{i}."
        if i == 0:                                                                         # skip_first_layer_pe
            queries = _dec_attn(W, L + "self_attn.", tok_T, tok_T, tok_T, B, Tq, Tq, T)
        else:
            qpe = ops.add(queries, tokens, out_dtype=T)
            queries = _dec_attn(W, L + "self_attn.", qpe, qpe, ops.cast(queries, T), B, Tq, Tq, T, residual=queries)
        queries = _ln(W, L + "norm1.", queries, 1e-5, F32)
        qpe = ops.add(queries, tokens, out_dtype=T)
        kpe = ops.add(keys, key_pe, out_dtype=T)
        keys_T = ops.cast(keys, T)
        queries = _dec_attn(W, L + "cross_attn_token_to_image.", qpe, kpe, keys_T, B, Tq, Tk, T, residual=queries)
        queries = _ln(W, L + "norm2.", queries, 1e-5, F32)
        h = _lin(W, L + "mlp.lin1.", ops.cast(queries, T), T, act=ACT_RELU)
        queries = _lin(W, L + "mlp.lin2.", h, F32, residual=queries)
        queries = _ln(W, L + "norm3.", queries, 1e-5, F32)
        qpe = ops.add(queries, tokens, out_dtype=T)
        keys = _dec_attn(W, L + "cross_attn_image_to_token.", kpe, qpe, ops.cast(queries, T), B, Tk, Tq, T, residual=keys)
        keys = _ln(W, L + "norm4.", keys, 1e-5, F32)
        if trace is not None:                                                              # per-stage parity tables (tests only)
            trace[f"tokens_l{i}"], trace[f"keys_l{i}"] = queries, keys
    qpe = ops.add(queries, tokens, out_dtype=T)
    kpe = ops.add(keys, key_pe, out_dtype=T)
    keys_T = ops.cast(keys, T)
    queries = _dec_attn(W, tr + "final_attn_token_to_image.", qpe, kpe, keys_T, B, Tq, Tk, T, residual=queries)
    hs = _ln(W, tr + "norm_final_attn.", queries, 1e-5, T)                                # [B*6, 256]

    # hyper-network MLPs on mask tokens 1..4, IoU head on token 0 (:123-140); rows picked by lda = 6*256
    if (p + "heads.w01") in W and fused_heads:                                                   # one launch instead of 15 (143 -> ~15 us at batch 32)
        hyper, iou = ops.decoder_heads(hs, W[p + "heads.w01"], W[p + "heads.b01"], W[p + "heads.w2"], W[p + "heads.b2"])
    else:
        hs3 = hs.view(B, Tq * C)
        hyper = torch.empty((B, 4, 32), dtype=F32, device=feat.device)
        for i in range(4):
            m = f"{p}output_hypernetworks_mlps.{i}.layers."
            a = hs3[:, (1 + i) * C:(2 + i) * C]
            a = _lin(W, m + "0.", a, T, act=ACT_RELU)
            a = _lin(W, m + "1.", a, T, act=ACT_RELU)
            _lin(W, m + "2.", a, F32, out=hyper.view(B, 128)[:, 32 * i:32 * (i + 1)])
        m = p + "iou_prediction_head.layers."
        a = _lin(W, m + "0.", hs3[:, 0:C], T, act=ACT_RELU)
        a = _lin(W, m + "1.", a, T, act=ACT_RELU)
        iou = _lin(W, m + "2.", a, F32)                                                    # [B,4]

    # upscaling (:132-137): keys [B*4096,256] are already the channels-last view of `src`
    y = _lin(W, p + "output_upscaling.0.", keys_T, T)                                      # ConvT 2x2 as GEMM, n=(dy,dx,co)
    u1 = ops.upscale_shuffle(y, B, g, g, 64, ln_w=W[p + "output_upscaling.1.weight"], ln_b=W[p + "output_upscaling.1.bias"],
                             eps=1e-6, act=ACT_GELU_ERF)
    w3, b3 = W[p + "output_upscaling.3.weight"], W[p + "output_upscaling.3.bias"]
    k_off, ksel = (1, 3) if multimask_output else (0, 1)                                   # :97-102
    best, hyper_sel = ops.iou_select(iou, hyper, k_off, ksel)
    final = ops.upscale_hyper(u1, w3, b3, hyper_sel, B, 2 * g, 2 * g, 1)
    masks_all = ops.upscale_hyper(u1, w3, b3, hyper, B, 2 * g, 2 * g, 4) if all_masks else None
    if trace is not None:
        trace.update(hs=hs, upscaled1=u1, hyper=hyper, iou=iou, masks_all=masks_all)
    return final, iou, best, masks_all, keys


OVERLAP_BRANCHES = False    # default of forward(overlap_branches=None); True: support branch on a second HIP stream (+3-4.5 % end to end; per-kernel
                            # event timings are then contended, so the eager default stays off; a captured graph (model.capture) turns it on)
_SIDE = {}


def _side_stream(device, which=0):
    key = (device.type, device.index, which)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


def forward_support(W, gcfg, mask_pooling, T, support_image_inputs, change_text_inputs, support_mask_inputs, two_chains=True, text_stream=None):
    """The support branch on the CURRENT stream (ref: lib/sam_with_sup_branch.py:79-80 -> lib/support_branch.py:56-87): SigLIP towers,
    mask adapter, fusion, dim_proj -> comb_support_feat fp32 [B,256]. two_chains: the text tower on one more stream (the towers are
    independent: each is a CHAIN of ~110 small dependent kernels, and a chain advances only where the encoder's persistent GEMMs
    leave CUs free, 2-3 kernels per big-kernel boundary). text_stream: a stream the CALLER has already forked from the stream its own
    stream was forked from (a fork of a forked stream inside a graph capture crashed hipGraph's capture_end on ROCm 7.0: forks stay flat)."""
    s_img = support_image_inputs.to(F32).contiguous()
    s_mask = support_mask_inputs.to(F32).contiguous()
    if not two_chains:
        vis = siglip_vision(W, s_img, gcfg, T)
        txt = siglip_text(W, change_text_inputs.to(s_img.device), gcfg, T)
        return support_head(W, vis, txt, s_mask, gcfg, mask_pooling, T)
    cur = torch.cuda.current_stream()
    side2 = text_stream
    if side2 is None:
        if torch.cuda.is_current_stream_capturing():
            # Under a capture the current stream may itself be a fork of the capture's origin stream; forking again from it made
            # hipGraph's capture_end crash (ROCm 7.0, profiles/r05_capture_nested_fork_record.txt). The caller did not provide a flat
            # fork (text_stream), so both towers stay on this stream: same results, no second chain.
            vis = siglip_vision(W, s_img, gcfg, T)
            txt = siglip_text(W, change_text_inputs.to(s_img.device), gcfg, T)
            return support_head(W, vis, txt, s_mask, gcfg, mask_pooling, T)
        side2 = _side_stream(s_img.device, 1)
        side2.wait_stream(cur)
    with torch.cuda.stream(side2):
        txt = siglip_text(W, change_text_inputs.to(s_img.device), gcfg, T)
    vis = siglip_vision(W, s_img, gcfg, T)
    cur.wait_stream(side2)
    txt.record_stream(cur)
    return support_head(W, vis, txt, s_mask, gcfg, mask_pooling, T)


def forward_decode(W, scfg, T, emb_tokens, feat, multimask_output=True, return_aux=False):
    """Mask decoder + the API's output layout (ref: lib/sam_with_sup_branch.py:82-104) from the encoder's tokens fp32 [B*4096,256]
    and comb_support_feat fp32 [B,256]."""
    B = feat.shape[0]
    final, iou, best, masks_all, _ = mask_decoder(W, emb_tokens, feat, T, multimask_output, all_masks=return_aux)   # :82-100
    g = scfg["img"] // scfg["patch"]
    emb = ops.tokens_to_nchw(emb_tokens, B, g * g, scfg["out"]).view(B, scfg["out"], g, g)
    out = (final, emb, feat.view(B, 1, -1))
    if return_aux:
        return out + (dict(masks=masks_all, iou=iou, best=best),)
    return out


def forward(W, scfg, gcfg, mask_pooling, T, query_image_inputs, support_image_inputs, change_text_inputs, support_mask_inputs,
            multimask_output=True, return_aux=False, overlap_branches=None):
    """ref: lib/sam_with_sup_branch.py:57-104. overlap_branches (None = module default OVERLAP_BRANCHES): per-call choice."""
    emb_tokens, feat = forward_encode(W, scfg, gcfg, mask_pooling, T, query_image_inputs, support_image_inputs, change_text_inputs,
                                      support_mask_inputs, overlap_branches)
    return forward_decode(W, scfg, T, emb_tokens, feat, multimask_output, return_aux)


def forward_encode(W, scfg, gcfg, mask_pooling, T, query_image_inputs, support_image_inputs, change_text_inputs, support_mask_inputs,
                   overlap_branches=None):
    """Everything up to the mask decoder (ref: lib/sam_with_sup_branch.py:76-80): SAM encoder tokens fp32 [B*4096,256] and
    comb_support_feat fp32 [B,256], both complete on the current stream on return. forward() = forward_encode + forward_decode; the
    two halves are captured as separate graphs by ForwardPipeline(stagger=True)."""
    if overlap_branches is None:
        overlap_branches = OVERLAP_BRANCHES
    q_img = query_image_inputs.to(F32).contiguous()
    if overlap_branches:
        # The support branch (SigLIP towers + ~100 tiny adapter/fusion kernels) is independent of the SAM encoder until
        # the mask decoder: enqueue it on a second HIP stream so its latency-bound kernels fill CUs the encoder leaves idle.
        # "one_chain": both towers on that stream (rounds 2-3, kept for A/B runs: bench.py --overlap 1); otherwise two chains.
        main = torch.cuda.current_stream()
        side, side2 = _side_stream(q_img.device), None
        side.wait_stream(main)
        if overlap_branches != "one_chain":
            side2 = _side_stream(q_img.device, 1)
            side2.wait_stream(main)
        with torch.cuda.stream(side):
            feat = forward_support(W, gcfg, mask_pooling, T, support_image_inputs, change_text_inputs, support_mask_inputs,
                                   two_chains=side2 is not None, text_stream=side2)
        emb_tokens = sam_encoder(W, q_img, scfg, T)                                        # :76
        main.wait_stream(side)
        feat.record_stream(main)
    else:
        emb_tokens = sam_encoder(W, q_img, scfg, T)
        feat = forward_support(W, gcfg, mask_pooling, T, support_image_inputs, change_text_inputs, support_mask_inputs, two_chains=False)
    return emb_tokens, feat
