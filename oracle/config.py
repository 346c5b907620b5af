"""Architecture tables + parameter inventory used by the oracle (test infrastructure).

ref: lib/build_model.py:31-49 (SAM sizes), lib/support_branch.py:19-26 (SigLIP dims),
lib/support_model/siglip_openclip.py:93-110 (token grids).
SigLIP tower hyper-parameters come from the published open_clip model configs
(open_clip_torch==2.31.0, not vendored in the reference): unverified offline.
"""
from __future__ import annotations

import numpy as np
import torch

SAM = {
    "sam_base": dict(dim=768, depth=12, heads=12, global_idx=(2, 5, 8, 11)),
    "sam_large": dict(dim=1024, depth=24, heads=16, global_idx=(5, 11, 17, 23)),
    "sam_huge": dict(dim=1280, depth=32, heads=16, global_idx=(7, 15, 23, 31)),
}

SIGLIP = {
    "ViT-B-16-SigLIP-384": dict(dim=768, depth=12, heads=12, mlp=3072, patch=16, image=384,
                                vocab=32000, ctx=64, t_depth=12, t_heads=12, t_mlp=3072, v_gelu="erf", t_gelu="erf"),
    "ViT-B-16-SigLIP2-384": dict(dim=768, depth=12, heads=12, mlp=3072, patch=16, image=384,
                                 vocab=256000, ctx=64, t_depth=12, t_heads=12, t_mlp=3072, v_gelu="tanh", t_gelu="tanh"),
    "ViT-L-16-SigLIP-384": dict(dim=1024, depth=24, heads=16, mlp=4096, patch=16, image=384,
                                vocab=32000, ctx=64, t_depth=24, t_heads=16, t_mlp=4096, v_gelu="erf", t_gelu="erf"),
    "ViT-L-16-SigLIP2-384": dict(dim=1024, depth=24, heads=16, mlp=4096, patch=16, image=384,
                                 vocab=256000, ctx=64, t_depth=24, t_heads=16, t_mlp=4096, v_gelu="tanh", t_gelu="tanh"),
    "ViT-SO400M-14-SigLIP-384": dict(dim=1152, depth=27, heads=16, mlp=4304, patch=14, image=384,
                                     vocab=32000, ctx=64, t_depth=27, t_heads=16, t_mlp=4304, v_gelu="erf", t_gelu="erf"),
}


def sam_cfg(name: str) -> dict:
    if name not in SAM:
        raise ValueError(f"Invalid SAM model: {name}")  # ref: lib/build_model.py:49
    return dict(SAM[name], window=14, img=1024, patch=16, out=256)


def siglip_cfg(name: str) -> dict:
    if name not in SIGLIP:
        raise ValueError(f"Invalid SigLIP model: {name}")  # ref: lib/support_branch.py:26
    return dict(SIGLIP[name])


# ----------------------------------------------------------------------------------------
# parameter inventory (name -> shape), mirrors the reference state_dict key names
# ----------------------------------------------------------------------------------------

def _lin(d, p, o, i, bias=True):
    d[p + "weight"] = (o, i)
    if bias:
        d[p + "bias"] = (o,)


def _ln(d, p, c):
    d[p + "weight"] = (c,)
    d[p + "bias"] = (c,)


def sam_encoder_spec(cfg: dict, p="image_encoder.") -> dict:
    """ref: lib/sam_model/image_encoder.py:57-102,152-167,212-223,386."""
    d = {}
    dim, g = cfg["dim"], cfg["img"] // cfg["patch"]
    hd = dim // cfg["heads"]
    d[p + "pos_embed"] = (1, g, g, dim)
    d[p + "patch_embed.proj.weight"] = (dim, 3, cfg["patch"], cfg["patch"])
    d[p + "patch_embed.proj.bias"] = (dim,)
    for i in range(cfg["depth"]):
        b = f"{p}blocks.{i}."
        s = g if i in cfg["global_idx"] else cfg["window"]
        _ln(d, b + "norm1.", dim)
        d[b + "attn.rel_pos_h"] = (2 * s - 1, hd)
        d[b + "attn.rel_pos_w"] = (2 * s - 1, hd)
        _lin(d, b + "attn.qkv.", 3 * dim, dim)
        _lin(d, b + "attn.proj.", dim, dim)
        _ln(d, b + "norm2.", dim)
        _lin(d, b + "mlp.lin1.", 4 * dim, dim)
        _lin(d, b + "mlp.lin2.", dim, 4 * dim)
    d[p + "neck.0.weight"] = (cfg["out"], dim, 1, 1)
    _ln(d, p + "neck.1.", cfg["out"])
    d[p + "neck.2.weight"] = (cfg["out"], cfg["out"], 3, 3)
    _ln(d, p + "neck.3.", cfg["out"])
    return d


def prompt_encoder_spec(p="prompt_encoder.") -> dict:
    """ref: lib/sam_model/my_prompt_encoder.py:42,57,186-189."""
    return {p + "pe_layer.positional_encoding_gaussian_matrix": (2, 128),
            p + "no_mask_embed.weight": (1, 256)}


def _dec_attn(d, p, dim, internal):
    for n in ("q_proj", "k_proj", "v_proj"):
        _lin(d, f"{p}{n}.", internal, dim)
    _lin(d, p + "out_proj.", dim, internal)


def mask_decoder_spec(p="mask_decoder.") -> dict:
    """ref: lib/sam_model/mask_decoder.py:44-64, lib/sam_model/transformer.py:41-59,129-147,199-210."""
    d = {}
    t = p + "transformer."
    for i in range(2):
        L = f"{t}layers.{i}."
        _dec_attn(d, L + "self_attn.", 256, 256)
        _ln(d, L + "norm1.", 256)
        _dec_attn(d, L + "cross_attn_token_to_image.", 256, 128)
        _ln(d, L + "norm2.", 256)
        _lin(d, L + "mlp.lin1.", 2048, 256)
        _lin(d, L + "mlp.lin2.", 256, 2048)
        _ln(d, L + "norm3.", 256)
        _ln(d, L + "norm4.", 256)
        _dec_attn(d, L + "cross_attn_image_to_token.", 256, 128)
    _dec_attn(d, t + "final_attn_token_to_image.", 256, 128)
    _ln(d, t + "norm_final_attn.", 256)
    d[p + "iou_token.weight"] = (1, 256)
    d[p + "mask_tokens.weight"] = (4, 256)
    d[p + "output_upscaling.0.weight"] = (256, 64, 2, 2)
    d[p + "output_upscaling.0.bias"] = (64,)
    _ln(d, p + "output_upscaling.1.", 64)
    d[p + "output_upscaling.3.weight"] = (64, 32, 2, 2)
    d[p + "output_upscaling.3.bias"] = (32,)
    for i in range(4):
        m = f"{p}output_hypernetworks_mlps.{i}.layers."
        _lin(d, m + "0.", 256, 256)
        _lin(d, m + "1.", 256, 256)
        _lin(d, m + "2.", 32, 256)
    m = p + "iou_prediction_head.layers."
    _lin(d, m + "0.", 256, 256)
    _lin(d, m + "1.", 256, 256)
    _lin(d, m + "2.", 4, 256)
    return d


def mask_adapter_spec(D: int, p="support_branch.mask_pooling.", cin=512, mid_mask=16, mid=256, maps=8) -> dict:
    """ref: lib/support_model/mask_adapter.py:30-50,83-94,97-142,197-208; lib/support_branch.py:30-36."""
    d = {}
    c = p + "channel_clip_to_maskadapter."
    d[c + "conv.weight"] = (cin, D, 1, 1)
    d[c + "conv.bias"] = (cin,)
    _ln(d, c + "norm.", cin)
    g = p + "get_mask_map."
    d[g + "fuse.weight"] = (mid, cin, 1, 1)
    d[g + "fuse.bias"] = (mid,)
    for i in (1, 2, 3):
        b = f"{g}cnext{i}."
        d[b + "gamma"] = (mid,)
        d[b + "dwconv.weight"] = (mid, 1, 7, 7)
        d[b + "dwconv.bias"] = (mid,)
        _ln(d, b + "norm.", mid)
        _lin(d, b + "pwconv1.", 4 * mid, mid)
        _lin(d, b + "pwconv2.", mid, 4 * mid)
    _ln(d, g + "norm.", mid)
    d[g + "final.weight"] = (maps, mid, 1, 1)
    d[g + "final.bias"] = (maps,)
    md = g + "mask_downscaling."
    d[md + "0.weight"] = (mid_mask // 4, 1, 3, 3)
    d[md + "0.bias"] = (mid_mask // 4,)
    _ln(d, md + "1.", mid_mask // 4)
    d[md + "3.weight"] = (mid_mask, mid_mask // 4, 3, 3)
    d[md + "3.bias"] = (mid_mask,)
    _ln(d, md + "4.", mid_mask)
    d[md + "6.weight"] = (cin, mid_mask, 1, 1)
    d[md + "6.bias"] = (cin,)
    return d


def fuse_spec(D: int, p="support_branch.cir_fuse.") -> dict:
    """ref: lib/support_model/cir_feature_fuse.py:20-42."""
    d = {}
    _lin(d, p + "atten_Image.0.", D, 2 * D)
    _lin(d, p + "atten_Image.3.", D, D)
    _lin(d, p + "atten_Text.0.", D, 2 * D)
    _lin(d, p + "atten_Text.3.", D, D)
    _lin(d, p + "dynamic_scalar.0.", D, 2 * D)
    _lin(d, p + "dynamic_scalar.3.", 1, D)
    return d


def support_head_spec(D: int, p="support_branch.") -> dict:
    """ref: lib/support_branch.py:43-54."""
    d = {}
    _ln(d, p + "ln_channel_first.", D)
    _ln(d, p + "ln_channel_last.", D)
    _lin(d, p + "dim_proj.0.", 512, D)
    _lin(d, p + "dim_proj.3.", 256, 512)
    return d


def siglip_spec(cfg: dict, p="support_branch.siglip.model.", with_map_head=True) -> dict:
    """open_clip CustomTextCLIP(visual=TimmModel(trunk=timm VisionTransformer), text=TextTransformer)
    key names (open_clip_torch 2.31.0 / timm 1.0.15; from the published sources, unverified offline).
    Touch-points in the reference: lib/support_model/siglip_openclip.py:12,26,30-35,53."""
    d = {}
    D, P = cfg["dim"], (cfg["image"] // cfg["patch"]) ** 2
    v = p + "visual.trunk."
    d[v + "pos_embed"] = (1, P, D)
    d[v + "patch_embed.proj.weight"] = (D, 3, cfg["patch"], cfg["patch"])
    d[v + "patch_embed.proj.bias"] = (D,)
    for i in range(cfg["depth"]):
        b = f"{v}blocks.{i}."
        _ln(d, b + "norm1.", D)
        _lin(d, b + "attn.qkv.", 3 * D, D)
        _lin(d, b + "attn.proj.", D, D)
        _ln(d, b + "norm2.", D)
        _lin(d, b + "mlp.fc1.", cfg["mlp"], D)
        _lin(d, b + "mlp.fc2.", D, cfg["mlp"])
    _ln(d, v + "norm.", D)
    if with_map_head:  # MAP pooling head: dead on the live path but part of the checkpoint
        a = v + "attn_pool."
        d[a + "latent"] = (1, 1, D)
        _lin(d, a + "q.", D, D)
        _lin(d, a + "kv.", 2 * D, D)
        _lin(d, a + "proj.", D, D)
        _ln(d, a + "norm.", D)
        _lin(d, a + "mlp.fc1.", cfg["mlp"], D)
        _lin(d, a + "mlp.fc2.", D, cfg["mlp"])
    t = p + "text."
    d[t + "token_embedding.weight"] = (cfg["vocab"], D)
    d[t + "positional_embedding"] = (cfg["ctx"], D)
    for i in range(cfg["t_depth"]):
        b = f"{t}transformer.resblocks.{i}."
        _ln(d, b + "ln_1.", D)
        d[b + "attn.in_proj_weight"] = (3 * D, D)
        d[b + "attn.in_proj_bias"] = (3 * D,)
        _lin(d, b + "attn.out_proj.", D, D)
        _ln(d, b + "ln_2.", D)
        _lin(d, b + "mlp.c_fc.", cfg["t_mlp"], D)
        _lin(d, b + "mlp.c_proj.", D, cfg["t_mlp"])
    _ln(d, t + "ln_final.", D)
    _lin(d, t + "text_projection.", D, D)
    d[p + "logit_scale"] = ()
    d[p + "logit_bias"] = ()
    return d


def model_spec(sam_model: str, siglip_model: str, mask_pooling: str = "MaskAdapterPooling") -> dict:
    """Full CirSegModelWithQuerySupportFeat parameter/buffer inventory (persistent entries only).
    ref: lib/build_model.py:57-93 ; pixel_mean/std are non-persistent (sam_with_sup_branch.py:50-51)."""
    sc, gc = sam_cfg(sam_model), siglip_cfg(siglip_model)
    if mask_pooling not in ("MaskedPooling", "MaskAdapterPooling"):
        raise ValueError(f"Invalid mask pooling method: {mask_pooling}")  # ref: lib/support_branch.py:40
    d = {}
    d.update(sam_encoder_spec(sc))
    d.update(prompt_encoder_spec())
    d.update(siglip_spec(gc))
    if mask_pooling == "MaskAdapterPooling":
        d.update(mask_adapter_spec(gc["dim"]))
    d.update(fuse_spec(gc["dim"]))
    d.update(support_head_spec(gc["dim"]))
    d.update(mask_decoder_spec())
    return d


def random_state(spec: dict, seed: int, embed_rows_cap: int | None = None) -> dict:
    """Seeded, NON-degenerate parameters (numpy PCG64: stable across machines).
    Matrices ~ N(0, 1/sqrt(fan_in)) so activations stay O(1) through depth; LN weight / gamma ~ U(0.5,1.5);
    biases, pos-embeds, rel-pos tables ~ N(0, 0.1-0.5) (the reference defaults are zeros: SURVEY section 7.1)."""
    rng = np.random.default_rng(seed)
    out = {}
    for k in sorted(spec):
        shp = tuple(spec[k])
        if embed_rows_cap and k.endswith("token_embedding.weight") and shp[0] > embed_rows_cap:
            shp = (embed_rows_cap, shp[1])
        leaf = k.rsplit(".", 1)[-1]
        if leaf == "gamma" or (leaf == "weight" and len(shp) == 1):
            a = rng.uniform(0.5, 1.5, shp)
        elif leaf == "bias":
            a = rng.standard_normal(shp) * 0.1
        elif leaf in ("pos_embed", "positional_embedding", "rel_pos_h", "rel_pos_w", "latent"):
            a = rng.standard_normal(shp) * 0.5
        elif leaf == "positional_encoding_gaussian_matrix":
            a = rng.standard_normal(shp)
        elif leaf in ("logit_scale", "logit_bias"):
            a = rng.standard_normal(shp)
        elif k.endswith("token_embedding.weight") or k.endswith("iou_token.weight") or \
                k.endswith("mask_tokens.weight") or k.endswith("no_mask_embed.weight"):
            a = rng.standard_normal(shp)
        else:
            fan_in = int(np.prod(shp[1:])) if len(shp) > 1 else 1
            if "output_upscaling" in k and len(shp) == 4:  # ConvTranspose2d weight is [in, out, kh, kw]
                fan_in = shp[0]
            a = rng.standard_normal(shp) / np.sqrt(max(fan_in, 1))
        out[k] = torch.from_numpy(np.asarray(a, dtype=np.float32))
    return out
