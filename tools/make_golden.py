#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own modules (imported from /root/reference,
read-only, never copied) on seeded inputs with seeded NON-degenerate parameters.

Runs only in the build container (the reference does not travel to the GPU box). What is committed is data:
seeds, shapes and fp32 outputs (sub-sampled where large, plus float64 full-tensor moments).
Parameters and inputs are re-created at test time from the same numpy PCG64 seeds
(oracle.config.random_state / tests.golden_util), so fixtures stay small.

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py [--only name ...]
"""
from __future__ import annotations

import argparse
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(1, "/root/reference")
sys.dont_write_bytecode = True

from oracle import config as ocfg  # noqa: E402
from tests.golden_util import GOLDEN_DIR, make_inputs, moments, strided  # noqa: E402

torch.set_grad_enabled(False)


def save(name, **arrays):
    path = os.path.join(GOLDEN_DIR, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"  wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")


def load_strict(module, sd, prefix):
    sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    module.load_state_dict(sub, strict=True)   # strict: also pins oracle.config's key inventory
    return module.eval()


# ------------------------------------------------------------------------------------------
def gen_sam_attention():
    """(i) Attention with non-zero rel_pos: windowed 14x14 and a 16x16 'global' grid. image_encoder.py:225-241"""
    from lib.sam_model.image_encoder import Attention
    for tag, S in (("win14", 14), ("glob16", 16)):
        dim, heads = 64, 2
        cfg = dict(dim=dim, heads=heads, depth=1, global_idx=(0,), window=14, img=S * 16, patch=16, out=16)
        spec = {k: v for k, v in ocfg.sam_encoder_spec(cfg, "e.").items() if k.startswith("e.blocks.0.attn.")}
        sd = ocfg.random_state(spec, seed=101)
        m = load_strict(Attention(dim, heads, True, True, True, (S, S)), sd, "e.blocks.0.attn.")
        x = make_inputs(102, x=(3, S, S, dim))["x"]
        save(f"sam_attention_{tag}", seed_params=101, seed_inputs=102, S=S, dim=dim, heads=heads, y=m(x).numpy())


def gen_sam_block():
    """(ii) Block with window padding (grid 20 -> pad 28) and a global block. image_encoder.py:169-185"""
    from lib.sam_model.image_encoder import Block
    from functools import partial
    dim, heads, g = 32, 2, 20
    for tag, win in (("window", 14), ("global", 0)):
        cfg = dict(dim=dim, heads=heads, depth=1, global_idx=(0,) if win == 0 else (), window=14, img=g * 16, patch=16, out=16)
        spec = {k: v for k, v in ocfg.sam_encoder_spec(cfg, "e.").items() if k.startswith("e.blocks.0.")}
        sd = ocfg.random_state(spec, seed=111)
        blk = Block(dim, heads, 4.0, True, partial(torch.nn.LayerNorm, eps=1e-6), torch.nn.GELU, True, True, win, (g, g))
        m = load_strict(blk, sd, "e.blocks.0.")
        x = make_inputs(112, x=(2, g, g, dim))["x"]
        save(f"sam_block_{tag}", seed_params=111, seed_inputs=112, g=g, dim=dim, heads=heads, window=win, y=m(x).numpy())


def gen_sam_encoder():
    """(iii) full tiny ImageEncoderViT: img 1024 (grid 64, pad 70) d=32 depth=2, and img 256 d=64 depth=3."""
    from lib.sam_model.image_encoder import ImageEncoderViT
    from functools import partial
    for tag, img, dim, depth, heads, gidx, out, B in (("img1024", 1024, 32, 2, 2, (1,), 16, 1),
                                                      ("img256", 256, 64, 3, 2, (2,), 32, 2)):
        cfg = dict(dim=dim, heads=heads, depth=depth, global_idx=gidx, window=14, img=img, patch=16, out=out)
        sd = ocfg.random_state(ocfg.sam_encoder_spec(cfg), seed=121)
        enc = ImageEncoderViT(img_size=img, patch_size=16, embed_dim=dim, depth=depth, num_heads=heads, mlp_ratio=4,
                              out_chans=out, qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6),
                              use_rel_pos=True, window_size=14, global_attn_indexes=gidx)
        m = load_strict(enc, sd, "image_encoder.")
        x = make_inputs(122, x=(B, 3, img, img))["x"]
        y = m(x)
        save(f"sam_encoder_{tag}", seed_params=121, seed_inputs=122, img=img, dim=dim, depth=depth, heads=heads,
             global_idx=np.array(gidx), out=out, B=B, y=y.numpy(), y_moments=moments(y))


def gen_decoder():
    """(iv) PromptEncoder + MaskDecoder + TwoWayTransformer at real dims, B=2, both multimask values."""
    from lib.sam_model.mask_decoder import MaskDecoder
    from lib.sam_model.transformer import TwoWayTransformer
    from lib.sam_model.my_prompt_encoder import PromptEncoder
    spec = dict(ocfg.mask_decoder_spec(), **ocfg.prompt_encoder_spec())
    sd = ocfg.random_state(spec, seed=131)
    dec = MaskDecoder(num_multimask_outputs=3, transformer=TwoWayTransformer(depth=2, embedding_dim=256, mlp_dim=2048, num_heads=8),
                      transformer_dim=256, iou_head_depth=3, iou_head_hidden_dim=256)
    dec = load_strict(dec, sd, "mask_decoder.")
    pe = load_strict(PromptEncoder(embed_dim=256, image_embedding_size=(64, 64)), sd, "prompt_encoder.")
    inp = make_inputs(132, emb=(2, 256, 64, 64), sparse=(2, 1, 256))
    dense_pe = pe.get_dense_pe()
    out = dict(seed_params=131, seed_inputs=132, dense_pe=strided(dense_pe, 4), dense_pe_moments=moments(dense_pe),
               no_mask=pe(2)[:, :, 0, 0].numpy())
    for mm in (False, True):
        masks, iou, src = dec(image_embeddings=inp["emb"], image_pe=dense_pe, sparse_prompt_embeddings=inp["sparse"],
                              dense_prompt_embeddings=pe(2), multimask_output=mm)
        out[f"masks_{int(mm)}"] = strided(masks, 4)
        out[f"masks_moments_{int(mm)}"] = moments(masks)
        out[f"iou_{int(mm)}"] = iou.numpy()
        if mm:
            out["src"] = strided(src, 4)          # [B,256,64,64] view of the keys
            out["src_moments"] = moments(src)
    save("mask_decoder", **out)


def gen_mask_pooling():
    """(v) MaskAdapterPooling (D=768, 1024, SO400M 1152 @27x27) and MaskedPooling."""
    from lib.support_model.mask_adapter import MaskAdapterPooling, MaskedPooling
    for D, g in ((768, 24), (1024, 24), (1152, 27)):
        sd = ocfg.random_state(ocfg.mask_adapter_spec(D, "mp."), seed=141)
        m = load_strict(MaskAdapterPooling(x_in_channel=D, mask_adatpet_network_in_channel=512, mask_downscaling_mid_channel=16,
                                           mask_adatpet_network_mid_channel=256, num_output_maps=8), sd, "mp.")
        inp = make_inputs(142, feat=(2, D, g, g), mask=("mask", 2, 384))
        # intermediate maps too (pins GenerateMaskAdapterMap)
        import torch.nn.functional as F
        mk = F.interpolate(inp["mask"], size=(g, g), mode="bilinear", align_corners=False)
        maps = m.get_mask_map(m.channel_clip_to_maskadapter(inp["feat"]), mk)
        save(f"mask_adapter_D{D}", seed_params=141, seed_inputs=142, D=D, g=g, y=m(inp["feat"], inp["mask"]).numpy(),
             maps=maps.numpy(), mask_small=mk.numpy())
    inp = make_inputs(143, feat=(2, 768, 24, 24), mask=("mask", 2, 384))
    save("masked_pooling", seed_inputs=143, y=MaskedPooling()(inp["feat"], inp["mask"]).numpy())


def gen_fuse():
    """(vi) CirFuseModule.compose_img_text."""
    from lib.support_model.cir_feature_fuse import CirFuseModule
    for D in (768, 1024):
        sd = ocfg.random_state(ocfg.fuse_spec(D, "f."), seed=151)
        m = load_strict(CirFuseModule(D, D), sd, "f.")
        inp = make_inputs(152, img=(4, D), txt=(4, D))
        r = m.compose_img_text(inp["img"], inp["txt"])
        save(f"cir_fuse_D{D}", seed_params=151, seed_inputs=152, D=D, y=r["repres"].numpy(), dyn=r["dynamic_scalar"].numpy())


def gen_region():
    """(viii) utils/loss_func.mask_pooling + cosine similarity."""
    from utils.loss_func import mask_pooling
    import torch.nn.functional as F
    inp = make_inputs(161, emb=(3, 256, 64, 64), mask=("mask", 3, 256), feat=(3, 1, 256))
    r = mask_pooling(inp["emb"], inp["mask"])
    save("region_embedding", seed_inputs=161, y=r.numpy(),
         cos=F.cosine_similarity(r, F.normalize(inp["feat"], dim=-1), dim=-1).numpy())


# ------------------------------------------------------------------------------------------
# (vii) top-level glue through an in-memory open_clip stand-in (generator-only; never shipped as product)
# ------------------------------------------------------------------------------------------
class _StandInBlock(torch.nn.Module):        # timm-style pre-LN block
    def __init__(self, D, heads, mlp):
        super().__init__()
        nn = torch.nn
        self.heads = heads
        self.norm1 = nn.LayerNorm(D, eps=1e-6)
        self.attn = nn.Module()
        self.attn.qkv = nn.Linear(D, 3 * D)
        self.attn.proj = nn.Linear(D, D)
        self.norm2 = nn.LayerNorm(D, eps=1e-6)
        self.mlp = nn.Module()
        self.mlp.fc1 = nn.Linear(D, mlp)
        self.mlp.fc2 = nn.Linear(mlp, D)

    def forward(self, x):
        import torch.nn.functional as F
        N, T, D = x.shape
        qkv = self.attn.qkv(self.norm1(x)).reshape(N, T, 3, self.heads, D // self.heads).permute(2, 0, 3, 1, 4)
        o = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2]).transpose(1, 2).reshape(N, T, D)
        x = x + self.attn.proj(o)
        return x + self.mlp.fc2(F.gelu(self.mlp.fc1(self.norm2(x))))


class _StandInPatch(torch.nn.Module):
    def __init__(self, D, patch):
        super().__init__()
        self.proj = torch.nn.Conv2d(3, D, patch, patch)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class _StandInTextBlock(torch.nn.Module):    # open_clip ResidualAttentionBlock
    def __init__(self, D, heads, mlp):
        super().__init__()
        nn = torch.nn
        self.ln_1 = nn.LayerNorm(D, eps=1e-6)
        self.attn = nn.MultiheadAttention(D, heads, batch_first=True)
        self.ln_2 = nn.LayerNorm(D, eps=1e-6)
        self.mlp = nn.Module()
        self.mlp.c_fc = nn.Linear(D, mlp)
        self.mlp.c_proj = nn.Linear(mlp, D)

    def forward(self, x):
        import torch.nn.functional as F
        h = self.ln_1(x)
        x = x + self.attn(h, h, h, need_weights=False)[0]
        return x + self.mlp.c_proj(F.gelu(self.mlp.c_fc(self.ln_2(x))))


class _StandInClip(torch.nn.Module):
    """Provides exactly the five touch-points the reference uses (siglip_openclip.py:12,15,26,30-35,53)."""

    def __init__(self, cfg):
        super().__init__()
        nn = torch.nn
        D = cfg["dim"]
        self.visual = nn.Module()
        t = self.visual.trunk = nn.Module()
        t.patch_embed = _StandInPatch(D, cfg["patch"])
        t.pos_embed = nn.Parameter(torch.zeros(1, (cfg["image"] // cfg["patch"]) ** 2, D))
        t.pos_drop = nn.Identity()
        t.blocks = nn.ModuleList(_StandInBlock(D, cfg["heads"], cfg["mlp"]) for _ in range(cfg["depth"]))
        t.norm = nn.LayerNorm(D, eps=1e-6)
        self.text = nn.Module()
        self.text.token_embedding = nn.Embedding(cfg["vocab"], D)
        self.text.positional_embedding = nn.Parameter(torch.zeros(cfg["ctx"], D))
        self.text.transformer = nn.Module()
        self.text.transformer.resblocks = nn.ModuleList(_StandInTextBlock(D, cfg["t_heads"], cfg["t_mlp"]) for _ in range(cfg["t_depth"]))
        self.text.ln_final = nn.LayerNorm(D, eps=1e-6)
        self.text.text_projection = nn.Linear(D, D)
        self.logit_scale = nn.Parameter(torch.zeros(()))
        self.logit_bias = nn.Parameter(torch.zeros(()))

    def encode_image(self, x):      # result is dead on the live path: any [N,D] tensor will do
        t = self.visual.trunk
        h = t.patch_embed(x) + t.pos_embed
        for b in t.blocks:
            h = b(h)
        return t.norm(h).mean(1)

    def encode_text(self, tokens):
        x = self.text.token_embedding(tokens) + self.text.positional_embedding[: tokens.shape[1]]
        for b in self.text.transformer.resblocks:
            x = b(x)
        return self.text.text_projection(self.text.ln_final(x)[:, -1])


def gen_toplevel():
    """(vii) lib.build_model.build_model_with_query_support_feat + forward, SigLIP replaced by a 2-block stand-in."""
    gcfg = dict(ocfg.siglip_cfg("ViT-B-16-SigLIP-384"), depth=2, t_depth=2, vocab=512)
    stub = types.ModuleType("open_clip")
    stub.create_model_and_transforms = lambda name, pretrained=None: (_StandInClip(gcfg), None, None)
    stub.get_tokenizer = lambda name: None
    sys.modules["open_clip"] = stub
    from lib.build_model import build_model_with_query_support_feat
    for pooling in ("MaskAdapterPooling", "MaskedPooling"):
        model = build_model_with_query_support_feat("sam_base", "ViT-B-16-SigLIP-384", None, None, pooling).eval()
        spec = ocfg.model_spec("sam_base", "ViT-B-16-SigLIP-384", pooling)
        # shrink SigLIP part of the inventory to the stand-in's depth / vocab and drop the MAP head
        spec = {k: v for k, v in spec.items() if "attn_pool" not in k and not (".siglip." in k and any(
            f".blocks.{i}." in k or f".resblocks.{i}." in k for i in range(2, 12)))}
        spec["support_branch.siglip.model.text.token_embedding.weight"] = (512, 768)
        sd = ocfg.random_state(spec, seed=171)
        model.load_state_dict(sd, strict=True)
        keys = sorted(model.state_dict().keys())
        inp = make_inputs(172, q=(1, 3, 1024, 1024), s=(1, 3, 384, 384), text=("tokens", 1, 64, 512), mask=("mask", 1, 384))
        out = dict(seed_params=171, seed_inputs=172, keys=np.array(keys))
        for mm in (True, False):
            masks, emb, feat = model(query_image_inputs=inp["q"], support_image_inputs=inp["s"], change_text_inputs=inp["text"],
                                     support_mask_inputs=inp["mask"], multimask_output=mm)
            out[f"masks_{int(mm)}"] = strided(masks, 4)
            out[f"masks_moments_{int(mm)}"] = moments(masks)
            out["emb"] = strided(emb, 4)
            out["emb_moments"] = moments(emb)
            out["feat"] = feat.numpy()
        save(f"toplevel_{pooling}", **out)
    # the reference's own full key inventory for the non-SigLIP parts (drop-in contract)
    ref_keys = [k for k in keys if ".siglip." not in k]
    save("state_dict_keys_sam_base", keys=np.array(ref_keys))


def gen_toplevel_autocast():
    """(vii-b) the SAME top-level model / parameters / inputs as gen_toplevel, run the way the reference runs inference:
    under bf16 autocast (utils/vailder.py:416 `with accelerator.autocast():`, config/vaild_config/vaild_a.yaml:4
    mixed_precision bf16) - here torch.autocast("cpu", dtype=torch.bfloat16), the only autocast this container can execute.
    What is committed is the reference's own bf16-mode ERROR against its fp32 outputs (the toplevel_* fixtures): the yardstick
    the HIP bf16 mode's error budget is tied to (tests/test_gpu_parity.py::test_full_depth_bf16_vs_reference_golden_with_counts)."""
    gcfg = dict(ocfg.siglip_cfg("ViT-B-16-SigLIP-384"), depth=2, t_depth=2, vocab=512)
    stub = types.ModuleType("open_clip")
    stub.create_model_and_transforms = lambda name, pretrained=None: (_StandInClip(gcfg), None, None)
    stub.get_tokenizer = lambda name: None
    sys.modules["open_clip"] = stub
    from lib.build_model import build_model_with_query_support_feat
    for pooling in ("MaskAdapterPooling", "MaskedPooling"):
        model = build_model_with_query_support_feat("sam_base", "ViT-B-16-SigLIP-384", None, None, pooling).eval()
        spec = ocfg.model_spec("sam_base", "ViT-B-16-SigLIP-384", pooling)
        spec = {k: v for k, v in spec.items() if "attn_pool" not in k and not (".siglip." in k and any(
            f".blocks.{i}." in k or f".resblocks.{i}." in k for i in range(2, 12)))}
        spec["support_branch.siglip.model.text.token_embedding.weight"] = (512, 768)
        model.load_state_dict(ocfg.random_state(spec, seed=171), strict=True)
        inp = make_inputs(172, q=(1, 3, 1024, 1024), s=(1, 3, 384, 384), text=("tokens", 1, 64, 512), mask=("mask", 1, 384))
        out = dict(seed_params=171, seed_inputs=172)
        for mm in (True, False):
            kw = dict(query_image_inputs=inp["q"], support_image_inputs=inp["s"], change_text_inputs=inp["text"],
                      support_mask_inputs=inp["mask"], multimask_output=mm)
            m32, e32, f32_ = model(**kw)
            with torch.autocast("cpu", dtype=torch.bfloat16):
                m16, e16, f16_ = model(**kw)
            m16, e16, f16_ = m16.float(), e16.float(), f16_.float()
            rel = lambda a, b: float((a - b).norm() / b.norm())
            out[f"masks_{int(mm)}"] = strided(m16, 4)
            out[f"masks_rel_l2_{int(mm)}"] = rel(m16, m32)
            out[f"masks_max_abs_{int(mm)}"] = float((m16 - m32).abs().max())
            out[f"masks_sign_flips_{int(mm)}"] = int(((m16 > 0) != (m32 > 0)).sum())
            out[f"masks_sign_flips_strided_{int(mm)}"] = int(((m16 > 0) != (m32 > 0))[..., ::4, ::4].sum())
            out["emb"] = strided(e16, 4)
            out["emb_rel_l2"] = rel(e16, e32)
            out["emb_max_abs"] = float((e16 - e32).abs().max())
            out["feat"] = f16_.numpy()
            out["feat_rel_l2"] = rel(f16_, f32_)
            out["feat_max_abs"] = float((f16_ - f32_).abs().max())
            print(f"  {pooling} multimask={mm}: autocast-bf16 vs fp32 of the reference: emb rel-L2 {out['emb_rel_l2']:.3e}, "
                  f"masks rel-L2 {out[f'masks_rel_l2_{int(mm)}']:.3e}, sign flips {out[f'masks_sign_flips_{int(mm)}']}/65536, feat rel-L2 {out['feat_rel_l2']:.3e}")
        save(f"toplevel_autocast_bf16_{pooling}", **out)


GENS = dict(attention=gen_sam_attention, block=gen_sam_block, encoder=gen_sam_encoder, decoder=gen_decoder,
            pooling=gen_mask_pooling, fuse=gen_fuse, region=gen_region, toplevel=gen_toplevel, toplevel_autocast=gen_toplevel_autocast)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    a = ap.parse_args()
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    for name, fn in GENS.items():
        if a.only and name not in a.only:
            continue
        print(f"[{name}]")
        fn()
