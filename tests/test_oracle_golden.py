"""Pins the CPU oracle (oracle/) against golden vectors produced by the REFERENCE's own modules
(tools/make_golden.py, run in the build container). CPU-only: runs under -m "not gpu"."""
import numpy as np
import pytest
import torch

from oracle import config as ocfg, sam, support, retrieval, model as omodel
from tests.golden_util import load, make_inputs, moments

torch.set_grad_enabled(False)
TOL = dict(rtol=2e-4, atol=2e-5)


def close(a, b, **kw):
    a = a.numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, **(kw or TOL))


@pytest.mark.parametrize("tag", ["win14", "glob16"])
def test_sam_attention(tag):
    g = load(f"sam_attention_{tag}")
    S, dim, heads = int(g["S"]), int(g["dim"]), int(g["heads"])
    cfg = dict(dim=dim, heads=heads, depth=1, global_idx=(0,), window=14, img=S * 16, patch=16, out=16)
    spec = {k: v for k, v in ocfg.sam_encoder_spec(cfg, "e.").items() if k.startswith("e.blocks.0.attn.")}
    sd = ocfg.random_state(spec, int(g["seed_params"]))
    x = make_inputs(int(g["seed_inputs"]), x=(3, S, S, dim))["x"]
    close(sam.vit_attention(sd, "e.blocks.0.attn.", x, heads), g["y"])


@pytest.mark.parametrize("tag", ["window", "global"])
def test_sam_block(tag):
    g = load(f"sam_block_{tag}")
    dim, heads, grid, win = int(g["dim"]), int(g["heads"]), int(g["g"]), int(g["window"])
    cfg = dict(dim=dim, heads=heads, depth=1, global_idx=(0,) if win == 0 else (), window=14, img=grid * 16, patch=16, out=16)
    spec = {k: v for k, v in ocfg.sam_encoder_spec(cfg, "e.").items() if k.startswith("e.blocks.0.")}
    sd = ocfg.random_state(spec, int(g["seed_params"]))
    x = make_inputs(int(g["seed_inputs"]), x=(2, grid, grid, dim))["x"]
    close(sam.vit_block(sd, "e.blocks.0.", x, heads, win), g["y"])


@pytest.mark.parametrize("tag", ["img256", "img1024"])
def test_sam_encoder(tag):
    g = load(f"sam_encoder_{tag}")
    cfg = dict(dim=int(g["dim"]), heads=int(g["heads"]), depth=int(g["depth"]), global_idx=tuple(int(i) for i in g["global_idx"]),
               window=14, img=int(g["img"]), patch=16, out=int(g["out"]))
    sd = ocfg.random_state(ocfg.sam_encoder_spec(cfg), int(g["seed_params"]))
    x = make_inputs(int(g["seed_inputs"]), x=(int(g["B"]), 3, cfg["img"], cfg["img"]))["x"]
    y = sam.image_encoder(sd, x, cfg)
    close(y, g["y"], rtol=5e-4, atol=5e-5)
    close(moments(y), g["y_moments"], rtol=1e-5, atol=1e-7)


def test_mask_decoder_and_prompt_encoder():
    g = load("mask_decoder")
    sd = ocfg.random_state(dict(ocfg.mask_decoder_spec(), **ocfg.prompt_encoder_spec()), int(g["seed_params"]))
    inp = make_inputs(int(g["seed_inputs"]), emb=(2, 256, 64, 64), sparse=(2, 1, 256))
    pe = sam.dense_pe(sd)
    close(pe[..., ::4, ::4], g["dense_pe"], rtol=1e-4, atol=2e-5)
    close(moments(pe), g["dense_pe_moments"], rtol=1e-5, atol=1e-7)
    dense = sam.dense_no_mask(sd, 2)
    close(dense[:, :, 0, 0], g["no_mask"])
    for mm in (0, 1):
        masks, iou, keys = sam.mask_decoder(sd, inp["emb"], pe, inp["sparse"], dense, bool(mm))
        close(masks[..., ::4, ::4], g[f"masks_{mm}"], rtol=1e-3, atol=1e-3)
        close(moments(masks), g[f"masks_moments_{mm}"], rtol=1e-4, atol=1e-6)
        close(iou, g[f"iou_{mm}"], rtol=1e-3, atol=1e-4)
    src = keys.transpose(1, 2).reshape(2, 256, 64, 64)   # mask_decoder.py:132 view
    close(src[..., ::4, ::4], g["src"], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("D,grid", [(768, 24), (1024, 24), (1152, 27)])
def test_mask_adapter_pooling(D, grid):
    g = load(f"mask_adapter_D{D}")
    sd = ocfg.random_state(ocfg.mask_adapter_spec(D, "mp."), int(g["seed_params"]))
    inp = make_inputs(int(g["seed_inputs"]), feat=(2, D, grid, grid), mask=("mask", 2, 384))
    close(support.bilinear_resize(inp["mask"], grid, grid), g["mask_small"], rtol=1e-5, atol=1e-6)
    y, maps = support.mask_adapter_pooling(sd, inp["feat"], inp["mask"], "mp.", return_maps=True)
    close(maps, g["maps"], rtol=1e-3, atol=1e-4)
    close(y, g["y"], rtol=1e-3, atol=1e-5)


def test_masked_pooling():
    g = load("masked_pooling")
    inp = make_inputs(int(g["seed_inputs"]), feat=(2, 768, 24, 24), mask=("mask", 2, 384))
    close(support.masked_pooling(inp["feat"], inp["mask"]), g["y"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("D", [768, 1024])
def test_cir_fuse(D):
    g = load(f"cir_fuse_D{D}")
    sd = ocfg.random_state(ocfg.fuse_spec(D, "f."), int(g["seed_params"]))
    inp = make_inputs(int(g["seed_inputs"]), img=(4, D), txt=(4, D))
    close(support.cir_fuse(sd, inp["img"], inp["txt"], "f."), g["y"], rtol=1e-4, atol=1e-6)


def test_region_embedding_and_cosine():
    g = load("region_embedding")
    inp = make_inputs(int(g["seed_inputs"]), emb=(3, 256, 64, 64), mask=("mask", 3, 256), feat=(3, 1, 256))
    r = retrieval.region_embedding(inp["emb"], inp["mask"])
    close(r, g["y"], rtol=1e-4, atol=1e-6)
    close(retrieval.cosine(r, torch.nn.functional.normalize(inp["feat"], dim=-1)), g["cos"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("pooling", ["MaskAdapterPooling", "MaskedPooling"])
def test_toplevel_forward(pooling):
    """The reference's own glue (build_model.py / sam_with_sup_branch.py / support_branch.py), run in the build
    container through an in-memory open_clip stand-in with a 2-block SigLIP (tools/make_golden.py gen_toplevel)."""
    g = load(f"toplevel_{pooling}")
    spec = ocfg.model_spec("sam_base", "ViT-B-16-SigLIP-384", pooling)
    spec = {k: v for k, v in spec.items() if "attn_pool" not in k and not (".siglip." in k and any(
        f".blocks.{i}." in k or f".resblocks.{i}." in k for i in range(2, 12)))}
    spec["support_branch.siglip.model.text.token_embedding.weight"] = (512, 768)
    assert sorted(spec) == [str(k) for k in g["keys"]]          # reference state_dict keys == oracle inventory
    sd = ocfg.random_state(spec, int(g["seed_params"]))
    inp = make_inputs(int(g["seed_inputs"]), q=(1, 3, 1024, 1024), s=(1, 3, 384, 384), text=("tokens", 1, 64, 512),
                      mask=("mask", 1, 384))
    import oracle.config as c
    name = "ViT-B-16-SigLIP-384"
    saved = dict(c.SIGLIP[name])
    c.SIGLIP[name] = dict(saved, depth=2, t_depth=2, vocab=512)
    try:
        for mm in (1, 0):
            masks, emb, feat = omodel.forward(sd, "sam_base", name, pooling, inp["q"], inp["s"], inp["text"], inp["mask"], bool(mm))
            close(masks[..., ::4, ::4], g[f"masks_{mm}"], rtol=2e-3, atol=2e-3)
            close(moments(masks), g[f"masks_moments_{mm}"], rtol=1e-3, atol=1e-4)
        close(emb[..., ::4, ::4], g["emb"], rtol=2e-3, atol=2e-4)
        close(feat, g["feat"], rtol=1e-3, atol=1e-5)
    finally:
        c.SIGLIP[name] = saved


def test_state_dict_inventory_matches_reference():
    keys = [str(k) for k in load("state_dict_keys_sam_base")["keys"]]
    spec = ocfg.model_spec("sam_base", "ViT-B-16-SigLIP-384", "MaskedPooling")
    mine = sorted(k for k in spec if ".siglip." not in k)
    assert mine == keys


# ----------------------------------------------------------------------------------------------------------------------
# input pre-processing: the oracle's restatement of Pillow's BILINEAR resize vs Pillow's own output
def test_preprocess_resize_matches_pillow_golden():
    """tests/golden/preprocess_resize.npz was produced by Pillow itself (tools/make_golden_preprocess.py); the oracle must
    reproduce it bit for bit, and the product's host tables (cor_amd.preprocess) must equal the oracle's."""
    import os
    from oracle import preprocess as OP
    from cor_amd import preprocess as CP
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "preprocess_resize.npz"))
    n = 0
    for k in d.files:
        if not k.endswith("_in"):
            continue
        want = d[k[:-3] + "_out"]
        got = OP.resize_bilinear_u8(d[k], want.shape[0], want.shape[1])
        assert got.dtype == np.uint8 and np.array_equal(got, want), k
        n += 1
    assert n >= 6
    for i, o in [(131, 24), (23, 48), (600, 1024), (1500, 384), (384, 384), (7, 16)]:
        b1, k1, s1 = OP.precompute_coeffs(i, o)
        b2, k2, s2 = CP._tables_host(i, o)
        assert s1 == s2 and np.array_equal(b1, b2) and np.array_equal(k1, k2)


def test_preprocess_oracle_vs_live_pillow_full_size():
    """Full-size property check against the installed Pillow (skipped where Pillow is absent): 1024 / 384 targets from odd sizes."""
    PIL_Image = pytest.importorskip("PIL.Image")
    from oracle import preprocess as OP
    rng = np.random.default_rng(5)
    for (h, w, c, s) in [(301, 457, 3, 384), (640, 480, 3, 1024), (480, 640, 1, 384)]:
        a = rng.integers(0, 256, size=(h, w, c), dtype=np.uint8)
        im = PIL_Image.fromarray(a if c == 3 else a[:, :, 0])
        want = np.asarray(im.resize((s, s), PIL_Image.BILINEAR))
        got = OP.resize_bilinear_u8(a if c == 3 else a[:, :, 0], s, s)
        assert np.array_equal(got, want)
    x = OP.preprocess_image(a[:, :, 0], 384, normalize=False)
    assert x.shape == (1, 384, 384) and x.dtype == np.float32 and 0.0 <= x.min() and x.max() <= 1.0


def test_mask_resize_restatement_vs_pillow_and_torch():
    """The oracle's restatement of cv2.resize(INTER_LINEAR) on float probability maps (utils/vailder.py:459-473; cv2 is not installed
    here, so this half of the post-processing is NOT pinned by the reference itself) against two independent implementations of
    the same published arithmetic (half-pixel centres, two taps, edge replication, no antialiasing): torch's
    F.interpolate(bilinear, align_corners=False) for up- and down-scaling, and Pillow's float ("F" mode) BILINEAR resize for
    up-scaling (Pillow antialiases when shrinking, cv2 does not)."""
    import numpy as np
    import torch
    from PIL import Image
    from oracle.support import bilinear_resize
    g = torch.Generator().manual_seed(11)
    for (ih, iw, oh, ow) in ((256, 256, 480, 640), (256, 256, 333, 500), (256, 256, 1024, 1024), (256, 256, 257, 300), (256, 256, 200, 120), (64, 64, 37, 91)):
        x = torch.rand((1, 1, ih, iw), generator=g)
        got = bilinear_resize(x, oh, ow)
        ref_t = torch.nn.functional.interpolate(x, size=(oh, ow), mode="bilinear", align_corners=False)
        assert float((got - ref_t).abs().max()) <= 1e-5, (ih, iw, oh, ow)        # tap weights in float32 here and there: rounding only
        if oh >= ih and ow >= iw:
            pil = np.asarray(Image.fromarray(x[0, 0].numpy(), mode="F").resize((ow, oh), Image.BILINEAR))
            assert float(np.abs(got[0, 0].numpy() - pil).max()) <= 3e-5, (ih, iw, oh, ow)   # Pillow computes its weights in double


@pytest.mark.parametrize("pooling", ["MaskAdapterPooling", "MaskedPooling"])
def test_reference_autocast_fixture_is_the_same_model_in_bf16(pooling):
    """tests/golden/toplevel_autocast_bf16_*.npz = the reference's top-level forward under torch.autocast("cpu", bf16)
    (tools/make_golden.py gen_toplevel_autocast) on the parameters / inputs of toplevel_*.npz: the same outputs up to bf16-level
    error (its recorded full-tensor error statistics agree with what the sub-sampled arrays show), i.e. a usable yardstick for
    the HIP bf16 mode's budgets."""
    g, ga = load(f"toplevel_{pooling}"), load(f"toplevel_autocast_bf16_{pooling}")
    assert int(g["seed_params"]) == int(ga["seed_params"]) and int(g["seed_inputs"]) == int(ga["seed_inputs"])
    for key, stat in (("emb", "emb_rel_l2"), ("masks_1", "masks_rel_l2_1"), ("masks_0", "masks_rel_l2_0"), ("feat", "feat_rel_l2")):
        a, b = torch.from_numpy(ga[key]).double(), torch.from_numpy(g[key]).double()
        rel = float((a - b).norm() / b.norm())
        assert 1e-4 < rel < 0.1, (key, rel)                       # bf16-level, not fp32-level and not garbage
        # (the arrays are every 4th pixel = ONE phase of the decoder's 4x4 up-sampling pattern: a biased sample of the full-tensor figure)
        assert 0.5 * float(ga[stat]) <= rel <= 2.0 * float(ga[stat]), (key, rel, float(ga[stat]))
