"""Architecture cross-check of the oracle's SigLIP towers (parity UNPINNED by the reference: open_clip / timm are not
vendored) against transformers.models.siglip built from config (random init, nothing downloaded): patch tokens after
the final norm, and the text feature (last token -> projection). CPU-only."""
import pytest
import torch

from oracle import siglip as osig

torch.set_grad_enabled(False)


def _remap(hf_sd, depth, D):
    sd = {}
    v, hv = "v.", "vision_model."
    sd[v + "patch_embed.proj.weight"] = hf_sd[hv + "embeddings.patch_embedding.weight"]
    sd[v + "patch_embed.proj.bias"] = hf_sd[hv + "embeddings.patch_embedding.bias"]
    sd[v + "pos_embed"] = hf_sd[hv + "embeddings.position_embedding.weight"][None]
    t, ht = "t.", "text_model."
    sd[t + "token_embedding.weight"] = hf_sd[ht + "embeddings.token_embedding.weight"]
    sd[t + "positional_embedding"] = hf_sd[ht + "embeddings.position_embedding.weight"]
    for i in range(depth):
        for (dst, src, names) in ((f"{v}blocks.{i}.", f"{hv}encoder.layers.{i}.", ("norm1", "norm2", "attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2")),
                                  (f"{t}transformer.resblocks.{i}.", f"{ht}encoder.layers.{i}.", ("ln_1", "ln_2", "attn.in_proj", "attn.out_proj", "mlp.c_fc", "mlp.c_proj"))):
            n1, n2, qkv, proj, f1, f2 = names
            for p in ("weight", "bias"):
                sd[f"{dst}{n1}.{p}"] = hf_sd[f"{src}layer_norm1.{p}"]
                sd[f"{dst}{n2}.{p}"] = hf_sd[f"{src}layer_norm2.{p}"]
                cat = torch.cat([hf_sd[f"{src}self_attn.{x}_proj.{p}"] for x in "qkv"], 0)
                sd[f"{dst}{qkv}.{p}" if qkv == "attn.qkv" else f"{dst}{qkv}_{p}"] = cat
                sd[f"{dst}{proj}.{p}"] = hf_sd[f"{src}self_attn.out_proj.{p}"]
                sd[f"{dst}{f1}.{p}"] = hf_sd[f"{src}mlp.fc1.{p}"]
                sd[f"{dst}{f2}.{p}"] = hf_sd[f"{src}mlp.fc2.{p}"]
    for p in ("weight", "bias"):
        sd[f"{v}norm.{p}"] = hf_sd[f"{hv}post_layernorm.{p}"]
        sd[f"{t}ln_final.{p}"] = hf_sd[f"{ht}final_layer_norm.{p}"]
        sd[f"{t}text_projection.{p}"] = hf_sd[f"{ht}head.{p}"]
    return sd


@pytest.mark.parametrize("act,kind", [("gelu", "erf"), ("gelu_pytorch_tanh", "tanh")])
def test_oracle_siglip_towers_match_hf_architecture(act, kind):
    tr = pytest.importorskip("transformers")
    from transformers import SiglipConfig, SiglipModel
    D, depth, heads = 64, 2, 4
    cfg = SiglipConfig(
        vision_config=dict(hidden_size=D, intermediate_size=4 * D, num_hidden_layers=depth, num_attention_heads=heads, image_size=32,
                           patch_size=8, hidden_act=act, layer_norm_eps=1e-6),
        text_config=dict(hidden_size=D, intermediate_size=4 * D, num_hidden_layers=depth, num_attention_heads=heads, vocab_size=100,
                         max_position_embeddings=16, hidden_act=act, layer_norm_eps=1e-6, projection_size=D))
    torch.manual_seed(0)
    hf = SiglipModel(cfg).eval()
    for p in hf.parameters():                      # HF inits biases / LN to constants: randomise everything
        p.copy_(torch.randn_like(p) * 0.2 + (1.0 if p.dim() == 1 and "layer_norm" in "" else 0.0))
    sd = _remap(hf.state_dict(), depth, D)
    g = dict(dim=D, depth=depth, heads=heads, t_depth=depth, t_heads=heads, gelu=kind)
    img = torch.randn(2, 3, 32, 32)
    ids = torch.randint(0, 100, (2, 16))
    ref_tok = hf.vision_model(pixel_values=img).last_hidden_state
    ref_txt = hf.text_model(input_ids=ids).pooler_output
    torch.testing.assert_close(osig.vision_tokens(sd, img, g, "v."), ref_tok, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(osig.text_features(sd, ids, g, "t.", normalize=False), ref_txt, rtol=1e-4, atol=1e-4)
