"""SigLIP towers: parameter tree under the open_clip / timm key names the released CORE checkpoints use
(`support_branch.siglip.model.visual.trunk.*`, `...model.text.*`), without depending on open_clip or timm.

The reference wraps `open_clip.create_model_and_transforms` (lib/support_model/siglip_openclip.py:12) and touches
exactly trunk.{patch_embed,pos_embed,blocks,norm} (:30-35) and encode_text (:53); this module provides those
parameters and cor_amd.engine.siglip_vision / siglip_text run them. Key names are restated from the published
open_clip_torch 2.31.0 / timm 1.0.15 sources (not available offline): treat as unverified until a real checkpoint
is loaded with strict=True. The MAP pooling head (`attn_pool.*`) is dead on the live path but kept so that
strict loading of a full checkpoint succeeds. Tokenisation is the data loader's job (utils/dataloader.py:128).
"""
import torch
from torch import nn

from ... import config


class _Mlp(nn.Module):
    def __init__(self, D, hidden, names=("fc1", "fc2")):
        super().__init__()
        setattr(self, names[0], nn.Linear(D, hidden))
        setattr(self, names[1], nn.Linear(hidden, D))


class _VitAttention(nn.Module):
    def __init__(self, D):
        super().__init__()
        self.qkv = nn.Linear(D, 3 * D)
        self.proj = nn.Linear(D, D)


class _VitBlock(nn.Module):
    def __init__(self, D, hidden):
        super().__init__()
        self.norm1 = nn.LayerNorm(D, eps=1e-6)
        self.attn = _VitAttention(D)
        self.norm2 = nn.LayerNorm(D, eps=1e-6)
        self.mlp = _Mlp(D, hidden)


class _PatchEmbed(nn.Module):
    def __init__(self, D, patch):
        super().__init__()
        self.proj = nn.Conv2d(3, D, kernel_size=patch, stride=patch)


class _AttnPool(nn.Module):            # timm AttentionPoolLatent (unused on the live path)
    def __init__(self, D, hidden):
        super().__init__()
        self.latent = nn.Parameter(torch.zeros(1, 1, D))
        self.q = nn.Linear(D, D)
        self.kv = nn.Linear(D, 2 * D)
        self.proj = nn.Linear(D, D)
        self.norm = nn.LayerNorm(D, eps=1e-6)
        self.mlp = _Mlp(D, hidden)


class _Trunk(nn.Module):
    def __init__(self, g, with_map_head):
        super().__init__()
        D = g["dim"]
        self.patch_embed = _PatchEmbed(D, g["patch"])
        self.pos_embed = nn.Parameter(torch.zeros(1, (g["image"] // g["patch"]) ** 2, D))
        self.blocks = nn.ModuleList(_VitBlock(D, g["mlp"]) for _ in range(g["depth"]))
        self.norm = nn.LayerNorm(D, eps=1e-6)
        if with_map_head:
            self.attn_pool = _AttnPool(D, g["mlp"])


class _Visual(nn.Module):
    def __init__(self, g, with_map_head):
        super().__init__()
        self.trunk = _Trunk(g, with_map_head)


class _Mha(nn.Module):                 # nn.MultiheadAttention's parameter names
    def __init__(self, D):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.zeros(3 * D, D))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * D))
        self.out_proj = nn.Linear(D, D)


class _ResBlock(nn.Module):
    def __init__(self, D, hidden):
        super().__init__()
        self.ln_1 = nn.LayerNorm(D, eps=1e-6)
        self.attn = _Mha(D)
        self.ln_2 = nn.LayerNorm(D, eps=1e-6)
        self.mlp = _Mlp(D, hidden, ("c_fc", "c_proj"))


class _TextTransformer(nn.Module):
    def __init__(self, g):
        super().__init__()
        self.resblocks = nn.ModuleList(_ResBlock(g["dim"], g["t_mlp"]) for _ in range(g["t_depth"]))


class _Text(nn.Module):
    def __init__(self, g):
        super().__init__()
        D = g["dim"]
        self.token_embedding = nn.Embedding(g["vocab"], D)
        self.positional_embedding = nn.Parameter(torch.zeros(g["ctx"], D))
        self.transformer = _TextTransformer(g)
        self.ln_final = nn.LayerNorm(D, eps=1e-6)
        self.text_projection = nn.Linear(D, D)


class _ClipModel(nn.Module):
    def __init__(self, g, with_map_head):
        super().__init__()
        self.visual = _Visual(g, with_map_head)
        self.text = _Text(g)
        self.logit_scale = nn.Parameter(torch.zeros(()))
        self.logit_bias = nn.Parameter(torch.zeros(()))


class SigLIP(nn.Module):
    def __init__(self, model_name: str = "ViT-SO400M-14-SigLIP-384", pretrained: str = None, cfg: dict = None,
                 with_map_head: bool = True):
        super().__init__()
        self.cfg = config.normalize_siglip_cfg(dict(cfg)) if cfg is not None else config.siglip_cfg(model_name)
        self.model = _ClipModel(self.cfg, with_map_head)
        self.text_tokenizer = None            # the reference resolves an HF-hub tokenizer by name; not available offline
        if pretrained is not None:            # open_clip checkpoints are a plain state_dict of `model`
            sd = torch.load(pretrained, map_location="cpu")
            self.model.load_state_dict(sd.get("state_dict", sd), strict=True)
            print(f"Load SigLIP Checkpoint: {pretrained}.")

    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False
