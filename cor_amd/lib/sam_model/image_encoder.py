"""SAM ViTDet image encoder: parameter tree with the reference's names (lib/sam_model/image_encoder.py:57-102,
152-167,212-223,386); the arithmetic is cor_amd.engine.sam_encoder on HIP kernels."""
from typing import Tuple

import torch
from torch import nn

from .common import LayerNorm2d, MLPBlock
from ... import engine


class PatchEmbed(nn.Module):
    def __init__(self, patch: int, in_chans: int, embed_dim: int):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch, stride=patch)


class Attention(nn.Module):
    def __init__(self, dim: int, num_heads: int, size: int):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)
        self.rel_pos_h = nn.Parameter(torch.zeros(2 * size - 1, dim // num_heads))
        self.rel_pos_w = nn.Parameter(torch.zeros(2 * size - 1, dim // num_heads))


class Block(nn.Module):
    def __init__(self, dim: int, num_heads: int, window_size: int, grid: int):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, num_heads, grid if window_size == 0 else window_size)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = MLPBlock(dim, dim * 4)
        self.window_size = window_size


class ImageEncoderViT(nn.Module):
    def __init__(self, img_size=1024, patch_size=16, embed_dim=768, depth=12, num_heads=12, out_chans=256,
                 window_size=14, global_attn_indexes: Tuple[int, ...] = ()):
        super().__init__()
        if embed_dim // num_heads not in (16, 32, 64, 80):
            raise ValueError("cor_amd SAM attention kernels take head_dim 64 (SAM-B/L), 80 (SAM-H) or 16/32 (reduced test models)")
        self.img_size = img_size
        grid = img_size // patch_size
        self.cfg = dict(dim=embed_dim, depth=depth, heads=num_heads, global_idx=tuple(global_attn_indexes), window=window_size,
                        img=img_size, patch=patch_size, out=out_chans)
        self.patch_embed = PatchEmbed(patch_size, 3, embed_dim)
        self.pos_embed = nn.Parameter(torch.zeros(1, grid, grid, embed_dim))
        self.blocks = nn.ModuleList(Block(embed_dim, num_heads, 0 if i in global_attn_indexes else window_size, grid) for i in range(depth))
        self.neck = nn.Sequential(nn.Conv2d(embed_dim, out_chans, kernel_size=1, bias=False), LayerNorm2d(out_chans),
                                  nn.Conv2d(out_chans, out_chans, kernel_size=3, padding=1, bias=False), LayerNorm2d(out_chans))

    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False
